#!/bin/bash
# which preceding test makes test_fullsize_cql_baseline_c5's graph replay crash?
for sel in "tacorl_step_hipgraph or test_fullsize_cql_baseline_c5" "hipgraph_survives or test_fullsize_cql_baseline_c5" "tacorl_q_ad or test_fullsize_cql_baseline_c5" "tacorl_bc_ad or test_fullsize_cql_baseline_c5" "test_fullsize_tacorl or test_fullsize_cql_baseline_c5"; do
  timeout 600 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "$sel" > gpurun_out/bisect.log 2>&1
  echo "[$sel] rc=$? $(grep -E 'passed|failed|Segmentation' gpurun_out/bisect.log | tail -1)"
done
