#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5ab11; mkdir -p $O
timeout 900 python -m pytest tests/test_step_gpu.py -q -m gpu -x -k "c4 or dualcam" 2>&1 | tail -2
for m in 0 1 0 1; do echo -n "MERGE=$m  "; TACORL_EF_MERGE_CAMS=$m timeout 300 python scratch/run_configs.py c4 2>/dev/null | tail -1; done | tee $O/c4_merge.txt
