#!/bin/bash
# same-box A/B of the conv-backward fusion switches on the headline step: plain bench (ms/step) and kernel stats under rocprofv3
for v in 0 1 0 1; do
  export TACORL_EBW_FUSE3=$v
  python bench.py --no-configs --no-cpu-baseline --steps 400 > gpurun_out/fuse_ab_$v.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/fuse_ab_$v.json'));print('FUSE3=$v ms_per_step',d['ms_per_step'],'median',d['step_time']['median_ms'])"
done
for v in 0 1; do
  export TACORL_EBW_FUSE3=$v
  bash scratch/gpu_ab.sh fuse$v
done
