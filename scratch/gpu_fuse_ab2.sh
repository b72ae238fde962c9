#!/bin/bash
for v in 0 1 0 1; do
  export TACORL_EBW_FUSE3=$v
  python bench.py --no-configs --no-cpu-baseline --steps 400 --ad-every 100000 > gpurun_out/fuse_ab2_$v.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/fuse_ab2_$v.json'));print('noAD FUSE3=$v ms_per_step',d['ms_per_step'],'median',d['step_time']['median_ms'])"
done
