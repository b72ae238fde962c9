#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5ab6; mkdir -p $O
for c in 256 248 240 224 192 256 240; do
  echo -n "CUS=$c  "; TACORL_MLP_PERS_CUS=$c timeout 300 python scratch/run_configs.py c5 2>/dev/null | tail -1
done | tee $O/cus.txt
