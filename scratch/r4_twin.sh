#!/bin/bash
mkdir -p gpurun_out/r4tw
for W in 0 1; do
  TACORL_RNN_TWIN_WIDE=$W timeout 300 python scratch/ab_plmp.py 256 ad.twin_pass False True 2 > gpurun_out/r4tw/abw_$W.txt 2>&1
  echo "wide=$W"; tail -2 gpurun_out/r4tw/abw_$W.txt
done
timeout 300 python scratch/ab_plmp.py 32 ad.twin_pass False True 2 > gpurun_out/r4tw/abt_32.txt 2>&1
tail -2 gpurun_out/r4tw/abt_32.txt
