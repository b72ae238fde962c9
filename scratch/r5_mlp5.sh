#!/bin/bash
# same-box alternating A/B of two library builds on the many-row MLP kernels alone
export TMPDIR=/tmp
O=gpurun_out/r5mlp; mkdir -p $O
for r in 1 2 3 4; do
  echo -n "head  "; TACORL_HIP_LIB=scratch/libs/head.so timeout 200 python scratch/bench_mlp_big.py 2>/dev/null | head -1
  echo -n "tree  "; timeout 200 python scratch/bench_mlp_big.py 2>/dev/null | head -1
done | tee $O/ab_packed.txt
