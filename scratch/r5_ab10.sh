#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5ab10; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "many_rows or mlp_persistent or mlp_fused_backward" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -x -k "c5 or c2" 2>&1 | tail -2
for r in 1 2 3; do
  echo -n "head  "; TACORL_HIP_LIB=scratch/libs/head.so timeout 200 python scratch/bench_mlp_big.py 2>/dev/null | head -1
  echo -n "tree  "; timeout 200 python scratch/bench_mlp_big.py 2>/dev/null | head -1
done | tee $O/ab_x32.txt
for r in 1 2; do
  echo -n "head "; TACORL_HIP_LIB=scratch/libs/head.so timeout 200 python scratch/ab_step.py engine.lean_mlp_acts True True 2 2>/dev/null | tail -1
  echo -n "tree "; timeout 200 python scratch/ab_step.py engine.lean_mlp_acts True True 2 2>/dev/null | tail -1
done | tee -a $O/ab_x32.txt
