#!/bin/bash
# alternating processes: HEAD build of mlp_fused.hip vs the tree, headline step
export TMPDIR=/tmp
O=gpurun_out/r5ab8; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "mlp_ or prep_and" 2>&1 | tail -2
for r in 1 2 3; do
  echo -n "head "; TACORL_HIP_LIB=scratch/libs/head.so timeout 200 python scratch/ab_step.py engine.lean_mlp_acts True True 2 2>/dev/null | tail -1
  echo -n "tree "; timeout 200 python scratch/ab_step.py engine.lean_mlp_acts True True 2 2>/dev/null | tail -1
done | tee $O/ab_tiles.txt
