#!/bin/bash
# same-box A/B of encoder_fused.hip builds at 128 x 128 (C4's launch shape: 4096 + 7 problems): usage r6_ef128_ab.sh lib1 lib2 ... ("product" = the tree's library)
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = product ]; then unset TACORL_HIP_LIB; else export TACORL_HIP_LIB=scratch/libs/$lib.so; fi
  echo -n "$lib 128: "; HW=128 timeout 120 python scratch/run_fused.py 4096 128 128 128 128 128a 128a 128a 128 128 128 128a 128a 128a 2>/dev/null | tail -1
  echo -n "$lib 84: "; HW=84 timeout 120 python scratch/run_fused.py 4096 512 256 256 256 512a 512a 512a 2>/dev/null | tail -1
done; done
