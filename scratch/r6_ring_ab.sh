#!/bin/bash
# same-box A/B of encoder_ring.hip builds (scratch/libs/*.so via scratch/mklib_file.sh): usage r6_ring_ab.sh lib1 lib2 ... ("product" = the tree's library)
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = product ]; then unset TACORL_HIP_LIB; else export TACORL_HIP_LIB=scratch/libs/$lib.so; fi
  echo -n "$lib: "; HxW=150x200 timeout 120 python scratch/run_fused.py 2048 128 64 64 64 2>/dev/null | tail -1
  echo -n "$lib (with act): "; HxW=150x200 timeout 120 python scratch/run_fused.py 2048 128 64 64 64 64a 64a 64a 128a 128a 128a 2>/dev/null | tail -1
done; done
