#!/bin/bash
# usage: gpu_r3f.sh lib...  : the bench's problem mix (three problems store their activations)
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O; rm -f $O/*
for v in "$@"; do
  echo "== $v" >> $O/fused.txt
  TACORL_SCRATCH_LIB=$PWD/scratch/libs/$v.so timeout 120 python scratch/run_fused.py 4096 512a 256 512a 512a 512 512 2>&1 | grep -v amdgpu.ids >> $O/fused.txt
done
cat $O/fused.txt
