#!/bin/bash
# usage: gpu_r3e.sh [-t] lib...   (-t: run the encoder kernel tests with the product library first)
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O; rm -f $O/*
if [ "$1" = "-t" ]; then shift; timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "encoder" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt; fi
for v in "$@"; do
  echo "== $v" >> $O/fused.txt
  TACORL_SCRATCH_LIB=$PWD/scratch/libs/$v.so timeout 120 python scratch/run_fused.py 4096 256 512 512 512 512 256 256 2>&1 | grep -v amdgpu.ids >> $O/fused.txt
done
cat $O/fused.txt
