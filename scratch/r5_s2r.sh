#!/bin/bash
mkdir -p gpurun_out/r5s2r; O=gpurun_out/r5s2r
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "encoder" > $O/test.txt 2>&1; tail -3 $O/test.txt
for r in 1 2 3; do
echo "old: $(HW=128 TACORL_SCRATCH_LIB=scratch/libs/ef_old.so timeout 100 python scratch/run_fused.py 4096 1408a 2>&1 | grep imgs)"
echo "new: $(HW=128 timeout 100 python scratch/run_fused.py 4096 1408a 2>&1 | grep imgs)"
done | tee $O/ab.txt
