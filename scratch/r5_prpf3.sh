#!/bin/bash
O=gpurun_out/r5prpf; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "plan_recognition or pr_" > $O/test3.txt 2>&1; tail -2 $O/test3.txt
for r in 1 2; do for sh in 64,32,64,32 256,16,64,32; do
echo "old $sh: $(PR_SHAPE=$sh TACORL_HIP_LIB=scratch/libs/pr_old.so timeout 100 python scratch/run_pr.py 2>&1 | grep 'sample in launch')"
echo "new $sh: $(PR_SHAPE=$sh timeout 100 python scratch/run_pr.py 2>&1 | grep 'sample in launch')"
done; done | tee $O/ab3.txt
