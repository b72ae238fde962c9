#!/bin/bash
mkdir -p gpurun_out/r5prpf; O=gpurun_out/r5prpf
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "plan_recognition or pr_" > $O/test.txt 2>&1; tail -3 $O/test.txt
for r in 1 2; do
echo "old: $(PR_SHAPE=64,32,64,32 TACORL_HIP_LIB=scratch/libs/pr_old.so timeout 100 python scratch/run_pr.py 2>&1 | grep 'sample in launch')"
echo "new: $(PR_SHAPE=64,32,64,32 timeout 100 python scratch/run_pr.py 2>&1 | grep 'sample in launch')"
done | tee $O/ab.txt
