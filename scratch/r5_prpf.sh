#!/bin/bash
# plan-recognition inference launch, d_model 64: attention split over the waves, operands requested a phase ahead
# (old = HEAD's pr_fused.hip as scratch/libs/pr_old.so; pr_st.so = the working tree's with -DPR_STAMPS)
mkdir -p gpurun_out/r5prpf; O=gpurun_out/r5prpf
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "plan_recognition or pr_" > $O/test.txt 2>&1; tail -3 $O/test.txt
for r in 1 2; do for sh in 64,32,64,32 256,16,64,32; do
echo "old $sh: $(PR_SHAPE=$sh TACORL_HIP_LIB=scratch/libs/pr_old.so timeout 100 python scratch/run_pr.py 2>&1 | grep 'sample in launch')"
echo "new $sh: $(PR_SHAPE=$sh timeout 100 python scratch/run_pr.py 2>&1 | grep 'sample in launch')"
done; done | tee $O/ab.txt
PR_SHAPE=64,32,64,32 TACORL_HIP_LIB=scratch/libs/pr_st.so timeout 100 python scratch/run_pr.py 2>&1 | tail -7 | tee $O/stamps.txt
