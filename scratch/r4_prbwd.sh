#!/bin/bash
mkdir -p gpurun_out/r4pb
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "plan_recognition" > gpurun_out/r4pb/test.txt 2>&1; echo "test rc=$?" >> gpurun_out/r4pb/test.txt
tail -5 gpurun_out/r4pb/test.txt
for B in ; do
  timeout 300 python scratch/ab_plmp.py $B pr.fused_backward False True 3 > gpurun_out/r4pb/ab_$B.txt 2>&1
  tail -3 gpurun_out/r4pb/ab_$B.txt
done
