#!/bin/bash
O=gpurun_out/r5prpf; mkdir -p $O
for r in 1 2 3; do
echo "old: $(TACORL_HIP_LIB=scratch/libs/pr_old.so timeout 200 python scratch/run_configs.py c4 2>&1 | tail -1)"
echo "new: $(timeout 200 python scratch/run_configs.py c4 2>&1 | tail -1)"
done | tee $O/c4_ab.txt
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu -k "c4 or tacorl_configs or real_world or dualcam" > $O/test2.txt 2>&1; tail -3 $O/test2.txt
