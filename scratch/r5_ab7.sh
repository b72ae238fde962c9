#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5ab7; mkdir -p $O
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_kernels_gpu.py -q -m gpu -x -k "playlmp or plan_recognition" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for B in 32 256; do
timeout 300 python scratch/ab_plmp.py $B pr.pr_wgrad_batched False True 2 2>/dev/null | tail -2
timeout 300 python scratch/ab_plmp.py $B ad.proj_in_ring False True 2 2>/dev/null | tail -2
done | tee $O/ab.txt
