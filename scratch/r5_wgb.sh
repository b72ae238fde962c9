#!/bin/bash
# batched RNN square weight gradients: test + same-process A/B on C3 and PlayLMP
mkdir -p gpurun_out/r5wgb; O=gpurun_out/r5wgb
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rnn_wgrad" > $O/test.txt 2>&1; tail -3 $O/test.txt
FINETUNE=1 timeout 200 python scratch/ab_step.py ad.wgrad_batched False True 3 > $O/ab_c3.txt 2>&1; tail -4 $O/ab_c3.txt
timeout 200 python scratch/ab_plmp.py 256 ad.wgrad_batched False True 3 > $O/ab_plmp256.txt 2>&1; tail -4 $O/ab_plmp256.txt
timeout 200 python scratch/ab_plmp.py 128 ad.wgrad_batched False True 3 > $O/ab_plmp128.txt 2>&1; tail -4 $O/ab_plmp128.txt
