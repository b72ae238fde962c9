#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5ab2; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "rnn_linear or action_decoder or rnn_" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x -k "bf16 or twin or playlmp_step or tacorl_step or trajectory" > $O/tests2.txt 2>&1; tail -5 $O/tests2.txt
timeout 300 python scratch/ab_step.py env:TACORL_AD_PROJ_RING 0 1 3 > $O/ab_ring.txt 2>&1; tail -2 $O/ab_ring.txt
FINETUNE=1 timeout 300 python scratch/ab_step.py env:TACORL_AD_PROJ_RING 0 1 2 > $O/ab_ring_c3.txt 2>&1; tail -2 $O/ab_ring_c3.txt
