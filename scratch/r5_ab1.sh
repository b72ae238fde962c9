#!/bin/bash
# round 5, session 2: tests of the new launch batching + same-box A/B of each switch on the headline step
export TMPDIR=/tmp
O=gpurun_out/r5ab1; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "rnn_linear or prep_and_reduce or mlp_fused_backward or encoder_fused_backward or encoder_bwd" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 600 python -m pytest tests/test_step_gpu.py -q -m gpu -x -k "tacorl_step or cql_offline_step" > $O/tests2.txt 2>&1; tail -3 $O/tests2.txt
timeout 300 python scratch/ab_step.py ops:prep_batch.enabled False True 3 > $O/ab_prep.txt 2>&1; tail -2 $O/ab_prep.txt
timeout 300 python scratch/ab_step.py ops:reduce_batch.enabled False True 3 > $O/ab_reduce.txt 2>&1; tail -2 $O/ab_reduce.txt
timeout 300 python scratch/ab_step.py env:TACORL_RNN_HEADS_TILE 0 1 3 > $O/ab_heads.txt 2>&1; tail -2 $O/ab_heads.txt
timeout 300 python scratch/ab_step.py env:TACORL_RNN_HEADS_TILE 0 2 2 > $O/ab_heads2.txt 2>&1; tail -2 $O/ab_heads2.txt
