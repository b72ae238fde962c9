#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r5ab3; mkdir -p $O
timeout 300 python scratch/marks2.py > $O/marks.txt 2>&1; tail -3 $O/marks.txt
timeout 300 python scratch/ab_step.py ops:reduce_batch.enabled False True 3 > $O/ab_reduce.txt 2>&1; tail -2 $O/ab_reduce.txt
timeout 300 python scratch/ab_step.py engine.wgrad_side_streams False True 3 > $O/ab_side.txt 2>&1; tail -2 $O/ab_side.txt
timeout 300 python scratch/ab_step.py engine.wgrad_side_streams False '("q",)' 3 > $O/ab_side_q.txt 2>&1; tail -2 $O/ab_side_q.txt
timeout 300 python scratch/ab_step.py engine.wgrad_side_streams False '("q","pi")' 3 > $O/ab_side_qpi.txt 2>&1; tail -2 $O/ab_side_qpi.txt
timeout 300 python scratch/ab_step.py engine.adam_writes_mirrors False True 3 > $O/ab_mirrors.txt 2>&1; tail -2 $O/ab_mirrors.txt
timeout 300 python scratch/ab_step.py env:TACORL_RNN_SMALL_UPTO 2 3 3 > $O/ab_small.txt 2>&1; tail -2 $O/ab_small.txt
timeout 300 python scratch/ab_step.py env:TACORL_RNN_SMALL_UPTO 2 1 3 > $O/ab_small1.txt 2>&1; tail -2 $O/ab_small1.txt
