#!/bin/bash
mkdir -p gpurun_out/r4st
timeout 600 python scratch/ab_step.py env:TACORL_RNN_SMALL_STAGES 2 4 3 2>&1 | tail -2
for B in 32 256; do
  for S in 2 4; do echo "stages=$S B=$B"; TACORL_RNN_SMALL_STAGES=$S timeout 300 python scratch/ab_plmp.py $B demb_dummy 0 1 1 2>&1 | tail -1; done
  echo "default B=$B"; timeout 300 python scratch/ab_plmp.py $B demb_dummy 0 1 1 2>&1 | tail -1
done
timeout 900 python -m pytest tests -x -q -m gpu -k "rnn or twin or bptt or tacorl_q_ad or playlmp_step" 2>&1 | tail -2
