#!/bin/bash
mkdir -p gpurun_out/r4hd
timeout 900 python -m pytest tests -x -q -m gpu -k "heads_dgrad or rnn_wgrad or bptt or c1 or tacorl_q_ad or playlmp_step or twin" > gpurun_out/r4hd/test.txt 2>&1; echo "test rc=$?" >> gpurun_out/r4hd/test.txt
tail -5 gpurun_out/r4hd/test.txt
for B in ; do
  timeout 300 python scratch/ab_plmp.py $B ad.heads_wgrad_slabs False True 2 > gpurun_out/r4hd/abs_$B.txt 2>&1
  tail -2 gpurun_out/r4hd/abs_$B.txt
done
