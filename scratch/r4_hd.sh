#!/bin/bash
mkdir -p gpurun_out/r4hd
timeout 900 python -m pytest tests -x -q -m gpu -k "heads_dgrad or bptt or c1 or tacorl_q_ad or playlmp_step" > gpurun_out/r4hd/test.txt 2>&1; echo "test rc=$?" >> gpurun_out/r4hd/test.txt
tail -5 gpurun_out/r4hd/test.txt
for B in ; do
  timeout 300 python scratch/ab_plmp.py $B ad.heads_dgrad_ring False True 2 > gpurun_out/r4hd/ab_$B.txt 2>&1
  tail -2 gpurun_out/r4hd/ab_$B.txt
done
