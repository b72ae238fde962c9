#!/bin/bash
mkdir -p gpurun_out/r4hd
for B in 32 256; do
  timeout 300 python scratch/ab_plmp.py $B pp_forward_side False True 2 > gpurun_out/r4hd/abpp_$B.txt 2>&1
  tail -2 gpurun_out/r4hd/abpp_$B.txt
done
timeout 900 python -m pytest tests -x -q -m gpu -k "playlmp or play_lmp or c1 or twin" > gpurun_out/r4hd/test.txt 2>&1; echo "test rc=$?" >> gpurun_out/r4hd/test.txt
tail -4 gpurun_out/r4hd/test.txt
