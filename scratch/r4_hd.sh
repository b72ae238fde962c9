#!/bin/bash
mkdir -p gpurun_out/r4hd
timeout 900 python -m pytest tests -x -q -m gpu -k "plan_recognition or playlmp or play_lmp or c1 or twin" > gpurun_out/r4hd/test.txt 2>&1; echo "test rc=$?" >> gpurun_out/r4hd/test.txt
tail -5 gpurun_out/r4hd/test.txt
for B in ; do
  timeout 300 python scratch/ab_plmp.py $B pr.composed_head False True 2 > gpurun_out/r4hd/abc_$B.txt 2>&1
  tail -2 gpurun_out/r4hd/abc_$B.txt
done
