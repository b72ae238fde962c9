#!/bin/bash
mkdir -p gpurun_out/r4hd
timeout 900 python -m pytest tests -x -q -m gpu -k "plan_recognition or playlmp or play_lmp or c1" > gpurun_out/r4hd/test.txt 2>&1; echo "test rc=$?" >> gpurun_out/r4hd/test.txt
tail -4 gpurun_out/r4hd/test.txt
for B in 32 256; do timeout 300 python scratch/ab_plmp.py $B demb_dummy 0 1 2 2>&1 | tail -1; done
