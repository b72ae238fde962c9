#!/bin/bash
mkdir -p gpurun_out/r4pb
timeout 900 python -m pytest tests -x -q -m gpu -k "c1 or play_lmp or playlmp or plmp or plan_recognition" > gpurun_out/r4pb/test2.txt 2>&1; echo "test rc=$?" >> gpurun_out/r4pb/test2.txt
tail -4 gpurun_out/r4pb/test2.txt
bash scratch/r4_plmp_seq.sh
cp gpurun_out/r4plseq/seq.txt gpurun_out/r4pb/seq32.txt
