#!/bin/bash
# round 4, first GPU pass: new distributed tests (1-rank RCCL), new full-size bf16 gradient tests, bench with the segment probe
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
rm -f gpurun_out/parity_margins.jsonl
timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -m gpu > $O/dist.txt 2>&1; echo "dist rc=$?" >> $O/dist.txt
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "bf16_vs_rounded or c1_matches" > $O/full.txt 2>&1; echo "full rc=$?" >> $O/full.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "many_rows or frozen_bf16_hidden" > $O/kern.txt 2>&1; echo "kern rc=$?" >> $O/kern.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
cp gpurun_out/parity_margins.jsonl $O/ 2>/dev/null
tail -5 $O/dist.txt $O/full.txt $O/kern.txt; tail -c 1500 $O/bench.json
