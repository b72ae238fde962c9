#!/bin/bash
# kernel stats of the C4-like step (TACORL dual camera 128x128, A=32, T=32, B=64): 13 steps
export TMPDIR=/tmp
O=gpurun_out/prof_c4; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 scratch/run_configs.py c4 > $O/run.log 2> $O/trace.err
python scratch/stats_top.py $O/trace 13 > $O/stats_top.txt
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/trace
cat $O/run.log | tail -2; head -32 $O/stats_top.txt
