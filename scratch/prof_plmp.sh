#!/bin/bash
# kernel stats of PlayLMP.training_step (B=256, T=16): 25 steps
export TMPDIR=/tmp
O=gpurun_out/prof_plmp; mkdir -p $O
ONLY=plmp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 scratch/run_c3.py > $O/run.log 2> $O/trace.err
python scratch/stats_top.py $O/trace 25 > $O/stats_top.txt
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/trace
cat $O/run.log | tail -2; head -40 $O/stats_top.txt
