#!/bin/bash
# usage: prof_cfg.sh <c4|c5> <steps traced (warm + timed)> : kernel stats of one of run_configs.py's steps
export TMPDIR=/tmp
O=gpurun_out/prof_$1; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 scratch/run_configs.py $1 > $O/run.log 2> $O/trace.err
python scratch/stats_top.py $O/trace $2 > $O/stats_top.txt
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/trace
cat $O/run.log | tail -2; head -34 $O/stats_top.txt
