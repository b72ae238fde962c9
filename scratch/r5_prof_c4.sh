#!/bin/bash
# usage: r5_prof_c4.sh <tag> [ENV=VAL ...]: kernel stats of the C4-share step (scratch/run_configs.py c4)
export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
O=gpurun_out/prof_$tag; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 scratch/run_configs.py c4 > $O/out.txt 2> $O/trace.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
tail -1 $O/out.txt
python scratch/stats_top.py $O 13 | head -40
rm -rf $O/trace
