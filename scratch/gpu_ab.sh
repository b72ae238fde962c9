#!/bin/bash
# usage: gpu_ab.sh <tag>  : kernel stats of the bench step with the env given by the caller
export TMPDIR=/tmp
tag=$1
O=gpurun_out/ab_$tag; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-distribution > $O/bench.json 2> $O/trace.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
python scratch/stats_top.py $O/trace 23 > $O/stats_top.txt
rm -rf $O/trace
grep -E "ebw|total kernel|softargmax" $O/stats_top.txt
python -c "import json;d=json.load(open('$O/bench.json'));print('ms_per_step',d['ms_per_step'],'enc',d['roofline']['avg_ms'])"
