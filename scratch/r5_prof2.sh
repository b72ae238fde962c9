#!/bin/bash
# usage: r5_prof2.sh <tag> [ENV=VAL ...] : kernel stats of the headline bench with env switches
export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
O=gpurun_out/prof_$tag; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-configs --no-cpu-baseline --steps 200 > $O/bench.json 2> $O/trace.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
python -c "import json;d=json.load(open('$O/bench.json'));r=d['roofline'];print('$tag ms_per_step',d['ms_per_step'],'in-step',r['avg_ms'],r['frac'])"
python scratch/stats_top.py $O 1 | head -45
rm -rf $O/trace
