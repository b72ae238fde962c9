#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4feed; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --feeder hbm --steps 200 --warmup 20 --no-cpu-baseline --no-configs --no-distribution > $O/line.json 2> $O/err.txt
f=$(ls $O/trace/*/*kernel_stats.csv | head -1)
head -12 $f | cut -c1-160 > $O/stats_head.txt
grep -i "pack\|gather\|copy" $f | cut -c1-200 >> $O/stats_head.txt
rm -rf $O/trace
cat $O/stats_head.txt; tail -c 600 $O/line.json
