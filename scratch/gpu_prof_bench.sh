#!/bin/bash
# usage: gpu_prof_bench.sh <tag> : rocprofv3 kernel trace + stats of `python3 bench.py --no-configs --no-cpu-baseline`
export TMPDIR=/tmp
tag=$1
O=gpurun_out/prof_$tag; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --no-configs --no-cpu-baseline > $O/bench.json 2> $O/trace.err
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
python scratch/encoder_launches.py $O/trace $O/encoder_launches.json
python -c "import json;d=json.load(open('$O/bench.json'));r=d['roofline'];print('ms_per_step',d['ms_per_step'],'in-step',r['avg_ms'],r['frac'],'b2b',r.get('back_to_back_ms'))"
rm -rf $O/trace
