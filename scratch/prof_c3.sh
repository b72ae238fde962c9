#!/bin/bash
# kernel stats of the C3 step (TACORL with action-decoder fine-tuning) and of PlayLMP.training_step: 25 steps each
export TMPDIR=/tmp
O=gpurun_out/prof_c3; mkdir -p $O
ONLY=c3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 scratch/run_c3.py > $O/run.log 2> $O/trace.err
python scratch/stats_top.py $O/trace 25 > $O/stats_top.txt
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/trace
cat $O/run.log | tail -2; head -30 $O/stats_top.txt
