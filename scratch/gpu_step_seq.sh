#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/stepseq; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline --no-distribution > $O/bench.json 2> $O/err.txt
python scratch/step_sequence.py $O/trace > $O/step_sequence.txt
rm -rf $O/trace
cat $O/step_sequence.txt
