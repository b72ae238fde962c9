#!/bin/bash
# the whole GPU suite + default bench (what the driver runs at round end)
export TMPDIR=/tmp
O=gpurun_out/r4suite; mkdir -p $O
rm -f gpurun_out/parity_margins.jsonl
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
cp gpurun_out/parity_margins.jsonl $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
tail -n 6 $O/tests.txt; tail -c 600 $O/bench.json
