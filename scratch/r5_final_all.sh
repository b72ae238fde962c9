#!/bin/bash
# the GPU suite (margins), smoke and the round-5 profiles from the same tree
export TMPDIR=/tmp
O=gpurun_out/r5suite; mkdir -p $O
rm -f gpurun_out/parity_margins.jsonl
timeout 2700 python -m pytest tests -q -m gpu > $O/tests.txt 2>&1; echo "tests rc=$?" >> $O/tests.txt
cp gpurun_out/parity_margins.jsonl $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
bash scratch/gpu_r05_final.sh > $O/final.txt 2>&1
tail -n 4 $O/tests.txt; tail -n 2 $O/smoke.txt; tail -n 12 $O/final.txt | cut -c1-300
