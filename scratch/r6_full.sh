#!/bin/bash
# full GPU suite + the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6/full_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6/full_tests.log
tail -8 gpurun_out/r6/full_tests.log
timeout 900 python bench.py > gpurun_out/r6/bench_full.json 2> gpurun_out/r6/bench_full.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/bench_full.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['chip_conditioning']['unconditioned_ms'], d['step_time']['median_ms'], d['roofline']['frac'], d['roofline']['avg_ms'], d['roofline']['back_to_back_ms'])
for k,v in d['configs'].items(): print(k, v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk not in('note','rccl','graphs_per_step')})
PY
