#!/bin/bash
# round 6, first check: the new tests (graph staleness, in-situ decoder backward rows) and the default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -x -q -m gpu -k "in_place_parameter_edits or playlmp_c1 or (configs_bf16 and c3)" > gpurun_out/r6/check1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6/check1_tests.log
tail -15 gpurun_out/r6/check1_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r6/bench_driver_form.json 2> gpurun_out/r6/bench_driver_form.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/bench_driver_form.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['chip_conditioning'], d['roofline']['frac'])
for k,v in d['configs'].items(): print(k, v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='note'})
PY
