#!/bin/bash
# round 6: leftover-pixel deferral in the fused encoder forward - parity tests, then the bench line without the configs block
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r6
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "encoder" > gpurun_out/r6/enc_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6/enc_tests.log
tail -15 gpurun_out/r6/enc_tests.log
timeout 600 python bench.py --steps 200 --warmup 20 --no-configs --no-cpu-baseline > gpurun_out/r6/bench_enc.json 2> gpurun_out/r6/bench_enc.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6/bench_enc.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['config']['chip_conditioning'].get('unconditioned_ms'), d['step_time']['median_ms'], d['roofline'])
PY
