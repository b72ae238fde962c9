"""The N-rank code path of bench.py / the modules (sharded batches, two all-reduces per step between
hipGraph segments) on a single-GPU box: two processes share cuda:0 and reduce through gloo (RCCL
refuses two ranks on one device; on a multi-GPU node the backend is RCCL).  Checks that the run
finishes, losses are finite and both replicas end with identical parameters."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_stay_in_sync():
    env = dict(os.environ, TACORL_DIST_BACKEND="gloo", TACORL_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--batch", "64", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2"
    assert res["config"]["losses_finite"] is True
    assert res["config"]["replicas_in_sync"] is True
