"""The N-rank code path of bench.py / the modules (sharded batches, two all-reduces per step between
hipGraph segments) on a single-GPU box: two processes share cuda:0 and reduce through gloo (RCCL
refuses two ranks on one device; on a multi-GPU node the backend is RCCL).  Checks that the run
finishes, losses are finite and both replicas end with identical parameters."""
import json
import os
import sys

import pytest

from tests.proc_util import free_port, run_group

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_stay_in_sync():
    env = dict(os.environ, TACORL_DIST_BACKEND="gloo", TACORL_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--batch", "64", "--no-cpu-baseline"]
    out = run_group(cmd, env, ROOT, timeout=420)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2"
    assert res["config"]["losses_finite"] is True
    assert res["config"]["replicas_in_sync"] is True


def test_bench_self_launcher_two_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment starts its own rank processes
    (the command the driver's scaling run would use if it were not under torch.distributed.run), reports the
    weak-scaling value and the strong-scaling block, noise differs per rank, replicas stay in sync."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TACORL_DIST_BACKEND="gloo", TACORL_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "64",
           "--no-cpu-baseline", "--no-distribution"]
    out = run_group(cmd, env, ROOT, timeout=420)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["replicas_in_sync"] is True
    st = res["strong"]
    assert st["per_gpu_batch"] == 32 and st["global_batch"] == 64 and st["value"] > 0 and st["replicas_in_sync"] is True


def test_bench_supervisor_falls_back_when_a_rank_hangs_in_the_step():
    """VERDICT r4 #2 on the real step: under torch.distributed.run (the driver's form) rank 1's worker stalls right after
    its first graph replay of the 'segments' attempt (injected).  Both supervisors see the silence, kill their workers and
    start fresh processes with the next form; the line comes from that form and carries the N-rank evidence fields."""
    env = dict(os.environ, TACORL_DIST_BACKEND="gloo", TACORL_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1",
               TACORL_BENCH_INJECT_HANG="segments:1", TACORL_BENCH_STAGE_TIMEOUT="45")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
           "--batch", "64", "--no-cpu-baseline", "--no-distribution"]
    out = run_group(cmd, env, ROOT, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    at = res["launcher"]["attempts"]
    assert [x["form"] for x in at] == ["segments", "eager"] and "stalled" in at[0]["outcome"] and at[1]["outcome"] == "ok"
    assert res["config"]["hip_graph"] is False and "no hipGraph" in res["config"]["collectives"]
    assert res["config"]["rccl_ranks_seen"] == 2 and res["config"]["replicas_in_sync"] is True
    assert res["config"]["rank_ms_per_step"]["min"] <= res["config"]["rank_ms_per_step"]["max"] == res["ms_per_step"]


def test_two_rank_shards_equal_the_full_batch_step():
    """TACORL (frozen / fine-tuned action decoder), CQL_Offline, PlayLMP: two ranks on per-sample shards with sharded
    noise, hipGraph segments around the all-reduces, against the single-rank full-batch step (tests/dist_shard_script.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "dist_shard_script.py")]
    out = run_group(cmd, env, ROOT, timeout=600)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def _one_rank(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    return run_group([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_script.py"), *args], env, ROOT, timeout=timeout)


def test_rccl_one_rank_collectives_between_graph_segments():
    """backend nccl (RCCL), world_size 1, TACORL_FORCE_COLLECTIVES=1: the broadcast, both all-reduces of a step between its
    three hipGraph segments, PlayLMP's arena all-reduce and the log reduction really execute in librccl; TACORL (frozen and
    fine-tuned decoder), CQL_Offline and PlayLMP end three graph-mode steps where the collective-free single graph ends."""
    out = _one_rank([])
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
    assert "librccl" in out.stdout


def test_rccl_one_rank_collectives_inside_the_graph():
    """TACORL_GRAPH_COLLECTIVES=1: the all-reduces captured as nodes of the step's ONE graph."""
    out = _one_rank(["--in-graph"])
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
