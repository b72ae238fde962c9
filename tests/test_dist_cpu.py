"""world_size-2 gloo test of the data-parallel contract the GPU path relies on: shard the batch and
the noise by sample, take per-rank gradients pre-scaled by 1/world, sum them with an all-reduce ->
identical to the full-batch gradients (checked with the CPU oracle; runs without a GPU)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.golden_util import Golden, spec_for


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import tacorl_oracle as O
    from tacorl_amd import dist as D

    g = Golden("cql_q")
    spec = spec_for(g)
    B = g.cfg["B"]
    batch, noise = g.batch(0), g.noise(0)
    # make the batch divisible by the world size: duplicate it (2B samples)
    dup = lambda t: torch.cat([t, t.flip(0)], 0)  # noqa: E731
    big = {"observations": {k: {c: dup(v) for c, v in d.items()} for k, d in batch["observations"].items()},
           "next_observations": {k: {c: dup(v) for c, v in d.items()} for k, d in batch["next_observations"].items()},
           "actions": dup(batch["actions"]), "rewards": dup(batch["rewards"]), "terminals": dup(batch["terminals"])}
    bign = {}
    for k, v in noise.items():
        if k in ("eps_cur", "eps_nxt", "g_cur", "g_nxt"):
            bign[k] = torch.cat([v, v.flip(1)], 1)
        elif k == "u_rand":
            n = spec.n
            w = v.view(n, B, -1)
            bign[k] = torch.cat([w, w.flip(1)], 1).reshape(n * 2 * B, -1)
        else:
            bign[k] = dup(v)

    def grads_of(bt, nz):
        P = O.require_grad_(g.params())
        _, gr = O.cql_step(P, O.make_opts(P, spec), spec, bt, nz, g.cfg["epoch"])
        return gr

    mine = grads_of(D.shard_batch(big, rank, world), D.shard_noise(bign, rank, world, spec.n))
    names = sorted(k for k in mine if k.startswith(("actor.", "q1.", "q2.")))
    flat = torch.cat([mine[k].reshape(-1) for k in names]) / world
    D.allreduce_sum_([flat])
    if rank == 0:
        full = grads_of(big, bign)
        ref = torch.cat([full[k].reshape(-1) for k in names])
        q.put(((flat - ref).norm() / ref.norm()).item())
    dist.destroy_process_group()


def test_sharded_gradients_equal_full_batch():
    world, port = 2, 29533 + os.getpid() % 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    err = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
    assert err < 5e-5, err


def test_bench_self_launcher_rendezvous():
    """bench.py's own launcher (python bench.py --gpus 2, no torch.distributed.run): two child ranks rendezvous on
    127.0.0.1 and all-reduce once through gloo on host tensors.  `--launch-check` stops before any GPU work, so
    this runs here; the full 2-rank step through the same launcher is tests/test_dist_gpu.py."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["TACORL_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    launcher = res.pop("launcher")
    assert res == {"launch_check": True, "world": 2, "sum": 3.0}
    assert launcher["form"] == "segments" and launcher["attempts"] == [{"form": "segments", "outcome": "ok"}]
    # a rank that fails takes the launch down with a non-zero exit code instead of hanging the others
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=dict(env, TACORL_DIST_BACKEND="no-such-backend"), cwd=root, capture_output=True, text=True,
                         timeout=300)
    assert bad.returncode != 0


def _launch_check(cmd_prefix, extra_env, timeout=300):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(TACORL_DIST_BACKEND="gloo", TACORL_BENCH_STAGE_TIMEOUT="6", **extra_env)
    out = subprocess.run([sys.executable, *cmd_prefix, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         cwd=root, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, [json.loads(ln) for ln in lines]


def _torchrun():
    from tests.proc_util import free_port

    return ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
            str(free_port())]


def test_bench_supervisor_survives_a_hung_rank_self_launched():
    """a rank that stalls inside a collective form (injected: rank 1 sleeps for ever in the 'segments'
    attempt) must cost the stage timeout, not the line - the supervisor kills the attempt's workers and starts FRESH
    processes with the next form, and the line says which form produced it."""
    out, res = _launch_check([], {"TACORL_BENCH_INJECT_HANG": "segments:1"})
    assert out.returncode == 0 and len(res) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    at = res[0]["launcher"]["attempts"]
    assert res[0]["launcher"]["form"] == "eager" and [x["form"] for x in at] == ["segments", "eager"]
    assert "stalled" in at[0]["outcome"] and "rank 1" in at[0]["outcome"] and at[1]["outcome"] == "ok"
    assert res[0]["sum"] == 3.0


def test_bench_supervisor_survives_a_hung_rank_under_torchrun():
    """The driver's launch form: torch.distributed.run starts the two ranks, each of which is a supervisor (gloo group on
    host tensors, no GPU call) with one worker; a hang on rank 0's worker is seen by both, both move on together."""
    out, res = _launch_check(_torchrun(), {"TACORL_BENCH_INJECT_HANG": "segments:0"})
    assert out.returncode == 0 and len(res) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    at = res[0]["launcher"]["attempts"]
    assert res[0]["launcher"]["form"] == "eager" and [x["form"] for x in at] == ["segments", "eager"]
    assert "stalled" in at[0]["outcome"] and at[1]["outcome"] == "ok"
    assert res[0]["world"] == 2 and res[0]["sum"] == 3.0


def test_bench_supervisor_under_torchrun_without_faults():
    out, res = _launch_check(_torchrun(), {})
    assert out.returncode == 0 and len(res) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    assert res[0]["launcher"]["attempts"] == [{"form": "segments", "outcome": "ok"}]


def test_bench_supervisor_gives_up_with_nonzero_exit_when_every_form_hangs():
    out, res = _launch_check([], {"TACORL_BENCH_INJECT_HANG": "segments:1", "TACORL_BENCH_FORMS": "segments"})
    assert out.returncode != 0 and not res
