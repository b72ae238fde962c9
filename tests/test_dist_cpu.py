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
    assert res == {"launch_check": True, "world": 2, "sum": 3.0}
    # a rank that fails takes the launch down with a non-zero exit code instead of hanging the others
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=dict(env, TACORL_DIST_BACKEND="no-such-backend"), cwd=root, capture_output=True, text=True,
                         timeout=300)
    assert bad.returncode != 0
