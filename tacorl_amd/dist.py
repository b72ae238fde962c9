"""Data-parallel plumbing: shard a replay batch and its noise by sample, all-reduce flat gradient
blocks.  Every loss on the hot path is a mean over per-sample terms (the CQL logsumexp is per sample,
reference cql_offline_lightning.py:374-387), so with equal shards the average of per-rank gradients
equals the full-batch gradient; kernels pre-scale by 1/world and the collective is a plain sum
(RCCL over xGMI on the GPUs, gloo in the CPU tests)."""
import os

import torch


def _slice(t, rank, world, dim=0):
    n = t.shape[dim]
    assert n % world == 0, f"batch dim {n} not divisible by world size {world}"
    k = n // world
    return t.narrow(dim, rank * k, k)


def shard_batch(batch, rank, world):
    """Per-sample slice of a (possibly nested) batch dict."""
    if isinstance(batch, dict):
        return {k: shard_batch(v, rank, world) for k, v in batch.items()}
    return _slice(batch, rank, world, 0) if torch.is_tensor(batch) and batch.dim() > 0 else batch


def shard_noise(noise, rank, world, n_samples):
    """Noise tensors are (B,..), (n,B,..) or sample-major flat (n*B,..): slice the B axis."""
    out = {}
    for k, v in noise.items():
        if isinstance(v, (list, tuple)):
            out[k] = [_slice(t, rank, world, 0) for t in v]
        elif k in ("eps_cur", "eps_nxt", "g_cur", "g_nxt"):
            out[k] = _slice(v, rank, world, 1)
        elif k == "u_rand":
            B = v.shape[0] // n_samples
            out[k] = _slice(v.view(n_samples, B, *v.shape[1:]), rank, world, 1).reshape(-1, *v.shape[1:])
        else:
            out[k] = _slice(v, rank, world, 0)
    return out


def group_ready():
    import torch.distributed as dist

    return dist.is_available() and dist.is_initialized()


def collectives_on(world):
    """Does a step with this world size issue its collectives?  With several ranks: always.  With one rank only when
    TACORL_FORCE_COLLECTIVES=1 and a process group exists: the 1-GPU boxes of the test pool then run the very RCCL calls
    (communicator init, stream ordering between hipGraph segments, capture inside a graph) an N-GPU step makes -
    an all-reduce over one rank is the identity, so results must equal the collective-free step bit for bit."""
    if world > 1:
        return True
    return os.environ.get("TACORL_FORCE_COLLECTIVES", "0") == "1" and group_ready()


def graph_collectives():
    """Are the step's all-reduces nodes of its ONE captured hipGraph (RCCL kernels captured on the step's stream) instead of
    eager calls between three graph segments?

    Default: **no** - segments, the form every N-rank test has run (gloo, 2 ranks) and the ordinary way RCCL is driven.
    The one-graph form is opt-in with TACORL_GRAPH_COLLECTIVES=1 (backend nccl only; gloo cannot be captured): it measures
    0.878 ms/step against 0.964 for segments on one MI355X with a 1-rank communicator (bench.py --probe segments), but an
    N > 1 RCCL capture has never run on this pool (1-GPU boxes), so the library does not bet a training run on it
    (ADVICE r4).  `bench.py --gpus N` opts in for its first attempt under a supervisor that falls back to segments
    (fresh processes) when the attempt fails or stalls; a refused capture inside a Trainer run falls back in
    GraphMixin._capture."""
    if os.environ.get("TACORL_GRAPH_COLLECTIVES", "0") != "1" or not group_ready():
        return False
    import torch.distributed as dist

    return dist.get_backend() == "nccl"


def refuse_graph_collectives(reason=""):
    """Called when a capture that contains collectives failed: from here on this process uses segments."""
    os.environ["TACORL_GRAPH_COLLECTIVES"] = "0"
    if reason:
        import sys

        print(f"[tacorl_amd.dist] all-reduces stay outside the hipGraph from now on: {reason}", file=sys.stderr, flush=True)


def any_rank(flag, device):
    """max over ranks of a host boolean (one tiny eager all-reduce; no-op without a group)."""
    if not group_ready():
        return bool(flag)
    import torch.distributed as dist

    t = torch.tensor([1.0 if flag else 0.0], device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(t.item() > 0)


def all_reduce_sum_(t):
    """In-place sum over ranks on the current stream (backend nccl = RCCL over xGMI; gloo in the CPU / 1-GPU tests)."""
    import torch.distributed as dist

    dist.all_reduce(t)


def allreduce_sum_(tensors, world=None):
    """In-place sum over ranks of a list of flat blocks (no-op without a process group)."""
    import torch.distributed as dist

    if not group_ready() or not collectives_on(dist.get_world_size() if world is None else world):
        return
    for t in tensors:
        dist.all_reduce(t)


def reduce_logs_(logs, world):
    """The step's logged scalars are per-rank batch means (or rank-independent values): their mean over ranks is the
    full-batch value the reference publishes with sync_dist=True (modules/tacorl/tacorl.py:196-202,
    play_lmp_for_rl.py:162,183,292-339).  One small collective, issued only on steps whose metrics are read back.
    Returns (summed copy, divisor to apply on the host)."""
    if not collectives_on(world) or not group_ready():
        return logs, 1.0
    import torch.distributed as dist

    out = logs.clone()  # (a slot that a later step does not rewrite must not be summed twice)
    dist.all_reduce(out)
    return out, float(dist.get_world_size())
