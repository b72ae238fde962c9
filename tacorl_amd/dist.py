"""Data-parallel plumbing: shard a replay batch and its noise by sample, all-reduce flat gradient
blocks.  Every loss on the hot path is a mean over per-sample terms (the CQL logsumexp is per sample,
reference cql_offline_lightning.py:374-387), so with equal shards the average of per-rank gradients
equals the full-batch gradient; kernels pre-scale by 1/world and the collective is a plain sum
(RCCL over xGMI on the GPUs, gloo in the CPU tests)."""
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


def allreduce_sum_(tensors):
    """In-place sum over ranks of a list of flat blocks (no-op without a process group)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in tensors:
        dist.all_reduce(t)
