"""Two ranks, one GPU (gloo): a training step on per-sample SHARDS with sharded noise must produce the parameters of the
single-rank step on the FULL batch - for TACORL (frozen and fine-tuned action decoder), CQL_Offline and PlayLMP, hipGraph
on (the multi-GPU path: collective-free graph segments around the eager all-reduces).  SURVEY 8e; reference
modules/tacorl/tacorl.py:206-233, modules/cql/cql_offline_lightning.py:519-542 under `strategy: ddp`.

Launched by tests/test_dist_gpu.py through torch.distributed.run with 2 processes.  The full batch is two golden
batches concatenated (even size); every loss on the path is a batch mean, the kernels pre-scale gradients by 1/world and
the collectives are sums, so shard and full agree to summation order: gradients to 1e-5 (norm-wise), parameters after
the two Adam steps to 1e-5 relative plus the one thing an Adam step adds on top - an element whose gradient is ~0 can take
its first update (lr * sign(g)) with either sign, so single elements may differ by a few lr."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tacorl_amd import dist as D  # noqa: E402
from tests import cfg_util as C  # noqa: E402
from tests.golden_util import Golden  # noqa: E402
from tests.test_step_gpu import ACTOR, build_tacorl, to_dev  # noqa: E402

PER_SAMPLE_MAJOR = ("eps_cur", "eps_nxt", "g_cur", "g_nxt")


def cat_tree(a, b, dim=0):
    if isinstance(a, dict):
        return {k: cat_tree(a[k], b[k], dim) for k in a}
    if isinstance(a, (list, tuple)):
        return [cat_tree(x, y, dim) for x, y in zip(a, b)]
    return torch.cat([torch.as_tensor(a), torch.as_tensor(b)], dim=dim) if torch.is_tensor(a) and a.dim() > 0 else a


def cat_noise(a, b, n_samples):
    out = {}
    for k in a:
        if k in PER_SAMPLE_MAJOR:
            out[k] = torch.cat([a[k], b[k]], dim=1)
        elif k == "u_rand":
            Ba, Bb = a[k].shape[0] // n_samples, b[k].shape[0] // n_samples
            out[k] = torch.cat([a[k].view(n_samples, Ba, -1), b[k].view(n_samples, Bb, -1)], dim=1).reshape(n_samples * (Ba + Bb), -1)
        else:
            out[k] = cat_tree(a[k], b[k])
    return out


def build(kind, g, world):
    if kind == "tacorl":
        return build_tacorl(g, world_size=world)
    if kind == "cql":
        from tacorl_amd.lightning import instantiate

        return instantiate(C.cql_cfg(device="cuda:0", world_size=world))
    from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP

    cams, c = sorted(g.cams), g.cfg
    pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=c["latent"],
              min_std=1e-4, dropout_p=0.0, max_position_embeddings=c["T"])
    ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10,
              latent_plan_dim=c["latent"], rnn_model="rnn_decoder", include_goal=False)
    return PlayLMP(plan_proposal=ACTOR, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams,
                   plan_proposal_goal_modalities=cams, plan_recognition_modalities=cams, action_decoder_modalities=cams,
                   real_world=True, lr=1e-4, kl_beta=1e-3, device="cuda:0", compute_dtype="f32", world_size=world)


def one_step(kind, mod, g, batch, noise):
    if "epoch" in g.cfg:
        mod.current_epoch = g.cfg["epoch"]
    mod.enable_graph()
    if kind == "playlmp":
        noise = {k: noise[k] for k in ("eps_plan", "u_plan") if k in noise}
    for _ in range(2):  # first call: eager warm-up + capture, second: graph replay - both are training steps
        mod.training_step(to_dev(batch, mod.device), noise=to_dev(noise, mod.device))
    torch.cuda.synchronize()
    grads = {k: v.detach().clone() for k, v in mod.named_gradients().items()}
    params = {k: v.detach().clone() for k, v in mod.state_dict().items() if v.dtype.is_floating_point}
    return grads, params, dict(mod.logged)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)
    cases = [("tacorl", "tacorl_q", 3e-4), ("tacorl", "tacorl_q_ad", 3e-4), ("cql", "cql_q", 3e-4), ("playlmp", "playlmp", 1e-4)]
    for kind, name, lr_max in cases:
        g = Golden(name)
        n_s = 4
        batch = cat_tree(g.batch(0), g.batch(1))
        noise = cat_noise(g.noise(0), g.noise(1), n_s)
        params0 = g.params()
        # the single-rank step on the full batch (world_size 1: no collectives)
        full = build(kind, g, 1)
        full.load_state_dict(params0, strict=False)
        g_full, p_full, l_full = one_step(kind, full, g, batch, noise)
        del full
        # the same step on this rank's shard; gradients meet in the module's collectives
        mod = build(kind, g, 2)
        mod.load_state_dict(params0, strict=False)
        sb, sn = D.shard_batch(batch, rank, world), D.shard_noise(noise, rank, world, n_s)
        g_sh, p_sh, l_sh = one_step(kind, mod, g, sb, sn)
        bad = []
        for k, v in g_full.items():
            if v.norm() > 0 and rel(g_sh[k], v) > 1e-5:
                # a gradient that is noise-level small against its block is exempt (summation order is all it is)
                if (g_sh[k] - v).norm() > 1e-6 * max(x.norm() for x in g_full.values()):
                    bad.append(f"grad {k}: rel {rel(g_sh[k], v):.3g}")
        for k, v in p_full.items():
            d = (p_sh[k] - v).abs()
            if rel(p_sh[k], v) > 1e-5 and float(d.max()) > 4.2 * lr_max:
                bad.append(f"param {k}: rel {rel(p_sh[k], v):.3g} max|d| {float(d.max()):.3g}")
            elif rel(p_sh[k], v) > 1e-3:
                bad.append(f"param {k}: rel {rel(p_sh[k], v):.3g}")
        # logged scalars: every rank publishes the mean over ranks of the per-shard batch means = the full-batch value
        # (the reference's sync_dist=True logs, modules/tacorl/tacorl.py:196-202, play_lmp_for_rl.py:162,183,292-339)
        assert l_full and set(l_full) <= set(l_sh), f"{name}: logged keys differ: {sorted(set(l_full) - set(l_sh))}"
        for k, v in l_full.items():
            tol = 2e-5 * max(abs(v), 1e-2)
            if "accuracy" in k:
                tol = 1e-6
            if abs(l_sh[k] - v) > tol:
                bad.append(f"log {k}: shards {l_sh[k]!r} full {v!r}")
        moved = sum(float((p_full[k] - params0[k].to(p_full[k].device)).abs().max()) > 0 for k in p_full if k in params0)
        assert moved > 0, f"{name}: the step did not move any parameter"
        assert not bad, f"{name} rank {rank}: shard != full\n" + "\n".join(bad[:20])
        if kind == "tacorl" and g.cfg.get("finetune_ad"):
            assert any(k.startswith("action_decoder.") for k in g_full), "fine-tuned decoder gradients missing"
            assert mod.ad.blk.grad.data_ptr() >= mod.engine.grad_arena.data_ptr() and \
                mod.ad.blk.grad.data_ptr() < mod.engine.grad_arena.data_ptr() + 4 * mod.engine.grad_arena.numel(), \
                "the decoder's gradient block is not part of the arena (second collective)"
        del mod
        torch.cuda.empty_cache()
        if rank == 0:
            print(f"{name}: shard == full ok", flush=True)
        dist.barrier()
    if rank == 0:
        print("ALL OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
