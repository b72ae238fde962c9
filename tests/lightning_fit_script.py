"""GPU half of the Lightning-boundary test, run in a fresh interpreter (see tests/test_lightning_gpu.py).
usage: lightning_fit_script.py <standin|mini>.  Trainer.fit drives TACORL (reference default: action-decoder
fine-tuning on), CQL_Offline and PlayLMP for a few real steps from host batches; a run interrupted by a checkpoint
and resumed in a NEW module ends bit-identical to the uninterrupted run (parameters + Adam state + step counters
travel through the PL-layout checkpoint)."""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flavour = sys.argv[1]

from tacorl_amd import lightning as L, synth  # noqa: E402
from tests import cfg_util as C  # noqa: E402

if flavour == "standin":
    import pytorch_lightning as pl

    assert L.LightningModuleBase is pl.LightningModule
    Trainer, tkw = pl.Trainer, dict(gpus=1)
else:
    Trainer, tkw = L.MiniTrainer, {}

CAMS = {"rgb_static": (84, 84)}


def seeded(cls):
    class Seeded(cls):  # the step draws its noise from torch's device generator: key it to the global step
        def on_train_batch_start(self, batch, batch_idx, unused=0):
            torch.manual_seed(1000 + self.global_step)
            torch.cuda.manual_seed(1000 + self.global_step)
    Seeded.__name__ = cls.__name__
    return Seeded


def build(kind):
    from tacorl_amd.modules.cql.cql_offline_lightning import CQL_Offline
    from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
    from tacorl_amd.modules.tacorl.tacorl import TACORL

    torch.manual_seed(5)
    strip = lambda c: {k: v for k, v in c.items() if k not in ("_target_", "_recursive_")}  # noqa: E731
    if kind == "cql":
        return seeded(CQL_Offline)(**strip(C.cql_cfg(device="cuda:0")))
    lmp = PlayLMP(**strip(C.playlmp_cfg(device="cuda:0")))
    if kind == "playlmp":
        torch.manual_seed(5)
        return seeded(PlayLMP)(**strip(C.playlmp_cfg(device="cuda:0")))
    return seeded(TACORL)(play_lmp=lmp, **strip(C.tacorl_cfg(device="cuda:0")))


def batches(kind, n):
    if kind == "cql":
        return [synth.make_transition_batch(50 + i, 3, CAMS) for i in range(n)]
    return [synth.make_play_batch(50 + i, 2, 16, CAMS) for i in range(n)]


for kind in ("tacorl", "cql", "playlmp"):
    data = batches(kind, 4)  # host tensors: the trainer moves them (transfer_batch_to_device)
    full = build(kind)
    # steps per batch as the trainer counts them: PL >= 1.6 counts optimizer steps (one per optimizer per batch, as the
    # reference's manual optimisation steps them), MiniTrainer counts batches
    per = 1 if Trainer is L.MiniTrainer else len(L._as_list(full.configure_optimizers()))
    tr = Trainer(max_epochs=1, max_steps=4 * per, log_every_n_steps=1, **tkw)
    tr.fit(full, train_dataloaders=data)
    torch.cuda.synchronize()
    want = "train/total_loss" if kind == "playlmp" else "train/q1_loss"
    assert want in tr.logged_metrics and tr.logged_metrics[want] == tr.logged_metrics[want], tr.logged_metrics
    assert tr.global_step == 4 * per, (tr.global_step, per)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "last.ckpt")
        first = build(kind)
        t1 = Trainer(max_epochs=1, max_steps=2 * per, log_every_n_steps=1, **tkw)
        t1.fit(first, train_dataloaders=data)
        t1.save_checkpoint(path)
        resumed = build(kind)
        with torch.no_grad():  # make sure the values really come from the checkpoint
            for p in resumed.parameters():
                p.mul_(0.5)
        # (the checkpoint was written after t1's epoch counter moved on: give the resumed run epochs to spend)
        t2 = Trainer(max_epochs=10, max_steps=4 * per, log_every_n_steps=1, **tkw)
        t2.fit(resumed, train_dataloaders=data[2:], ckpt_path=path)
    torch.cuda.synchronize()
    assert t2.global_step == 4 * per
    sa, sb = full.state_dict(), resumed.state_dict()
    for k in sa:
        if sa[k].dtype.is_floating_point:
            assert torch.isfinite(sa[k]).all(), f"{kind}: {k} is not finite after 4 steps"
            assert torch.equal(sa[k], sb[k]), f"{kind}: resumed run differs from the uninterrupted one in {k}"
    names = {id(p): n for n, p in full.named_parameters()}
    for o, o2 in zip(tr.optimizers, t2.optimizers):
        for (_, p, m, v), (_, _, m2, v2) in zip(o._triples(), o2._triples()):
            assert torch.equal(m, m2) and torch.equal(v, v2), (
                kind, o.name, names.get(id(p)), (m - m2).abs().max().item(), (v - v2).abs().max().item())
    print(f"{kind}: ok  {want}={tr.logged_metrics[want]:.5g}")
print("ALL OK")
