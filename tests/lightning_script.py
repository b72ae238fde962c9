"""Run by tests/test_lightning_cpu.py / test_lightning_gpu.py in a fresh interpreter, optionally with tests/fake_pl on
PYTHONPATH (then `import pytorch_lightning` is the strict stand-in and the tacorl_amd classes derive from ITS
LightningModule).  usage: lightning_script.py <cpu|gpu> <standin|mini>"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
where, flavour = sys.argv[1], sys.argv[2]
dev = "cpu" if where == "cpu" else "cuda:0"

from tacorl_amd import lightning as L  # noqa: E402
from tests import cfg_util as C  # noqa: E402

if flavour == "standin":
    import pytorch_lightning as pl

    assert "standin" in pl.__version__ and L.HAVE_PL and L.LightningModuleBase is pl.LightningModule
    Trainer = pl.Trainer
    tkw = dict(gpus=1) if where == "gpu" else {}
else:
    assert not L.HAVE_PL
    Trainer, tkw = L.MiniTrainer, {}


def build(kind):
    if kind == "cql":
        return L.instantiate(C.cql_cfg(device=dev))
    lmp = L.instantiate(C.playlmp_cfg(device=dev))
    if kind == "playlmp":
        return lmp
    return L.instantiate(C.tacorl_cfg(device=dev), play_lmp=lmp)


def perturb(model, opts, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * torch.randn(p.shape, generator=g).to(p.device))
        for o in opts:
            for blk, _, m, v in o._triples():
                m.copy_(torch.randn(m.shape, generator=g).to(m.device))
                v.copy_(torch.rand(v.shape, generator=g).to(v.device))
            for blk, *_ in o._entries:
                blk.step.fill_(7)


def same(a, b):
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if torch.is_tensor(a):
        return torch.equal(a.cpu(), b.cpu())
    return a == b


n_opt = {"cql": 5, "playlmp": 1, "tacorl": 6}
for kind in ("cql", "playlmp", "tacorl"):
    model = build(kind)
    assert isinstance(model, L.LightningModuleBase) and not model.automatic_optimization
    assert model.current_epoch == 0 and str(model.device) == dev
    model.log("train/outside_a_trainer", 1.0)  # no trainer attached: recorded, not forwarded
    assert model.logged["train/outside_a_trainer"] == 1.0
    tr = Trainer(max_epochs=1, **tkw)
    tr.fit(model, train_dataloaders=[])  # every check on the way in + hooks; no batches
    assert len(tr.optimizers) == n_opt[kind] and all(isinstance(o, torch.optim.Optimizer) for o in tr.optimizers)
    trainable = [p for p in model.parameters() if p.requires_grad]
    covered = {id(p) for o in tr.optimizers for g in o.param_groups for p in g["params"]}
    frozen_ok = {id(p) for n, p in model.named_parameters() if n.startswith(("target_q",))}
    assert {id(p) for p in trainable} - frozen_ok <= covered, f"{kind}: trainable parameters without an optimizer"
    # checkpoint round trip through the trainer: parameters, Adam moments, step counters, hyper-parameters
    perturb(model, tr.optimizers, 3)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "last.ckpt")
        tr.save_checkpoint(path)
        ck = torch.load(path, map_location="cpu", weights_only=False)
        assert {"epoch", "global_step", "state_dict", "optimizer_states", "hyper_parameters"} <= set(ck)
        hp = ck["hyper_parameters"]
        assert "play_lmp" not in hp and hp["real_world"] is True
        model2 = build(kind)
        tr2 = Trainer(max_epochs=1, **tkw)
        tr2.fit(model2, train_dataloaders=[], ckpt_path=path)
    assert same(dict(model.state_dict()), dict(model2.state_dict())), f"{kind}: state_dict round trip"
    for o, o2 in zip(tr.optimizers, tr2.optimizers):
        assert same(o.state_dict()["state"], o2.state_dict()["state"]), f"{kind}: optimizer state round trip"
        assert all(int(b.step.item()) == 7 for b, *_ in o2._entries)
    # the epoch comes from the trainer, and can be pinned by hand (tests, scripts without a trainer)
    tr2.current_epoch = 6
    assert model2.current_epoch == 6
    model2.current_epoch = 2
    assert model2.current_epoch == 2
    # moving / casting is refused, the no-op PL performs is accepted
    model2.to(dev)
    for bad in (lambda: model2.half(), lambda: model2.to("cpu" if where == "gpu" else "meta")):
        try:
            bad()
        except RuntimeError:
            pass
        else:
            raise AssertionError("moving / casting the module must raise")
    print(f"{kind}: ok ({len(trainable)} trainable tensors, {n_opt[kind]} optimizers)")

# a plain nn.Module is rejected by the trainer, as by PL
try:
    Trainer(max_epochs=1).fit(torch.nn.Linear(2, 2), train_dataloaders=[])
except TypeError:
    pass
else:
    raise AssertionError("Trainer.fit must reject a non-LightningModule")

# sub-configs the HIP path does not honour fail loudly instead of training another model
bad_cfgs = [
    C.cql_cfg(device=dev, actor_encoder={"_target_": "x.LateFusion", "networks": {"rgb_static": {"_target_": "tacorl.networks.visual_encoders.encoder.ResNet18"}}}),
    C.cql_cfg(device=dev, critic_encoder={"_target_": "x.LateFusion", "networks": {"rgb_static": dict(C.ENC, normalize_output=True)}}),
    C.cql_cfg(device=dev, goal_encoder=dict(C.GOAL_ENC, activation_function="Tanh")),
    C.cql_cfg(device=dev, actor=dict(C.ACTOR, policy={"_target_": "tacorl.networks.actor_critic.actor.D2RLPolicy"})),
    C.cql_cfg(device=dev, with_dr3=True),
    C.playlmp_cfg(device=dev, plan_recognition=dict(C.plan_recognition(16, 16), encoder_normalize=True)),
    C.playlmp_cfg(device=dev, perceptual_encoder={"_target_": "x.LateFusion", "networks": {"rgb_static": dict(C.ENC, latent_dim=64)}}),
    C.playlmp_cfg(device=dev, action_decoder=dict(C.action_decoder(16), policy_rnn_dropout_p=0.1)),
]
for i, cfg in enumerate(bad_cfgs):
    try:
        L.instantiate(cfg)
    except NotImplementedError:
        continue
    raise AssertionError(f"config {i} must raise NotImplementedError")
print("ALL OK")
