"""The PyTorch-Lightning side of the drop-in boundary (reference scripts/train.py:28-66).

The reference's module classes are `pl.LightningModule`s that `Trainer.fit` drives
(`modules/cql/cql_offline_lightning.py:24,115-116,481,553-574`, `modules/play_lmp/play_lmp_for_rl.py:17,
362-368`).  The classes in `tacorl_amd.modules` derive from `LightningModuleBase`:

* pytorch_lightning importable  -> `LightningModuleBase` IS `pl.LightningModule`: `Trainer.fit` accepts the
  module, `self.log` reaches PL's result collection, `save_hyperparameters` / checkpoints are PL's.
* not importable (this image, the GPU box) -> a small stand-in with the same surface, driven by
  `MiniTrainer` below; `instantiate` replaces `hydra.utils.instantiate` for `_recursive_: False` dict configs.

`configure_optimizers()` returns `BlockAdam`s: real `torch.optim.Optimizer` subclasses whose state IS the
engine's flat Adam blocks (`state_dict()` in torch-Adam format, round-trips through checkpoints).  Their
`step()` does not compute: the fused clip + Adam + Polyak kernels already ran inside `training_step`
(manual optimisation, as the reference: `self.automatic_optimization = False`, `:115`).
"""
import copy
import importlib
import inspect
import os

import torch
import torch.nn as nn

try:  # pragma: no cover - pytorch_lightning is not installed in the build image
    import pytorch_lightning as pl

    HAVE_PL = True
except ImportError:
    pl = None
    HAVE_PL = False


class _MiniLightningModule(nn.Module):
    """The slice of pl.LightningModule (1.5/1.6) that the reference modules and `Trainer.fit` use."""

    def __init__(self):
        super().__init__()
        self._trainer = None
        self.automatic_optimization = True
        self._hparams = {}

    # -- trainer-owned state
    @property
    def trainer(self):
        return self._trainer

    @trainer.setter
    def trainer(self, t):
        self._trainer = t

    @property
    def global_step(self):
        return self._trainer.global_step if self._trainer is not None else 0

    @property
    def hparams(self):
        return self._hparams

    def save_hyperparameters(self, *args, ignore=None, frame=None, **kwargs):
        """Constructor arguments -> self.hparams (what PL embeds in checkpoints): the calling __init__'s and,
        walking up the stack, those of the subclasses' __init__s that called it (PL's collect_init_args)."""
        fr = frame or inspect.currentframe().f_back
        hp = {}
        while fr is not None and fr.f_code.co_name == "__init__" and fr.f_locals.get("self") is self:
            av = inspect.getargvalues(fr)
            cur = {k: av.locals[k] for k in av.args if k != "self"}
            if av.keywords:
                cur.update(av.locals[av.keywords])
            for k, v in cur.items():
                hp.setdefault(k, v)
            fr = fr.f_back
        drop = set(ignore or [])
        self._hparams = {k: v for k, v in hp.items() if k not in drop and not isinstance(v, nn.Module)}

    def log(self, name, value, **kw):
        if self._trainer is not None:
            self._trainer._log(name, float(value), **kw)

    def log_dict(self, d, **kw):
        for k, v in d.items():
            self.log(k, v, **kw)

    def optimizers(self):
        opts = self._trainer.optimizers if self._trainer is not None else _as_list(self.configure_optimizers())
        return opts if len(opts) != 1 else opts[0]

    def manual_backward(self, loss, *a, **kw):
        loss.backward(*a, **kw)

    # -- hooks (no-ops)
    def on_fit_start(self):
        pass

    def on_fit_end(self):
        pass

    def on_train_start(self):
        pass

    def on_train_epoch_start(self):
        pass

    def on_train_epoch_end(self):
        pass

    def on_train_batch_start(self, batch, batch_idx, unused=0):
        pass

    def on_train_batch_end(self, outputs, batch, batch_idx, unused=0):
        pass

    def on_validation_epoch_start(self):
        pass

    def on_validation_epoch_end(self):
        pass

    def on_save_checkpoint(self, checkpoint):
        pass

    def on_load_checkpoint(self, checkpoint):
        pass

    def transfer_batch_to_device(self, batch, device, dataloader_idx=0):
        return move_to(batch, device)


LightningModuleBase = pl.LightningModule if HAVE_PL else _MiniLightningModule


def _as_list(x):
    return list(x) if isinstance(x, (list, tuple)) else [x]


def move_to(x, device):
    if isinstance(x, dict):
        return {k: move_to(v, device) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(move_to(v, device) for v in x)
    return x.to(device, non_blocking=True) if torch.is_tensor(x) else x


def default_device():
    """One process per GPU: PL's DDP launcher and torch.distributed.run export LOCAL_RANK."""
    return torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}")


# ----------------------------------------------------------------------------------- optimiser
class BlockAdam(torch.optim.Optimizer):
    """torch.optim.Optimizer view of flat Adam blocks (param / m / v / step live in the engine).

    `entries`: [(blk, {name: nn.Parameter}, {name: m_view}, {name: v_view})] - the parameters are views into
    blk.param.  `state_dict()` has torch.optim.Adam's layout (per parameter `step`, `exp_avg`, `exp_avg_sq`);
    `load_state_dict()` copies into the blocks.  `step()` runs the optional closure and nothing else: the
    update is part of the module's training_step kernels (clip + Adam + Polyak, rl_ops.hip adam_kernel)."""

    def __init__(self, name, entries, lr):
        self.name = name
        self._entries = entries
        params = [p for _, ps, _, _ in entries for p in ps.values() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False))

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def step(self, closure=None):
        return closure() if closure is not None else None

    def zero_grad(self, set_to_none=False):
        pass  # the backward kernels overwrite the gradient blocks

    def _triples(self):
        for blk, ps, ms, vs in self._entries:
            for n, p in ps.items():
                if p.requires_grad:
                    yield blk, p, ms[n], vs[n]

    def state_dict(self):
        state = {}
        for i, (blk, _, m, v) in enumerate(self._triples()):
            state[i] = {"step": blk.step.detach().to(torch.float32).reshape(()).clone(), "exp_avg": m.detach().clone(),
                        "exp_avg_sq": v.detach().clone()}
        groups = [{**{k: v for k, v in g.items() if k != "params"}, "params": list(range(len(state)))}
                  for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        tri = list(self._triples())
        st = sd["state"]
        if st and len(st) != len(tri):
            raise ValueError(f"optimizer {self.name}: {len(st)} parameter states for {len(tri)} parameters")
        with torch.no_grad():
            seen = set()
            for i, (blk, _, m, v) in enumerate(tri):
                s = st.get(i, st.get(str(i))) if st else None
                if s is None:
                    continue
                m.copy_(s["exp_avg"].to(m.device))
                v.copy_(s["exp_avg_sq"].to(v.device))
                if id(blk) not in seen:
                    blk.step.fill_(int(float(s["step"])))
                    seen.add(id(blk))
        for g, gs in zip(self.param_groups, sd.get("param_groups", [])):
            g["lr"] = gs.get("lr", g["lr"])


# ------------------------------------------------------------------------ config instantiation
def instantiate(cfg, *args, **kwargs):
    """`hydra.utils.instantiate` for the reference's `_recursive_: False` configs (every module / network YAML sets it,
    e.g. config/module/tacorl.yaml:7): import `_target_`, call it with the remaining keys + kwargs, leave nested
    configs as plain dicts (children instantiate themselves).  A config without `_target_` is returned as it is."""
    if cfg is None or "_target_" not in cfg:
        return cfg
    cfg = copy.deepcopy(dict(cfg))
    target = cfg.pop("_target_")
    if cfg.pop("_recursive_", False):
        raise NotImplementedError("instantiate: only _recursive_: False configs (as the reference's) are supported")
    cfg.pop("_convert_", None)
    mod, _, name = target.rpartition(".")
    fn = getattr(importlib.import_module(mod), name)
    cfg.update(kwargs)
    return fn(*args, **cfg)


# ------------------------------------------------------------------------------------ trainer
class MiniTrainer:
    """Minimal fit loop for boxes without pytorch_lightning (same call order as PL's: configure_optimizers,
    on_fit_start, per batch on_train_batch_start -> training_step(batch, batch_idx) -> on_train_batch_end,
    validation at epoch end, checkpoints in PL's layout).  Enforces what PL enforces on the way in."""

    def __init__(self, max_epochs=1, max_steps=-1, log_every_n_steps=50, limit_val_batches=None, default_root_dir=None,
                 callbacks=None, logger=None, **unused):
        self.max_epochs, self.max_steps, self.log_every_n_steps = max_epochs, max_steps, log_every_n_steps
        self.limit_val_batches = limit_val_batches
        self.default_root_dir = default_root_dir
        self.callbacks = list(callbacks or [])
        self.logger = logger
        self.current_epoch, self.global_step = 0, 0
        self.world_size = 1
        self.optimizers = []
        self.logged_metrics = {}
        self.model = None

    def _log(self, name, value, **kw):
        self.logged_metrics[name] = value

    @staticmethod
    def _validate(model):
        if not isinstance(model, LightningModuleBase):
            raise TypeError(f"`Trainer.fit()` requires a `LightningModule`, got: {type(model).__qualname__}")
        for fn in ("training_step", "configure_optimizers"):
            if getattr(type(model), fn, None) is getattr(LightningModuleBase, fn, None):
                raise RuntimeError(f"No `{fn}()` method defined. Lightning `Trainer` expects it")

    def _attach(self, model):
        self._validate(model)
        self.model = model
        model.trainer = self
        import torch.distributed as dist

        self.world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        opts = model.configure_optimizers()
        opts = _as_list(opts[0] if isinstance(opts, tuple) and len(opts) == 2 and isinstance(opts[0], list) else opts)
        for o in opts:
            if not isinstance(o, torch.optim.Optimizer):
                raise TypeError(f"configure_optimizers() must return torch.optim.Optimizer instances, got {type(o).__qualname__}")
        self.optimizers = opts

    def fit(self, model, train_dataloaders=None, val_dataloaders=None, datamodule=None, ckpt_path=None):
        self._attach(model)
        if datamodule is not None:
            train_dataloaders = datamodule.train_dataloader()
            val_dataloaders = datamodule.val_dataloader() if hasattr(datamodule, "val_dataloader") else None
        if ckpt_path is not None:
            self.load_checkpoint(ckpt_path)
        model.train()
        model.on_fit_start()
        model.on_train_start()
        done = False
        while self.current_epoch < self.max_epochs and not done:
            model.on_train_epoch_start()
            for batch_idx, batch in enumerate(train_dataloaders):
                batch = model.transfer_batch_to_device(batch, model.device, 0)
                model.on_train_batch_start(batch, batch_idx)
                out = model.training_step(batch, batch_idx)
                if model.automatic_optimization:
                    raise RuntimeError("MiniTrainer drives manual-optimisation modules only (the in-scope ones all are)")
                model.on_train_batch_end(out, batch, batch_idx)
                self.global_step += 1
                if 0 < self.max_steps <= self.global_step:
                    done = True
                    break
            model.on_train_epoch_end()
            if val_dataloaders is not None:
                model.on_validation_epoch_start()
                for batch_idx, batch in enumerate(val_dataloaders):
                    if self.limit_val_batches is not None and batch_idx >= self.limit_val_batches:
                        break
                    model.validation_step(model.transfer_batch_to_device(batch, model.device, 0), batch_idx)
                model.on_validation_epoch_end()
            self.current_epoch += 1
        model.on_fit_end()

    # -- checkpoints (PL's dictionary layout: scripts/train.py:47-66 resumes from <dirpath>/last.ckpt)
    def dump_checkpoint(self):
        m = self.model
        ck = {"epoch": self.current_epoch, "global_step": self.global_step, "state_dict": m.state_dict(),
              "optimizer_states": [o.state_dict() for o in self.optimizers], "lr_schedulers": [],
              "hyper_parameters": dict(getattr(m, "hparams", {}) or {})}
        m.on_save_checkpoint(ck)
        return ck

    def save_checkpoint(self, path):
        torch.save(self.dump_checkpoint(), path)

    def load_checkpoint(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.model.on_load_checkpoint(ck)
        self.model.load_state_dict(ck["state_dict"])
        for o, s in zip(self.optimizers, ck.get("optimizer_states", [])):
            o.load_state_dict(s)
        self.current_epoch, self.global_step = ck.get("epoch", 0), ck.get("global_step", 0)
