"""TEST INFRASTRUCTURE: a strict stand-in for pytorch_lightning 1.5/1.6 (not installed in this image).

It enforces the checks `Trainer.fit` makes on the way in (PL 1.6 `ConfigValidator`, `Trainer._run`,
`LightningModule` properties) so that `tests/test_lightning_*.py` can verify that the tacorl_amd module classes are
acceptable LightningModules: read-only `current_epoch` / `global_step` / `device` properties on the base class,
`self.log` only legal inside a trainer-run hook and only with PL's keyword arguments, `configure_optimizers`
must return `torch.optim.Optimizer`s, `model.to(device)` is called, hooks are called with PL's signatures, and the
checkpoint dictionary has PL's keys.  Put `tests/fake_pl` on PYTHONPATH *before* importing tacorl_amd."""
import inspect

import torch
import torch.nn as nn
from torch.optim import Optimizer

__version__ = "1.6.5-standin"
_LOG_KW = {"prog_bar", "logger", "on_step", "on_epoch", "reduce_fx", "enable_graph", "sync_dist", "sync_dist_group",
           "add_dataloader_idx", "batch_size", "metric_attribute", "rank_zero_only"}


class MisconfigurationException(Exception):
    pass


class Callback:
    def on_fit_start(self, trainer, pl_module):
        pass

    def on_train_batch_start(self, trainer, pl_module, batch, batch_idx, unused=0):
        pass

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, unused=0):
        pass

    def on_train_epoch_end(self, trainer, pl_module):
        pass


class LightningDataModule:
    pass


class LightningModule(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._trainer = None
        self._automatic_optimization = True
        self._hparams = {}
        self._current_fx_name = None
        self._device = torch.device("cpu")

    # ---- read-only state, as in PL
    @property
    def trainer(self):
        return self._trainer

    @trainer.setter
    def trainer(self, t):
        self._trainer = t

    @property
    def current_epoch(self):
        return self._trainer.current_epoch if self._trainer else 0

    @property
    def global_step(self):
        return self._trainer.global_step if self._trainer else 0

    @property
    def device(self):
        return self._device

    @property
    def automatic_optimization(self):
        return self._automatic_optimization

    @automatic_optimization.setter
    def automatic_optimization(self, v):
        self._automatic_optimization = bool(v)

    @property
    def hparams(self):
        return self._hparams

    def save_hyperparameters(self, *args, ignore=None, frame=None, logger=True):
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
        for k in (ignore or []):
            hp.pop(k, None)
        self._hparams = hp

    def to(self, *args, **kwargs):
        dev = torch._C._nn._parse_to(*args, **kwargs)[0]
        if dev is not None:
            self._device = dev
        return super().to(*args, **kwargs)

    def log(self, name, value, **kw):
        if self._trainer is None or self._current_fx_name is None:
            raise MisconfigurationException("You are trying to `self.log()` but the loop's result collection is not registered yet")
        bad = set(kw) - _LOG_KW
        if bad:
            raise TypeError(f"log() got unexpected keyword arguments {sorted(bad)}")
        if not isinstance(value, (int, float, torch.Tensor)):
            raise ValueError(f"`self.log({name}, {value})` was called, but `{type(value).__name__}` values cannot be logged")
        self._trainer._results[name] = float(value)

    def log_dict(self, d, **kw):
        for k, v in d.items():
            self.log(k, v, **kw)

    def optimizers(self, use_pl_optimizer=True):
        o = self._trainer.optimizers
        if use_pl_optimizer:
            o = [LightningOptimizer(x, self._trainer) for x in o]
        return o[0] if len(o) == 1 else o

    def manual_backward(self, loss, *args, **kwargs):
        if self.automatic_optimization:
            raise MisconfigurationException("manual_backward with automatic optimisation")
        loss.backward(*args, **kwargs)

    def training_step(self, *args, **kwargs):
        raise NotImplementedError

    def configure_optimizers(self):
        raise NotImplementedError

    # hooks
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
        def mv(x):
            if isinstance(x, dict):
                return {k: mv(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return type(x)(mv(v) for v in x)
            return x.to(device) if torch.is_tensor(x) else x
        return mv(batch)


def _is_overridden(name, model):
    return getattr(type(model), name) is not getattr(LightningModule, name)


class LightningOptimizer:
    """PL 1.6: what `LightningModule.optimizers()` hands out; its step() is where `trainer.global_step` advances
    (manual optimisation: loops/optimization/manual_loop.py, optim_step_progress)."""

    def __init__(self, optimizer, trainer):
        self._optimizer, self._trainer = optimizer, trainer

    def __getattr__(self, name):
        return getattr(self._optimizer, name)

    def step(self, closure=None, **kw):
        out = self._optimizer.step(closure) if closure is not None else self._optimizer.step()
        self._trainer.global_step += 1
        return out


class Trainer:
    def __init__(self, max_epochs=1, max_steps=-1, gpus=None, devices=None, accelerator=None, strategy=None, precision=32,
                 log_every_n_steps=50, callbacks=None, logger=None, gradient_clip_val=None, limit_val_batches=None,
                 default_root_dir=None, **kw):
        self.max_epochs, self.max_steps, self.log_every_n_steps = max_epochs, max_steps, log_every_n_steps
        self.callbacks = list(callbacks or [])
        self.gradient_clip_val = gradient_clip_val
        self.limit_val_batches = limit_val_batches
        self.root_device = torch.device("cuda:0" if (gpus or devices) and torch.cuda.is_available() else "cpu")
        self.current_epoch = self.global_step = 0
        self.world_size = 1
        self.optimizers, self._results, self.logged_metrics = [], {}, {}
        self.lightning_module = None

    def _verify(self, model):
        if not isinstance(model, LightningModule):
            raise TypeError(f"`Trainer.fit()` requires a `LightningModule`, got: {model.__class__.__qualname__}")
        if not _is_overridden("training_step", model):
            raise MisconfigurationException("No `training_step()` method defined. Lightning `Trainer` expects as minimum a `training_step()`")
        if not _is_overridden("configure_optimizers", model):
            raise MisconfigurationException("No `configure_optimizers()` method defined.")
        if not model.automatic_optimization and self.gradient_clip_val:
            raise MisconfigurationException("Automatic gradient clipping is not supported for manual optimization.")

    def _init_optimizers(self, model):
        o = model.configure_optimizers()
        if isinstance(o, Optimizer):
            o = [o]
        elif isinstance(o, (list, tuple)) and all(isinstance(x, Optimizer) for x in o):
            o = list(o)
        else:
            raise MisconfigurationException("Unknown configuration for model optimizers. Output from `model.configure_optimizers()` "
                                            "should be one of: `Optimizer`, `List[Optimizer]`, ...")
        self.optimizers = o

    def _call(self, model, hook, *args):
        prev, model._current_fx_name = model._current_fx_name, hook
        try:
            return getattr(model, hook)(*args)
        finally:
            model._current_fx_name = prev

    def fit(self, model, train_dataloaders=None, val_dataloaders=None, datamodule=None, ckpt_path=None):
        self._verify(model)
        self.lightning_module = model
        model.trainer = self
        if datamodule is not None:
            train_dataloaders = datamodule.train_dataloader()
            val_dataloaders = datamodule.val_dataloader() if hasattr(datamodule, "val_dataloader") else None
        model.to(self.root_device)  # PL moves the module to its device before configure_optimizers
        self._init_optimizers(model)
        if ckpt_path is not None:
            self._restore(ckpt_path)
        for cb in self.callbacks:
            cb.on_fit_start(self, model)
        self._call(model, "on_fit_start")
        model.train()
        self._call(model, "on_train_start")
        stop = False
        while self.current_epoch < self.max_epochs and not stop:
            self._call(model, "on_train_epoch_start")
            for batch_idx, batch in enumerate(train_dataloaders or []):
                batch = model.transfer_batch_to_device(batch, self.root_device, 0)
                for cb in self.callbacks:
                    cb.on_train_batch_start(self, model, batch, batch_idx)
                if self._call(model, "on_train_batch_start", batch, batch_idx) == -1:
                    break
                kw = [batch, batch_idx]  # PL passes (batch, batch_idx) positionally
                if len(self.optimizers) > 1 and model.automatic_optimization:
                    kw.append(0)  # optimizer_idx
                out = self._call(model, "training_step", *kw)
                if model.automatic_optimization:
                    raise MisconfigurationException("stand-in Trainer: automatic optimisation not modelled (the in-scope modules are manual)")
                self._call(model, "on_train_batch_end", out, batch, batch_idx)
                for cb in self.callbacks:
                    cb.on_train_batch_end(self, model, out, batch, batch_idx)
                # (global_step advances in LightningOptimizer.step(), as in PL 1.6 - a module that never steps its
                # optimizers never reaches max_steps and is never checkpointed every_n_train_steps)
                if self.global_step % self.log_every_n_steps == 0:
                    self.logged_metrics.update(self._results)
                if 0 < self.max_steps <= self.global_step:
                    stop = True
                    break
            self._call(model, "on_train_epoch_end")
            for cb in self.callbacks:
                cb.on_train_epoch_end(self, model)
            if val_dataloaders is not None:
                model.eval()
                self._call(model, "on_validation_epoch_start")
                for batch_idx, batch in enumerate(val_dataloaders):
                    if self.limit_val_batches is not None and batch_idx >= self.limit_val_batches:
                        break
                    self._call(model, "validation_step", model.transfer_batch_to_device(batch, self.root_device, 0), batch_idx)
                self._call(model, "on_validation_epoch_end")
                model.train()
            self.current_epoch += 1
        self._call(model, "on_fit_end")
        self.logged_metrics.update(self._results)
        model.cpu()  # strategy teardown (PL 1.5 / 1.6: lightning_module.cpu() after fit on an accelerator)

    # ---- checkpoints: PL's dictionary
    def save_checkpoint(self, path):
        m = self.lightning_module
        ck = {"epoch": self.current_epoch, "global_step": self.global_step, "pytorch-lightning_version": __version__,
              "state_dict": m.state_dict(), "optimizer_states": [o.state_dict() for o in self.optimizers],
              "lr_schedulers": [], "hyper_parameters": dict(m.hparams), "callbacks": {}}
        m.on_save_checkpoint(ck)
        torch.save(ck, path)

    def _restore(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=False)
        m = self.lightning_module
        m.on_load_checkpoint(ck)
        m.load_state_dict(ck["state_dict"])
        for o, s in zip(self.optimizers, ck["optimizer_states"]):
            o.load_state_dict(s)
        self.current_epoch, self.global_step = ck["epoch"], ck["global_step"]
