"""Shared plumbing for the reference-shaped module classes (no arithmetic here)."""
import copy

import torch
import torch.nn as nn

from .._lib import BF16, F32


def to_plain(cfg):
    """OmegaConf.to_container(cfg, resolve=True) when omegaconf is around, else a deep copy."""
    try:
        from omegaconf import OmegaConf  # noqa: WPS433

        if OmegaConf.is_config(cfg):
            return OmegaConf.to_container(cfg, resolve=True)
    except ImportError:
        pass
    return copy.deepcopy(dict(cfg)) if isinstance(cfg, dict) else copy.deepcopy(cfg)


def compute_flag(compute_dtype):
    if compute_dtype in (F32, "f32", "fp32", torch.float32, 32, "32"):
        return F32
    if compute_dtype in (BF16, "bf16", torch.bfloat16, 16, "16"):
        return BF16
    raise ValueError(f"compute_dtype {compute_dtype!r}: use 'f32' or 'bf16'")


def register_views(root, prefix, views, requires_grad=True):
    """Expose flat-block views as nn.Parameters under the reference's dotted names."""
    for name, t in views.items():
        parts = (prefix + name).split(".")
        mod = root
        for p in parts[:-1]:
            if not hasattr(mod, p) or not isinstance(getattr(mod, p), nn.Module):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], nn.Parameter(t, requires_grad=requires_grad))


class LoggerMixin:
    """`self.log(name, value, ...)` as LightningModule offers it; values are kept in
    `self.logged` (and forwarded to a trainer-provided sink when one is attached)."""

    def log(self, name, value, **kw):
        self.logged[name] = float(value)
        sink = getattr(self, "_log_sink", None)
        if sink is not None:
            sink(name, float(value), **kw)
