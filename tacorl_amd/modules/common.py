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


class GraphMixin:
    """hipGraph replay of the device side of a step, shared by the three module classes.  The host
    class provides `world_size` and (optionally) `_force_graph_split`."""

    _use_graph = False

    def enable_graph(self, on=True):
        """Replay the device side of the step from a captured hipGraph (one graph on a single GPU; with
        several ranks one graph per collective-free segment, the RCCL all-reduces stay eager between them)."""
        self._use_graph = bool(on)
        self._graphs = {}

    def load_state_dict(self, *args, **kwargs):
        """nn.Module.load_state_dict, then drop the captured graphs: a capture may have skipped weight-only
        preparation that was already valid at capture time (frozen networks), so new weights need new captures."""
        out = super().load_state_dict(*args, **kwargs)
        if getattr(self, "_graphs", None):
            self._graphs = {}
        return out

    def _run_segments(self, key, segs, collectives, side=None):
        """Run the device side of a step: `segs` are collective-free kernel sequences over fixed buffers,
        `collectives[i]` runs between segs[i] and segs[i+1] (RCCL all-reduces; no-ops on one GPU).
        Eager, or - with enable_graph() - each segment replayed from a captured hipGraph (one graph for
        the whole step on a single GPU; collectives always stay eager between graphs).
        side = (i, fn): work that depends only on segs[:i+1] and that nothing later in the step reads
        (TACORL's logging-only action-decoder pass).  In split mode it is its own graph, replayed on a side
        stream right after segment i and joined at the end of the step, so it overlaps the collectives and
        the remaining segments; otherwise it simply runs after segment i."""
        def eager():
            for i, f in enumerate(segs):
                f()
                if side is not None and side[0] == i:
                    side[1]()
                if i < len(collectives):
                    collectives[i]()

        if not self._use_graph:
            return eager()
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        from .. import ops
        gs = self._graphs.get(key)
        if gs is not None and gs[2] != ops.alloc_epoch():
            # some step buffer or workspace was (re)allocated since the capture (another batch shape came
            # through - the last, partial batch of an epoch - or another module grew a shared workspace): every
            # capture of this module may hold freed addresses.  Drop them all and capture again.
            self._graphs = {}
            gs = None
        if gs is None:
            eager()  # warm-up: sizes every workspace, so the capture allocates nothing
            torch.cuda.synchronize()
            split = getattr(self, "world_size", 1) > 1 or getattr(self, "_force_graph_split", False)
            if split:
                parts = [[f] for f in segs]
            else:
                parts = [list(segs)]
                if side is not None:
                    parts[0].insert(side[0] + 1, side[1])
            gs = []
            for part in parts:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for f in part:
                        f()
                gs.append(g)
            g_side = None
            if split and side is not None:
                g_side = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g_side):
                    side[1]()
                if getattr(self, "_side_replay_stream", None) is None:
                    self._side_replay_stream = torch.cuda.Stream(device=self.device)
            self._graphs[key] = (gs, g_side, ops.alloc_epoch())
            return
        gs, g_side, _ = gs
        stepped = getattr(self, "_stepped_blocks", None)
        if stepped is not None:  # a replay runs no python: the optimiser kernels' writes are announced here
            ops.touched(*stepped())
        cur = torch.cuda.current_stream()
        for i, g in enumerate(gs):
            g.replay()
            if g_side is not None and side[0] == i:
                self._side_replay_stream.wait_stream(cur)
                with torch.cuda.stream(self._side_replay_stream):
                    g_side.replay()
            if len(gs) > 1 and i < len(collectives):
                collectives[i]()
        if g_side is not None:
            cur.wait_stream(self._side_replay_stream)
