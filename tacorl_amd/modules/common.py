"""Shared plumbing for the reference-shaped module classes (no arithmetic here)."""
import copy
import re

import os

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


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def load_resolved_yaml(path):
    """OmegaConf.load(path) + OmegaConf.to_container(cfg, resolve=True) (reference utils/networks.py:142): the saved
    run config keeps `${latent_plan_dim}` / `${datamodule.dataset.max_window_size}` style interpolations
    (config/networks/plan_recognition/transformer.yaml:7,13).  With omegaconf installed it does the work; without it
    the absolute-key `${a.b.c}` interpolations those configs use are resolved here (resolvers such as `${oc.env:..}`
    are not supported and raise)."""
    try:
        from omegaconf import OmegaConf  # noqa: WPS433

        return OmegaConf.to_container(OmegaConf.load(path), resolve=True)
    except ImportError:
        pass
    import yaml

    with open(path) as f:
        root = yaml.safe_load(f)

    def lookup(key, stack):
        if ":" in key or key.startswith("."):
            raise NotImplementedError(f"{path}: interpolation ${{{key}}} needs omegaconf")
        if key in stack:
            raise ValueError(f"{path}: interpolation cycle at ${{{key}}}")
        node = root
        for part in key.split("."):
            node = node[int(part)] if isinstance(node, list) else node[part]
        return walk(node, stack + (key,))

    def walk(node, stack=()):
        if isinstance(node, dict):
            return {k: walk(v, stack) for k, v in node.items()}
        if isinstance(node, list):
            return [walk(v, stack) for v in node]
        if isinstance(node, str):
            m = _INTERP.fullmatch(node)
            if m:
                return lookup(m.group(1).strip(), stack)
            while _INTERP.search(node):
                node = _INTERP.sub(lambda mm: str(lookup(mm.group(1).strip(), stack)), node)
        return node

    return walk(root)


def find_checkpoint(run_dir, epoch=-1):
    """get_checkpoint_i_from_dir (reference utils/networks.py:120-136): `last.ckpt` for epoch -1, else the checkpoint
    whose file name carries exactly `epoch_<N>` / `epoch=<N>`; otherwise the `epoch`-th checkpoint by modification time
    (the reference sorts by mtime and discards the result - an unsorted rglob order; here the sort is kept)."""
    from pathlib import Path

    cks = sorted(Path(run_dir).rglob("*.ckpt"), key=lambda f: f.stat().st_mtime)
    if not cks:
        raise FileNotFoundError(f"no .ckpt under {run_dir}")
    if epoch == -1:
        for c in cks:
            if c.stem == "last":
                return c
    for c in cks:
        if any(int(n) == epoch for n in re.findall(r"epoch[_=](\d+)", c.name)):
            return c
    return cks[epoch]


def compute_flag(compute_dtype):
    if compute_dtype in (F32, "f32", "fp32", torch.float32, 32, "32"):
        return F32
    if compute_dtype in (BF16, "bf16", torch.bfloat16, 16, "16"):
        return BF16
    raise ValueError(f"compute_dtype {compute_dtype!r}: use 'f32' or 'bf16'")


def register_views(root, prefix, views, requires_grad=True):
    """Expose flat-block views as nn.Parameters under the reference's dotted names.
    Returns {view name: Parameter} (what BlockAdam is built from)."""
    made = {}
    for name, t in views.items():
        parts = (prefix + name).split(".")
        mod = root
        for p in parts[:-1]:
            if not hasattr(mod, p) or not isinstance(getattr(mod, p), nn.Module):
                mod.add_module(p, nn.Module())
            mod = getattr(mod, p)
        made[name] = nn.Parameter(t, requires_grad=requires_grad)
        mod.register_parameter(parts[-1], made[name])
    return made


def broadcast_blocks(blocks, src=0):
    """Make every rank's parameter blocks and optimiser state equal to rank `src`'s (what DDP does when it wraps
    the reference module, so callers that seed per rank still start from one model).  No-op without a
    process group."""
    import torch.distributed as dist

    from .. import dist as D

    if not D.group_ready() or not D.collectives_on(dist.get_world_size()):
        return
    for blk in blocks:
        for name in ("param", "m", "v", "step"):
            t = getattr(blk, name, None)
            if t is not None:
                dist.broadcast(t, src)


class ModuleMixin:
    """What the three module classes share on top of LightningModuleBase (pl.LightningModule when
    pytorch_lightning is importable, tacorl_amd.lightning._MiniLightningModule otherwise): the parameter
    blocks live on ONE device chosen at construction, `self.log` also keeps the last value in `self.logged`,
    `current_epoch` can be pinned by hand (tests / scripts without a trainer), optimisers are BlockAdams."""

    def _init_runtime(self, device, compute_dtype, image_dtype, world_size):
        from .. import _lib
        from ..lightning import default_device

        _lib.lib()  # fail loudly, now, if the HIP extension is missing (no CPU / eager fallback)
        self.dev = torch.device(device) if device is not None else default_device()
        self.logged = {}
        self.compute = compute_flag(compute_dtype)
        self.img_dtype = torch.bfloat16 if compute_flag(image_dtype) == BF16 else torch.float32
        self.world_size = int(world_size)
        self.log_every_n_steps = 1  # PL Trainer(log_every_n_steps=...) semantics: metrics are read back (one D2H
        self._step_count = 0        # sync) only on these steps
        self._graphs, self._use_graph = {}, False

    # -- trainer-owned state, overridable without a trainer
    def _attached_trainer(self):
        tr = self.__dict__.get("_trainer")
        if tr is None:
            tr = getattr(self, "_trainer", None)
        return tr

    @property
    def current_epoch(self):
        ov = self.__dict__.get("_epoch_override")
        if ov is not None:
            return ov
        tr = self._attached_trainer()
        return int(tr.current_epoch) if tr is not None else 0

    @current_epoch.setter
    def current_epoch(self, v):
        self.__dict__["_epoch_override"] = v

    @property
    def device(self):
        return self.dev

    def log(self, name, value, **kw):
        self.logged[name] = float(value)
        if self._attached_trainer() is not None:
            super().log(name, float(value), **kw)

    def _apply(self, fn, *args, **kwargs):
        """nn.Module.to / .cuda / .float land here.  The parameters are views into flat device blocks that the
        kernels address by raw pointer, so the module cannot be moved or cast after construction; a call that
        would not change anything (what Trainer.fit does to a module that is already on its GPU) is accepted."""
        probe = fn(torch.empty(0, device=self.dev))
        if probe.device.type == "cpu" and probe.dtype == torch.float32:
            # pytorch_lightning 1.5 / 1.6 end Trainer.fit with `lightning_module.cpu()` (strategy teardown).  The blocks
            # stay where they are - the module remains usable on its GPU and state_dict() / checkpoints are unaffected.
            return self
        if probe.device != self.dev or probe.dtype != torch.float32:
            raise RuntimeError(
                f"{type(self).__name__} lives on {self.dev} as fp32 blocks (views, raw pointers): build it on its GPU "
                "(device=..., default cuda:$LOCAL_RANK) and select precision with compute_dtype / image_dtype; "
                f".to({probe.device}, {probe.dtype}) cannot move or cast it")
        return self

    def _tick_optimizers(self):
        """Under a pytorch_lightning Trainer: one `.step()` per optimizer per training_step, as the reference's manual
        optimisation does.  BlockAdam.step() itself does nothing (the update is part of the step's kernels), but PL >= 1.6
        counts `trainer.global_step` - max_steps, every_n_train_steps checkpoints, the logger's x axis - in
        LightningOptimizer.step() calls."""
        if self._attached_trainer() is None:
            return
        opts = self.optimizers()
        for o in (opts if isinstance(opts, (list, tuple)) else [opts]):
            o.step()

    def on_fit_start(self):
        """Adopt the trainer's world size (PL DDP: one process per GPU; the gradient blocks are all-reduced
        through torch.distributed's default group - RCCL with backend nccl)."""
        tr = self._attached_trainer()
        ws = int(getattr(tr, "world_size", 1) or 1)
        if ws != self.world_size:
            self._set_world_size(ws)

    def _set_world_size(self, ws):
        self.world_size = ws
        if getattr(self, "engine", None) is not None:
            self.engine.world = ws
        self._graphs = {}
        if ws > 1:
            self.sync_from_rank0()

    def _make_adam(self, name, entries, lr):
        from ..lightning import BlockAdam

        return BlockAdam(name, entries, lr)


class GraphMixin:
    """hipGraph replay of the device side of a step, shared by the three module classes.  The host
    class provides `world_size` and (optionally) `_force_graph_split`."""

    _use_graph = False

    def _collectives_in_graph(self):
        from .. import dist as D

        return D.graph_collectives() and D.collectives_on(getattr(self, "world_size", 1))

    def _segmented(self):
        """Is the device side of a step run as collective-free segments with eager all-reduces between them?  Yes with
        several ranks (or TACORL_FORCE_COLLECTIVES=1 on one) unless the collectives are captured into the step's graph;
        `_force_graph_split` is the test switch that splits without any collective."""
        from .. import dist as D

        if getattr(self, "_force_graph_split", False):
            return True
        return D.collectives_on(getattr(self, "world_size", 1)) and not D.graph_collectives()

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
        side = (i, fn) or (i, fn, j): work that depends only on segs[:i+1] and that nothing later in the step reads
        (TACORL's logging-only action-decoder pass) - or that only collectives[j] and what follows read (the fine-tuned
        decoder's loss + backward: its gradient block is part of the arena all-reduce #2 sums).  In split mode it is its own
        graph, replayed on a side stream right after segment i and joined at the end of the step (in front of collectives[j]),
        so it overlaps the collectives and the remaining segments; otherwise it simply runs after segment i."""
        side_join = side[2] if side is not None and len(side) > 2 else None
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
        base_key = key
        # the collective form is part of what a capture froze: a graph captured without collectives, or as
        # one graph, must not be replayed after TACORL_FORCE_COLLECTIVES / TACORL_GRAPH_COLLECTIVES / _force_graph_split moved
        key = (base_key, self._segmented(), self._collectives_in_graph())
        gs = self._graphs.get(key)
        if gs is not None:  # least-recently-used order: a hit moves the key to the end
            self._graphs[key] = self._graphs.pop(key)
        stale = getattr(self, "_derived_stale", None)
        if gs is not None and stale is not None and stale():
            # the captured step relies on weight-derived buffers that the previous step's optimiser launch left current
            # (bf16 mirrors written by the Adam kernel); something else has written the parameters since - a user's
            # in-place edit, a broadcast - so run the step in python once more (it refreshes what is stale) and re-capture
            self._graphs = {}
            gs = None
        if gs is not None and gs[2] != ops.alloc_epoch():
            # some step buffer or workspace was (re)allocated since the capture (another batch shape came
            # through - the last, partial batch of an epoch - or another module grew a shared workspace): every
            # capture of this module may hold freed addresses.  Drop them all and capture again.
            self._graphs = {}
            gs = None
        if gs is None:
            with ops.capture_lock:  # no feeder thread prepares a batch (allocations, event waits) beside the capture
                self._capture(base_key, segs, collectives, side, eager)
            return
        gs, g_side, _ = gs
        stepped = getattr(self, "_stepped_blocks", None)
        if stepped is not None:  # a replay runs no python: the optimiser kernels' writes are announced here
            ops.touched(*stepped())
            after = getattr(self, "_after_replay_touch", None)
            if after is not None:
                after()
        cur = torch.cuda.current_stream()
        for i, g in enumerate(gs):
            g.replay()
            if g_side is not None and side[0] == i:
                self._side_replay_stream.wait_stream(cur)
                with torch.cuda.stream(self._side_replay_stream):
                    g_side.replay()
            if len(gs) > 1 and i < len(collectives):
                if g_side is not None and side_join == i:
                    cur.wait_stream(self._side_replay_stream)
                collectives[i]()
        if g_side is not None:
            cur.wait_stream(self._side_replay_stream)

    def _capture(self, key, segs, collectives, side, eager):
        """One eager pass (sizes every workspace, so the capture allocates nothing; it IS this call's step), then the
        capture(s) of the step for later calls."""
        eager()  # warm-up: sizes every workspace, so the capture allocates nothing
        torch.cuda.synchronize()
        self._capture_only(key, segs, collectives, side)

    def _capture_only(self, key, segs, collectives, side):
        from .. import dist as D
        from .. import ops

        split = self._segmented()
        in_graph = (not split) and self._collectives_in_graph()
        if split:
            parts = [[f] for f in segs]
        else:
            # one graph for the whole step; with TACORL_GRAPH_COLLECTIVES=1 on several ranks the all-reduces are
            # captured between the segments as nodes of that graph (RCCL kernels on the capturing stream)
            parts = [[]]
            for i, f in enumerate(segs):
                parts[0].append(f)
                if side is not None and side[0] == i:
                    parts[0].append(side[1])
                if i < len(collectives) and in_graph:
                    parts[0].append(collectives[i])
        gs, failed = [], None
        try:
            for part in parts:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    for f in part:
                        f()
                gs.append(g)
        except Exception as e:  # noqa: BLE001
            if not in_graph:
                raise
            failed = e
        if in_graph and D.any_rank(failed is not None, self.device):
            # The capture of a collective was refused on some rank (every rank asks, so every rank lands here together):
            # drop it everywhere and capture the step again as collective-free segments with eager all-reduces between
            # them - the form the N-rank tests run.  (A capture that *hangs* cannot be caught in-process: that is what
            # bench.py's supervisor and a Trainer's own timeout are for.)
            del gs
            D.refuse_graph_collectives(f"capture failed ({type(failed).__name__}: {str(failed)[:160]})" if failed
                                       else "capture failed on another rank")
            torch.cuda.synchronize()
            return self._capture_only(key, segs, collectives, side)
        g_side = None
        if split and side is not None:
            g_side = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_side, capture_error_mode="thread_local"):
                side[1]()
            if getattr(self, "_side_replay_stream", None) is None:
                self._side_replay_stream = torch.cuda.Stream(device=self.device)
        key = (key, split, in_graph)
        self._graphs[key] = (gs, g_side, ops.alloc_epoch())
        # A long job alternates keys (train / validation, the epoch's last partial batch, the BC -> Q phase switch):
        # keep the most recently used few and release the rest - dozens of live instantiated graphs in one process
        # end in a hipGraphLaunch crash on ROCm 7.2, and each one pins its workspaces' addresses.
        cap = max(1, int(os.environ.get("TACORL_MAX_GRAPHS", "6")))
        if len(self._graphs) > cap:
            torch.cuda.synchronize()  # nothing may still be replaying a graph that is about to be destroyed
            while len(self._graphs) > cap:
                del self._graphs[next(iter(self._graphs))]
