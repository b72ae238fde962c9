"""Deterministic synthetic replay batches and parameter values.

Everything here is driven by numpy's legacy ``RandomState`` (bit-stable across
machines), so a golden fixture only has to store a seed, never image data or
weights.  The batch schemas are the reference's:

* play window batch   - PlayDataset.__getitem__  (reference
  src/tacorl/datamodule/dataset/play_dataset.py:115-169): ``states[cam]``
  (B,T,3,H,W) f32 in [-1,1], ``actions`` (B,T,7), ``goal[cam]`` (B,3,H,W),
  ``disp`` (B,) int (geometric k, or -1).
* goal-cond transition - GoalCondReplayBufferDataset.get_transition
  (goal_cond_replay_buffer_dataset.py:275-295).

Image values are U(-1,1): the range after ``Normalize(0.5, 0.5)`` in
config/datamodule/transform_manager/transforms/rl_train.yaml:12-14.
"""
import zlib

import numpy as np
import torch


def _rs(seed, tag):
    return np.random.RandomState((zlib.crc32(tag.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)


def _img(rs, *shape):
    return torch.from_numpy(rs.uniform(-1.0, 1.0, size=shape).astype(np.float32))


def _actions(rs, *lead):
    a = rs.uniform(-1.0, 1.0, size=(*lead, 7)).astype(np.float32)
    a[..., -1] = np.where(rs.uniform(size=lead) < 0.5, -1.0, 1.0)
    return torch.from_numpy(a)


def _disp(rs, B):
    d = rs.geometric(0.3, size=B).astype(np.int64)  # config/datamodule/dataset/tacorl.yaml:11
    d[rs.uniform(size=B) < 0.1] = -1  # similar_robot_obs share (:12-14)
    return torch.from_numpy(d)


def make_play_batch(seed, B, T, cams_hw):
    """cams_hw: dict cam -> (H, W).  Returns the PlayDataset batch dict."""
    rs = _rs(seed, "play")
    states = {c: _img(rs, B, T, 3, h, w) for c, (h, w) in sorted(cams_hw.items())}
    goal = {c: _img(rs, B, 3, h, w) for c, (h, w) in sorted(cams_hw.items())}
    disp = _disp(rs, B)
    if B >= 2:
        disp[0], disp[1] = 1, 3  # make sure both reward branches appear
    return {
        "states": states,
        "actions": _actions(rs, B, T),
        "goal": goal,
        "disp": disp,
        "idx": torch.arange(B),
        "window_size": torch.full((B,), T),
    }


def make_transition_batch(seed, B, cams_hw):
    """GoalCondReplayBufferDataset batch (flat CQL baseline)."""
    rs = _rs(seed, "transition")
    cams = sorted(cams_hw.items())
    obs = {c: _img(rs, B, 3, h, w) for c, (h, w) in cams}
    nxt = {c: _img(rs, B, 3, h, w) for c, (h, w) in cams}
    goal = {c: _img(rs, B, 3, h, w) for c, (h, w) in cams}
    r = (rs.uniform(size=B) < 0.3).astype(np.int64)
    if B >= 2:
        r[0], r[1] = 1, 0
    return {
        "observations": {"observation": obs, "goal": goal},
        "actions": _actions(rs, B),
        "next_observations": {"observation": nxt, "goal": goal},
        "rewards": torch.from_numpy(r),
        "terminals": torch.from_numpy(r.copy()),
    }


def param_values(name, shape, seed):
    """Deterministic value for parameter ``name`` (logical reference shape)."""
    rs = _rs(seed, "param:" + name)
    shape = tuple(shape)
    if len(shape) >= 2:
        b = 1.0 / np.sqrt(float(np.prod(shape[1:])))
        v = rs.uniform(-b, b, size=shape)
    elif name.endswith("temperature"):
        v = rs.uniform(0.8, 1.2, size=shape)
    elif "norm" in name and name.endswith("weight"):
        v = rs.uniform(0.9, 1.1, size=shape)
    elif name.startswith("log_alpha"):
        v = rs.uniform(-0.3, 0.3, size=shape)
    else:
        v = rs.uniform(-0.05, 0.05, size=shape)
    return torch.from_numpy(v.astype(np.float32))


@torch.no_grad()
def fill_params_(module, seed):
    """Overwrite every parameter of an nn.Module, keyed by its state-dict name."""
    for name, p in module.named_parameters():
        p.copy_(param_values(name, p.shape, seed).to(p.device))
    return module


def tensor_stats(t, k=16):
    """(l2 norm, sum, k sampled elements) - the compact fingerprint goldens store."""
    f = t.detach().double().reshape(-1).cpu()
    n = f.numel()
    idx = (np.arange(k, dtype=np.int64) * 2654435761 + 12345) % max(n, 1)
    return np.concatenate([[f.norm().item(), f.sum().item()], f[torch.from_numpy(idx)].numpy()])
