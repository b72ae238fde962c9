"""tests/golden/augment.npz: outputs of the reference's RandomShiftsAug (utils/transforms.py:265-299) on seeded
uint8 frames, with the randint draws recorded.  Build container only (needs /root/reference)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as H  # noqa: E402

H.install_shims()
from tacorl.utils.transforms import RandomShiftsAug  # noqa: E402

out = {}
for tag, (n, hw, pad, seed) in {"a": (6, 20, 4, 1), "b": (5, 84, 4, 2), "c": (3, 128, 6, 3)}.items():
    rs = np.random.RandomState(seed)
    frames = rs.randint(0, 256, size=(n, hw, hw, 3)).astype(np.uint8)
    x = torch.from_numpy(frames).permute(0, 3, 1, 2).float()  # the dataset hands CHW float 0..255 tensors to the transform
    rec = {}
    o_randint = torch.randint

    def randint(*a, **kw):
        kw2 = dict(kw)
        r = o_randint(*a, **kw2)
        rec["shift"] = r.clone()
        return r

    torch.manual_seed(seed)
    torch.randint = randint
    try:
        y = RandomShiftsAug(pad)(x)
    finally:
        torch.randint = o_randint
    out[f"{tag}/cfg"] = np.array([n, hw, pad, seed])
    out[f"{tag}/shift"] = rec["shift"].reshape(n, 2).numpy().astype(np.int32)  # (..., 0) = x, (..., 1) = y
    out[f"{tag}/out"] = y.permute(0, 2, 3, 1).numpy().astype(np.float32)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "augment.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) / 1e3, "kB")
