"""tests/golden/resize.npz: torch.nn.functional.interpolate(mode="bilinear", align_corners=False) - what
torchvision.transforms.Resize computes for float tensors in the reference's torchvision generation (no antialias;
config/datamodule/transform_manager/transforms/rl_train.yaml:3-4,16-17) - on seeded uint8 frames given as 0..255 floats,
CHW, as the reference's TransformManager hands them over (utils/transforms.py:29-44).  Needs torch only."""
import os

import numpy as np
import torch
import torch.nn.functional as F

out = {}
# (frames, source H, W, target H, W, seed): the real-world cameras are 200x200 -> 128x128 (static) / 84x84 (gripper)
for tag, (n, hs, ws, ht, wt, seed) in {"static": (1, 200, 200, 128, 128, 1), "gripper": (1, 200, 200, 84, 84, 2),
                                        "rect": (1, 150, 200, 84, 84, 3), "up": (1, 40, 60, 44, 60, 4)}.items():
    frames = np.random.RandomState(seed).randint(0, 256, size=(n, hs, ws, 3)).astype(np.uint8)
    x = torch.from_numpy(frames).permute(0, 3, 1, 2).float()
    y = F.interpolate(x, size=(ht, wt), mode="bilinear", align_corners=False)
    out[f"{tag}/cfg"] = np.array([n, hs, ws, ht, wt, seed])
    out[f"{tag}/out"] = y.permute(0, 2, 3, 1).numpy().astype(np.float32)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "resize.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path) / 1e3, "kB")
