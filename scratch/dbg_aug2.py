import sys, torch
sys.path.insert(0, '.')
from oracle import augment_oracle as A
from tests.test_data_gpu import _pack
from tacorl_amd import _lib
_lib.call("tacorl_hip_init", 0)
g = torch.Generator(device="cuda:0").manual_seed(1)
n, hw, pad = 2, 20, 2
frames = torch.randint(0, 256, (n, hw, hw, 3), device="cuda:0", dtype=torch.uint8, generator=g)
x = (frames.cpu().float() / 255).permute(0, 3, 1, 2)
for order in ([2, 3, 1, 0], [1, 3, 2, 0], [0, 3, 1, 2], [2, 1, 3, 0], [3, 2, 1, 0]):
    jit = torch.tensor([[1, 1, 0.2] + order + [1.0]] * n, device="cuda:0", dtype=torch.float32)
    got = _pack(frames, None, jit, pad, torch.float32).cpu()
    res = {}
    for reps, hf in ((0, 0.0), (1, 0.2), (2, 0.4), (1, -0.2), (1, 0.1)):
        ref = ((A.adjust_hue(x, hf) if reps else x) - 0.5) / 0.5
        res[(reps, hf)] = round((got - ref.permute(0, 2, 3, 1)).abs().max().item(), 4)
    print(order, res)
