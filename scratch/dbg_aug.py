import sys, torch
sys.path.insert(0, '.')
from oracle import augment_oracle as A
from tests.test_data_gpu import _pack
from tacorl_amd import _lib
_lib.call("tacorl_hip_init", 0)
g = torch.Generator(device="cuda:0").manual_seed(1)
n, hw, pad = 3, 20, 2
frames = torch.randint(0, 256, (n, hw, hw, 3), device="cuda:0", dtype=torch.uint8, generator=g)
for name, (b, c, h), order in [("neutral", (1, 1, 0), [0, 1, 2, 3]), ("bright", (1.2, 1, 0), [0, 1, 2, 3]), ("contrast", (1, 0.8, 0), [0, 1, 2, 3]),
                                ("hue", (1, 1, 0.1), [0, 1, 2, 3]), ("hue-first", (1.1, 0.9, 0.1), [3, 1, 2, 0]), ("mix", (1.2, 0.8, -0.2), [2, 3, 1, 0]), ("neg-hue-last", (1, 1, -0.2), [0, 1, 2, 3]), ("neg-hue-first", (1, 1, -0.2), [3, 1, 2, 0]),
                                ("pos-mix", (1.2, 0.8, 0.2), [2, 3, 1, 0]), ("mix-nobc", (1, 1, 0.2), [2, 3, 1, 0]), ("mix-b", (1.2, 1, 0.2), [2, 3, 1, 0]), ("mix-c", (1, 0.8, 0.2), [2, 3, 1, 0])]:
    jit = torch.tensor([[b, c, h] + order + [1.0]] * n, device="cuda:0", dtype=torch.float32)
    got = _pack(frames, None, jit, pad, torch.float32).cpu()
    ref = A.train_pipeline(frames.cpu(), None, jit.cpu(), pad)
    d = (got - ref).abs()
    print(name, d.max().item(), d.mean().item(), got[0, 0, 0].tolist(), ref[0, 0, 0].tolist())
