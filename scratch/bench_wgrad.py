import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tacorl_amd import ops, _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
def timeit(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000
R, H = 3840, 2048
dz = torch.randn(R, H, device=dev).to(torch.bfloat16); x = torch.randn(R, H, device=dev).to(torch.bfloat16)
dw = torch.empty(H, H, device=dev); db = torch.empty(H, device=dev)
f = lambda: ops.call("tacorl_rnn_wgrad", ops.ptr(dz), H, ops.ptr(x), H, R, H, H, ops.ptr(dw), ops.ptr(db), 0, ops.stream())
t = timeit(f)
ref = dz.float().t() @ x.float()
print(f"rnn_wgrad R={R} {H}x{H}: {t:.1f} us = {2 * R * H * H / t / 1e6:.0f} TFLOP/s; rel err {((dw - ref).norm() / ref.norm()).item():.2e}")
