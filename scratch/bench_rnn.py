import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tacorl_amd import ops, _lib
dev = torch.device("cuda:0")
_lib.call("tacorl_hip_init", 0)
def timeit(fn, reps=40):
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
for M, K, N in [(256, 2048, 2048), (3840, 2048, 2048)]:
    x = torch.randn(M, K, device=dev); xb = x.to(torch.bfloat16)
    w = torch.randn(N, K, device=dev) * 0.02; wb = w.to(torch.bfloat16)
    b = torch.zeros(N, device=dev); add = torch.randn(M, N, device=dev)
    y = torch.empty(M, N, device=dev); yb = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(max(256, _lib.lib().tacorl_linear_add_fwd_ws_bytes(1, ops.int_array([M]), K, N)), dtype=torch.uint8, device=dev)
    base = lambda: ops.call("tacorl_linear_add_fwd", 1, ops.ptr_array([x]), K, ops.ptr_array([w]), ops.ptr_array([b]), ops.ptr_array([add]), N,
                            ops.ptr_array([y]), N, ops.int_array([M]), K, N, 1, 1, ops.ptr(ws), ws.numel(), ops.stream())
    new = lambda: ops.call("tacorl_rnn_linear_fwd", ops.ptr(xb), ops.ptr(wb), ops.ptr(b), ops.ptr(add), N, ops.ptr(y), ops.ptr(yb), M, K, N, 1, ops.stream())
    print(f"M={M}: generic (split-K + reduce) {timeit(base):.2f} us   LDS-DMA ring {timeit(new):.2f} us", flush=True)
import ctypes as C
M, K, N = 256, 2048, 2048
xs = [torch.randn(M, K, device=dev).to(torch.bfloat16) for _ in range(3)]
wsb = [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(3)]
bb = [torch.zeros(N, device=dev) for _ in range(3)]; adds = [torch.randn(M, N, device=dev) for _ in range(3)]
ys = [torch.empty(M, N, device=dev) for _ in range(3)]; ybs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(3)]
for n in (1, 2, 3):
    f = lambda: ops.call("tacorl_rnn_linear_fwd_batch", n, ops.ptr_array(xs[:n]), ops.ptr_array(wsb[:n]), ops.ptr_array(bb[:n]),
                         ops.ptr_array(adds[:n]), N, ops.ptr_array(ys[:n]), ops.ptr_array(ybs[:n]), M, K, N, ops.int_array([1] * n), ops.stream())
    print(f"batch ring (2-stage) nprob={n}: {timeit(f):.2f} us", flush=True)
f(); torch.cuda.synchronize()
for p in range(3):
    ref = torch.relu(xs[p].float() @ wsb[p].float().T + bb[p] + adds[p])
    err = (ys[p] - ref).abs().max().item(); errb = (ybs[p].float() - ref).abs().max().item()
    print(f"problem {p}: max |y - ref| = {err:.3e} (bf16 copy {errb:.3e}), ref max {ref.abs().max().item():.2f}", flush=True)
