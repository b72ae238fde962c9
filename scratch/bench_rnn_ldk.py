"""Probe: batched ring GEMM (3 problems, M=256, 2048x2048) with contiguous operand rows (4096 B apart) vs padded rows"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tacorl_amd import ops, _lib
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
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
K = N = 2048
for M in (64, 256, 512):
    for ld in (2048, 2048 + 64, 2048 + 128, 2048 + 192):
        if ld == 2048: os.environ.pop("TACORL_RNN_LDK", None)
        else: os.environ["TACORL_RNN_LDK"] = str(ld)
        xs = [torch.randn(M, ld, device=dev).to(torch.bfloat16) for _ in range(3)]
        wsb = [(torch.randn(N, ld, device=dev) * 0.02).to(torch.bfloat16) for _ in range(3)]
        bb = [torch.zeros(N, device=dev) for _ in range(3)]; adds = [torch.randn(M, N, device=dev) for _ in range(3)]
        ys = [torch.empty(M, N, device=dev) for _ in range(3)]; ybs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(3)]
        f = lambda: ops.call("tacorl_rnn_linear_fwd_batch", 3, ops.ptr_array(xs), ops.ptr_array(wsb), ops.ptr_array(bb), ops.ptr_array(adds), N,
                             ops.ptr_array(ys), ops.ptr_array(ybs), M, K, N, ops.int_array([1] * 3), ops.stream())
        t = timeit(f)
        f(); torch.cuda.synchronize()
        ref = torch.relu(xs[0][:, :K].float() @ wsb[0][:, :K].float().T + adds[0])
        print(f"M={M} row stride {ld * 2} B: {t:.2f} us per 3-problem launch   max err {(ys[0] - ref).abs().max().item():.2e}", flush=True)
