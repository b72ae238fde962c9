"""Micro-benchmark of skinny linear_fwd shapes through a captured graph (GPU-side time per launch)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tacorl_amd import ops, _lib
dev = torch.device("cuda:0")
_lib.call("tacorl_hip_init", 0)
def run(M, K, N, nprob=1, act=2, reps=50):
    xs = [torch.randn(M, K, device=dev) for _ in range(nprob)]
    ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(nprob)]
    bs = [torch.zeros(N, device=dev) for _ in range(nprob)]
    ys = [torch.empty(M, N, device=dev) for _ in range(nprob)]
    def one():
        ops.call("tacorl_linear_fwd", nprob, ops.ptr_array(xs), K, ops.ptr_array(ws), ops.ptr_array(bs), ops.ptr_array(ys),
                 None, ops.int_array([M] * nprob), K, N, act, 1, ops.stream())
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        one(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): one()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"M={M:5d} K={K:5d} N={N:4d} nprob={nprob}: {e0.elapsed_time(e1) / reps * 1000:7.2f} us/launch", flush=True)
for shp in [(256, 32, 256), (256, 64, 256), (256, 256, 256), (256, 256, 32), (1024, 256, 256), (4096, 272, 256), (4096, 256, 256), (4096, 256, 1),
            (256, 2048, 2048), (4096, 128, 2048)]:
    run(*shp)
run(256, 256, 256, nprob=5)
