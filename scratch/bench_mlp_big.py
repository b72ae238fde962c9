"""Many-row Q-MLP kernels alone (C5: 2 networks x 99 328 rows, dims [71,256,256,256,1]): time forward / input-gradient chain /
weight gradients back to back; with a -DMLP_STAMPS scratch library (TACORL_LIB=scratch/libs/stamps.so) print the phase clocks of
one workgroup's wave 0."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from tacorl_amd import _lib, blocks, ops

if os.environ.get("TACORL_LIB"):  # a scratch build of the library (scratch/mklib2.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["TACORL_LIB"])
dev = torch.device("cuda:0")
_lib.call("tacorl_hip_init", 0)
dims, acts, ld = [71, 256, 256, 256, 1], [2, 2, 2, 0], 72
M = int(os.environ.get("ROWS", 99328))
L = len(dims) - 1
xs, flats, fb, act, douts, dxs, grads = [], [], [], [], [], [], []
for i in range(2):
    flat = torch.zeros(blocks.mlp_size(dims), device=dev)
    v = blocks.mlp_views(flat, 0, dims, [(f"l{l}.w", f"l{l}.b") for l in range(L)])
    g = torch.Generator().manual_seed(i)
    for l in range(L):
        v[f"l{l}.w"].copy_((torch.rand(dims[l + 1], dims[l], generator=g) * 2 - 1) / math.sqrt(dims[l]))
    x = torch.zeros(M, ld, device=dev)
    x[:, :71] = torch.rand(M, 71, device=dev) * 2 - 1
    xs.append(x); flats.append(flat); fb.append(flat.to(torch.bfloat16))
    act.append(torch.zeros(ops.mlp_act_layout(M, dims, acts)[2], device=dev))
    douts.append(torch.randn(M, 1, device=dev)); dxs.append(torch.zeros(M, ld, device=dev)); grads.append(torch.zeros_like(flat))
Ms = [M, M]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


fwd = lambda: ops.mlp_fwd(xs, ld, flats, act, Ms, dims, acts, 1, params_bf16=fb, lean=True)
bwd = lambda: ops.mlp_bwd_fused_dgrad(flats, act, douts, 1, dxs, ld, Ms, dims, acts, "bench_big", lean=True)
wg = lambda: ops.mlp_bwd_fused_wgrad(xs, ld, act, douts, 1, grads, Ms, dims, acts, "bench_big", lean=True)
print(f"rows {M} x 2 networks: forward {timeit(fwd):.1f} us, input-gradient chain {timeit(bwd):.1f} us, weight gradients {timeit(wg):.1f} us")
L_ = _lib.lib()
if hasattr(L_, "tacorl_dbg_mlp_stamps"):
    buf = (C.c_ulonglong * 128)()
    fwd(); torch.cuda.synchronize()
    L_.tacorl_dbg_mlp_stamps(buf)
    t = list(buf)[:64]
    names = {1: "layer start", 2: "mfma loop done", 3: "barrier A", 4: "epilogue", 5: "barrier B", 6: "s -> lds + barrier C", 7: "copy-out issued"}
    print("forward, workgroup 300, wave 0 (clocks since kernel entry of that wave):")
    prev = t[0]
    for l in range(L):
        for j in range(1, 8):
            k = 8 * l + j
            if t[k]:
                print(f"  layer {l} {names[j]:<22} +{t[k] - prev:7d}  (at {t[k] - t[0]})")
                prev = t[k]
