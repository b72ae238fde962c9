"""Repeated fused-vs-per-layer encoder forward comparisons on fresh random data (screens for sporadic errors that a
hand-padded MFMA hazard would cause: wrong values on some lanes of some launches)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tacorl_amd import _lib, blocks, ops
dev = torch.device("cuda:0"); _lib.call("tacorl_hip_init", 0)
H = W = 84
worst = 0.0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for it in range(iters):
    torch.manual_seed(1000 + it)
    n = [int(torch.randint(1, 700, (1,))) for _ in range(3)]
    flats = [torch.randn(blocks.encoder_size(), device=dev) * 0.05 for _ in n]
    for f in flats:
        blocks.encoder_views(f)["model.6.temperature"].fill_(1.0)
    imgs = [(torch.rand(k, H, W, 3, device=dev) * 2 - 1).to(torch.bfloat16) for k in n]
    o_f = [torch.empty(k, 32, device=dev) for k in n]
    o_r = [torch.empty(k, 32, device=dev) for k in n]
    a_f = [torch.empty(ops.encoder_act_layout(k, H, W)[1], device=dev) for k in n]
    a_r = [torch.empty(ops.encoder_act_layout(k, H, W)[1], device=dev) for k in n]
    packed = [torch.empty(_lib.lib().tacorl_encoder_fused_wpk_bytes(), dtype=torch.uint8, device=dev) for _ in n]
    ops.call("tacorl_encoder_pack_weights", len(n), ops.ptr_array(flats), ops.ptr_array(packed), ops.stream())
    ops.call("tacorl_encoder_fwd_fused", len(n), ops.ptr_array(imgs), ops.ptr_array(packed), ops.ptr_array(flats),
             ops.ptr_array(o_f), ops.ptr_array(a_f), ops.int_array(n), H, W, ops.stream())
    ops.call("tacorl_encoder_fwd", len(n), ops.ptr_array(imgs), ops.ptr_array(flats), ops.ptr_array(o_r), ops.ptr_array(a_r),
             ops.int_array(n), H, W, 1, 1, ops.stream())
    torch.cuda.synchronize()
    for x, y, ax, ay in zip(o_f, o_r, a_f, a_r):
        d = (x - y).abs().max().item() / max(y.abs().max().item(), 1e-6)
        da = (ax - ay).abs().max().item() / max(ay.abs().max().item(), 1e-6)
        worst = max(worst, d, da)
        assert torch.isfinite(x).all() and torch.isfinite(ax).all()
print(f"{iters} iterations, worst relative max-abs difference (outputs and saved activations): {worst:.3e}")
