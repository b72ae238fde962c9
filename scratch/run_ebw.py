"""Full-size fused encoder backward (3 networks x 512 images of 84x84) for profiling."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tacorl_amd import ops, blocks, _lib
dev = torch.device("cuda:0")
_lib.call("tacorl_hip_init", 0)
H = W = 84
NI = int(os.environ.get("NIMG", "512"))
n = [NI, NI, NI]
torch.manual_seed(0)
img = (torch.rand(NI, H, W, 3, device=dev) * 2 - 1).to(torch.bfloat16)
flats = [torch.randn(blocks.encoder_size(), device=dev) * 0.05 for _ in n]
for f in flats:
    blocks.encoder_views(f)["model.6.temperature"].fill_(1.0)
outs = [torch.empty(k, 32, device=dev) for k in n]
acts = [torch.empty(ops.encoder_act_layout(k, H, W)[1], device=dev) for k in n]
douts = [torch.randn(k, 32, device=dev) for k in n]
grads = [torch.zeros_like(f) for f in flats]
ops.encoder_fwd([img] * 3, flats, outs, acts, H, W, 1)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    ops.encoder_bwd([img] * 3, flats, acts, douts, grads, H, W, 1, fused=True)
torch.cuda.synchronize()
print("ok", [float(g.abs().sum()) for g in grads])
