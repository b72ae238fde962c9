import os, sys, subprocess, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# child mode: run the fused forward with the library in TACORL_SCRATCH_LIB (or the product one) and save outputs
from tacorl_amd import _lib, blocks, ops
if os.environ.get("TACORL_SCRATCH_LIB"): _lib.LIB_PATH = os.environ["TACORL_SCRATCH_LIB"]
dev = torch.device("cuda:0")
torch.manual_seed(3)
n = [int(x) for x in sys.argv[2].split(",")]
wg = int(sys.argv[3])
H = W = 84
flats, imgs, outs, packed, acts = [], [], [], [], []
use_act = os.environ.get("DBG_ACT") == "1"
for k in n:
    f = torch.randn(blocks.encoder_size(), device=dev) * 0.05
    for kk, vv in blocks.encoder_views(f).items():
        if kk.endswith('temperature'): vv.fill_(1.0)
    flats.append(f)
    imgs.append((torch.rand(k, H, W, 3, device=dev) * 2 - 1).to(torch.bfloat16))
    outs.append(torch.full((k, 32), float("nan"), device=dev))
    packed.append(torch.empty(_lib.lib().tacorl_encoder_fused_wpk_bytes(), dtype=torch.uint8, device=dev))
    acts.append(torch.zeros(ops.encoder_act_layout(k, H, W)[1], device=dev) if use_act else None)
ops.call("tacorl_encoder_pack_weights", len(n), ops.ptr_array(flats), ops.ptr_array(packed), ops.stream())
for rep in range(3):
    ops.call("tacorl_encoder_fwd_fused_wg", len(n), ops.ptr_array(imgs), ops.ptr_array(packed), ops.ptr_array(flats),
             ops.ptr_array(outs), ops.ptr_array(acts) if use_act else None, ops.int_array(n), H, W, wg, ops.stream())
    torch.cuda.synchronize()
torch.save(([o.cpu() for o in outs], [a.cpu() if a is not None else None for a in acts], [ops.encoder_act_layout(k, H, W) for k in n]), sys.argv[1])
