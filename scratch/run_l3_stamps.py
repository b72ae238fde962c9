"""Phase clocks of the fused conv3 backward (a -DL3_STAMPS build): back-to-back launches, then the stamps of the last one."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from tacorl_amd import _lib
if os.environ.get("TACORL_SCRATCH_LIB"): _lib.LIB_PATH = os.environ["TACORL_SCRATCH_LIB"]
from tacorl_amd import ops, blocks
dev = torch.device("cuda:0")
_lib.call("tacorl_hip_init", 0)
H = W = 84
NI = int(os.environ.get("NIMG", "512"))
n = [NI] * 3
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
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for _ in range(reps):
    ops.encoder_bwd([img] * 3, flats, acts, douts, grads, H, W, 1, fused=True)
torch.cuda.synchronize()
L = _lib.lib()
if hasattr(L, "tacorl_l3_stamps_read"):
    buf = (ctypes.c_ulonglong * 16)()
    L.tacorl_l3_stamps_read(buf)
    names = ["prologue", "softargmax", "flush+bar", "role phase", "put+bar", "epilogue", "total clk", "wall(100MHz)"]
    for o, role in ((0, "dgrad wave"), (8, "wgrad wave")):
        tot, wall = buf[o + 6], buf[o + 7]
        print(role, " ".join(f"{nm}={buf[o + i]}" for i, nm in enumerate(names)), f"-> {tot / max(wall, 1) * 100:.0f} MHz, {wall / 100:.1f} us")
