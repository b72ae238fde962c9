import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from tacorl_amd import _lib, blocks, ops
from tests.test_kernels_gpu import _enc_params, rnd
dev = torch.device('cuda:0'); H = W = 84; k = 19
P = _enc_params(70); img = rnd(k, 3, H, W, seed=80)
xb = img.to(torch.bfloat16).float()
y1_ref = F.relu(F.conv2d(xb, P["model.0.weight"].to(torch.bfloat16).float(), P["model.0.bias"], stride=4)).permute(0, 2, 3, 1).reshape(k, -1, 32)
flat = torch.zeros(blocks.encoder_size(), device=dev); blocks.load_named(blocks.encoder_views(flat), P)
im = img.permute(0, 2, 3, 1).contiguous().to(dev).to(torch.bfloat16)
out_f, out_g = torch.empty(k, 32, device=dev), torch.empty(k, 32, device=dev)
offs, tot = ops.encoder_act_layout(k, H, W)
act_f, act_g = torch.full((tot,), float('nan'), device=dev), torch.empty(tot, device=dev)
pk = torch.empty(_lib.lib().tacorl_encoder_fused_wpk_bytes(), dtype=torch.uint8, device=dev)
ops.call("tacorl_encoder_pack_weights", 1, ops.ptr_array([flat]), ops.ptr_array([pk]), ops.stream())
ops.call("tacorl_encoder_fwd_fused", 1, ops.ptr_array([im]), ops.ptr_array([pk]), ops.ptr_array([flat]), ops.ptr_array([out_f]), ops.ptr_array([act_f]), ops.int_array([k]), H, W, ops.stream())
ops.encoder_fwd([im], [flat], [out_g], [act_g], H, W, 1)
torch.cuda.synchronize()
yf = act_f[:offs[1]].view(k, -1, 32).cpu(); yg = act_g[:offs[1]].view(k, -1, 32).cpu()
print("fused vs ref", ((yf - y1_ref).norm() / y1_ref.norm()).item(), "generic vs ref", ((yg - y1_ref).norm() / y1_ref.norm()).item())
d = (yf - y1_ref).abs()
bad = (d > 0.02)
print("bad frac", bad.float().mean().item(), "by channel", bad.float().mean((0, 1))[:32].tolist())
print("by pixel (first 40)", bad.float().mean((0, 2))[:40].tolist())
