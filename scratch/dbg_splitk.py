import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tacorl_amd import ops
from tacorl_amd._lib import call, ptr
import torch.nn.functional as F
dev = torch.device('cuda:0')
for rep in range(2):
  for compute in (0,):
    for (M, K, N, act) in [(45, 2048, 2048, 0), (48, 2048, 32, 0), (3, 4096, 32, 0), (45, 2048, 182, 0), (48, 32, 2048, 1), (3, 2048, 2048, 1)]:
        torch.manual_seed(M + K + N + rep)
        x = torch.randn(M, K); w = torch.randn(N, K) / math.sqrt(K); b = torch.randn(N) * 0.1
        add = torch.randn(M, N) if act else None
        ref = F.linear(x, w, b) + (add if add is not None else 0)
        if act: ref = F.relu(ref)
        xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
        ad = add.to(dev) if add is not None else None
        y = torch.full((M, N), float('nan'), device=dev)
        nb = ops.L.lib().tacorl_linear_add_fwd_ws_bytes(1, ops.int_array([M]), K, N)
        ws = ops.workspace(nb, dev, "t")
        call("tacorl_linear_add_fwd", 1, ops.ptr_array([xd]), K, ops.ptr_array([wd]), ops.ptr_array([bd]),
             ops.ptr_array([ad]) if ad is not None else None, N, ops.ptr_array([y]), N, ops.int_array([M]), K, N, act, compute,
             ptr(ws), ws.numel(), ops.stream())
        torch.cuda.synchronize()
        err = ((y.cpu() - ref).norm() / ref.norm()).item()
        print(rep, (M, K, N, act), "ws", nb, "relerr", f"{err:.3g}")
