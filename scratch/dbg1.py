import torch, math, sys
sys.path.insert(0, '.')
from tacorl_amd import ops
import torch.nn.functional as F
dev = torch.device('cuda:0')
for compute in (0, 1):
    for (M, K, N) in [(37, 64, 256), (128, 32, 64), (16,32,16)]:
        torch.manual_seed(0)
        x = torch.randn(M, K); w = torch.randn(N, K) / math.sqrt(K); b = torch.zeros(N)
        ref = F.linear(x, w, b)
        y = ops.linear_fwd([x.to(dev)], [w.to(dev)], [b.to(dev)], 0, compute)[0].cpu()
        err = (y - ref).abs()
        print("compute", compute, (M, K, N), "max err", err.max().item())
        bad = err > 1e-2
        print(" bad rows:", bad.any(1).nonzero().flatten().tolist()[:40])
        print(" bad cols:", bad.any(0).nonzero().flatten().tolist()[:80])
