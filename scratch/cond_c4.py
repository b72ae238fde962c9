"""How well-conditioned are the tacorl_c4 step-1 gradients?  fp32 oracle vs fp64 oracle (same algorithm)."""
import sys, torch
sys.path.insert(0, '.')
from oracle import tacorl_oracle as O
from tests.golden_util import Golden, spec_for
name = sys.argv[1] if len(sys.argv) > 1 else "tacorl_c4"
g = Golden(name); spec = spec_for(g)
def cast(x, dt):
    if isinstance(x, dict): return {k: cast(v, dt) for k, v in x.items()}
    if isinstance(x, list): return [cast(v, dt) for v in x]
    return x.to(dt) if torch.is_tensor(x) and x.is_floating_point() else x
res = {}
for dt in (torch.float32, torch.float64):
    torch.set_default_dtype(dt)
    P = O.require_grad_(cast(g.params(), dt), frozen_prefixes=("perceptual_encoder.", "plan_recognition."))
    opts = O.make_opts(P, spec)
    out = []
    for step in range(g.cfg["steps"]):
        logs, plan, grads = O.tacorl_step(P, opts, spec, cast(g.batch(step), dt), cast(g.noise(step), dt), g.cfg["epoch"])
        out.append({k: v.detach().clone() for k, v in grads.items()})
        if dt == torch.float64:  # keep both runs on the same parameter trajectory (the fp32 one)
            pass
    res[dt] = out
torch.set_default_dtype(torch.float32)
for step in range(g.cfg["steps"]):
    a, b = res[torch.float32][step], res[torch.float64][step]
    worst = sorted(((((a[k].double() - b[k]).norm() / b[k].norm().clamp_min(1e-300)).item(), k) for k in a if b[k].norm() > 0), reverse=True)[:6]
    print("step", step, [(f"{e:.2e}", k) for e, k in worst])
