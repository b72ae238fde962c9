"""How reproducible is a gradient under bf16 operand rounding?  Evaluate the rounded oracle twice - the second time with
every parameter and input perturbed by one fp32 ulp-sized relative noise (what a different summation order does) - and
report per-tensor relative differences.  This is the floor no bf16 kernel can be held under."""
import sys, torch
sys.path.insert(0, '.')
from oracle import tacorl_oracle as O
from tests.golden_util import Golden, spec_for
name = sys.argv[1]
eps = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
g = Golden(name); spec = None if g.cfg["kind"] == "playlmp" else spec_for(g)
def run(seed):
    P = g.params()
    if seed:
        gen = torch.Generator().manual_seed(seed)
        P = {k: v * (1 + eps * torch.randn(v.shape, generator=gen)) for k, v in P.items()}
    P = O.require_grad_(P, frozen_prefixes=("perceptual_encoder.", "plan_recognition.") if g.cfg["kind"] == "tacorl" else ())
    with O.operand_rounding(torch.bfloat16):
        if g.cfg["kind"] == "tacorl":
            return O.tacorl_step(P, O.make_opts(P, spec), spec, g.batch(0), g.noise(0), g.cfg["epoch"])[2]
        if g.cfg["kind"] == "cql":
            return O.cql_step(P, O.make_opts(P, spec), spec, g.batch(0), g.noise(0), g.cfg["epoch"])[1]
        return O.playlmp_step(P, O.Adam([n for n in P], 1e-4), g.batch(0), g.noise(0), sorted(g.cams))[1]
a = run(0)
worst = {}
for s in (1, 2, 3):
    b = run(s)
    for k in a:
        if a[k].norm() > 0:
            e = ((a[k] - b[k]).norm() / a[k].norm()).item()
            worst[k] = max(worst.get(k, 0), e)
for e, k in sorted(((e, k) for k, e in worst.items()), reverse=True)[:12]:
    print(f"{e:.2e}  {k}")
