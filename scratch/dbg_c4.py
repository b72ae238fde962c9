"""GPU: tacorl_c4 f32, step-1 gradients of HIP vs fp32 oracle vs fp64 oracle, all from the module's own step-1 parameters."""
import sys, torch
sys.path.insert(0, '.')
from oracle import tacorl_oracle as O
from tests.golden_util import Golden, spec_for
from tests.test_step_gpu import build_tacorl, to_dev
name = sys.argv[1] if len(sys.argv) > 1 else "tacorl_c4"
g = Golden(name); spec = spec_for(g)
mod = build_tacorl(g); mod.load_state_dict(g.params(), strict=False); mod.current_epoch = g.cfg["epoch"]
def cast(x, dt):
    if isinstance(x, dict): return {k: cast(v, dt) for k, v in x.items()}
    if isinstance(x, list): return [cast(v, dt) for v in x]
    return x.to(dt).clone() if torch.is_tensor(x) and x.is_floating_point() else x
def rel(a, b): return ((a.double().cpu().reshape(b.shape) - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()
for step in range(g.cfg["steps"]):
    sd = {k: v.detach().cpu().clone() for k, v in mod.state_dict().items() if v.dtype == torch.float32 and k in g.names}
    batch, noise = g.batch(step), g.noise(step)
    mod.training_step(to_dev(batch, mod.device), noise=to_dev(noise, mod.device)); torch.cuda.synchronize()
    hip = {k: v.detach().cpu().clone() for k, v in mod.named_gradients().items()}
    og = {}
    for dt in (torch.float32, torch.float64):
        torch.set_default_dtype(dt)
        P = O.require_grad_(cast(sd, dt), frozen_prefixes=("perceptual_encoder.", "plan_recognition."))
        _, _, og[dt] = O.tacorl_step(P, O.make_opts(P, spec), spec, cast(batch, dt), cast(noise, dt), g.cfg["epoch"])
    torch.set_default_dtype(torch.float32)
    rows = sorted(((rel(hip[k], og[torch.float64][k]), rel(og[torch.float32][k], og[torch.float64][k]), rel(hip[k], og[torch.float32][k]), k) for k in og[torch.float32] if k in hip and og[torch.float64][k].norm() > 0), reverse=True)[:8]
    print("step", step)
    for a, b, c, k in rows: print(f"   hip-vs-f64 {a:.2e}   torch32-vs-f64 {b:.2e}   hip-vs-torch32 {c:.2e}  {k}")
