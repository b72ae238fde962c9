import sys, torch
sys.path.insert(0, '.')
from tacorl_amd import synth
from tests import cfg_util as C
from tacorl_amd.modules.cql.cql_offline_lightning import CQL_Offline
torch.manual_seed(5)
strip = lambda c: {k: v for k, v in c.items() if k not in ("_target_", "_recursive_")}
m = CQL_Offline(**strip(C.cql_cfg(device="cuda:0")))
opts = m.configure_optimizers()
def mv(x):
    if isinstance(x, dict): return {k: mv(v) for k, v in x.items()}
    return x.cuda() if torch.is_tensor(x) else x
for i in range(3):
    torch.manual_seed(1000 + i); torch.cuda.manual_seed(1000 + i)
    m.training_step(mv(synth.make_transition_batch(50 + i, 3, {"rgb_static": (84, 84)})), i)
    torch.cuda.synchronize()
    bad_g = [k for k, v in m.named_gradients().items() if not torch.isfinite(v).all()]
    bad_p = [k for k, v in m.state_dict().items() if v.is_floating_point() and not torch.isfinite(v).all()]
    bad_m = [(o.name, n) for o in opts for (blk, ps, ms, vs) in o._entries for n in ms if not torch.isfinite(ms[n]).all() or not torch.isfinite(vs[n]).all()]
    print(i, "grads", bad_g[:5], "params", bad_p[:5], "moments", bad_m[:5], {k: round(v, 4) for k, v in list(m.logged.items())[:4]})
e = m.engine
print("actor temp grad", m.named_gradients()["actor.encoder.networks.rgb_static.model.6.temperature"], "m", e.actor.views_of(e.actor.m)["encoder.networks.rgb_static.model.6.temperature"])
from tacorl_amd import ops
B = 3
for (k, c), a in e.enc_act.items():
    nimg = {"a_og": 2*B, "a_nx": B, "q1": 2*B, "q2": 2*B, "tq1": 2*B, "tq2": 2*B}[k]
    offs, tot = ops.encoder_act_layout(nimg, 84, 84)
    segs = {n: a[offs[j]:(offs[j+1] if j+1 < 5 else tot)] for j, n in enumerate(["y1","y2","y3","feat","fc1"])}
    print(k, {n: (bool(torch.isfinite(s).all()), float(s.abs().max())) for n, s in segs.items()})
print({k: bool(torch.isfinite(v).all()) for k, v in e.enc_dout.items()})
print("nonfinite grads:", [k for k, v in m.named_gradients().items() if not torch.isfinite(v).all()])
print("temps", {k: v.item() for k, v in m.state_dict().items() if k.endswith("temperature")})
