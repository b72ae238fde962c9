import sys, copy, torch
sys.path.insert(0, '.')
from oracle import tacorl_oracle as O
from tests.golden_util import Golden, spec_for
from tests.test_step_gpu import build_tacorl, to_dev
g = Golden("tacorl_bc_ad"); spec = spec_for(g)
k = "actor.encoder.networks.rgb_static.model.6.temperature"
for mode in ("bf16", "f32"):
    mod = build_tacorl(g, compute=mode); mod.load_state_dict(g.params(), strict=False); mod.current_epoch = g.cfg["epoch"]
    mod.training_step(to_dev(g.batch(0), mod.device), noise=to_dev(g.noise(0), mod.device)); torch.cuda.synchronize()
    print(mode, "hip", mod.named_gradients()[k].item(), {n: v.item() for n, v in mod.named_gradients().items() if n.endswith("temperature")})
for dt in (None, torch.bfloat16):
    P = O.require_grad_(g.params(), frozen_prefixes=("perceptual_encoder.", "plan_recognition."))
    with O.operand_rounding(dt):
        gr = O.tacorl_step(P, O.make_opts(P, spec), spec, g.batch(0), g.noise(0), g.cfg["epoch"])[2]
    print("oracle", dt, {n: v.item() for n, v in gr.items() if n.endswith("temperature")})
