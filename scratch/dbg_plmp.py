"""GPU: playlmp f32 step-1 gradients: HIP vs fp32 oracle vs fp64 oracle, all from the module's own step-1 parameters."""
import sys, torch
sys.path.insert(0, '.')
from oracle import tacorl_oracle as O
from tests.golden_util import Golden
from tests.test_step_gpu import ACTOR, to_dev
from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
g = Golden("playlmp"); cams, c = sorted(g.cams), g.cfg
pr = dict(num_heads=8, num_layers=2, encoder_hidden_size=2048, fc_hidden_size=4096, latent_plan_dim=c["latent"], min_std=1e-4, dropout_p=0.0, max_position_embeddings=c["T"])
ad = dict(n_mixtures=10, num_layers=2, hidden_size=2048, out_features=7, num_classes=10, latent_plan_dim=c["latent"], rnn_model="rnn_decoder", include_goal=False)
mod = PlayLMP(plan_proposal=ACTOR, plan_recognition=pr, action_decoder=ad, plan_proposal_obs_modalities=cams, plan_proposal_goal_modalities=cams, plan_recognition_modalities=cams, action_decoder_modalities=cams, real_world=True, lr=1e-4, kl_beta=1e-3, device="cuda:0", compute_dtype="f32")
mod.load_state_dict(g.params(), strict=False)
def cast(x, dt):
    if isinstance(x, dict): return {k: cast(v, dt) for k, v in x.items()}
    if isinstance(x, list): return [cast(v, dt) for v in x]
    return x.to(dt).clone() if torch.is_tensor(x) and x.is_floating_point() else x
def rel(a, b): return ((a.double().cpu().reshape(b.shape) - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()
for step in range(2):
    sd = {k: v.detach().cpu().clone() for k, v in mod.state_dict().items() if v.dtype == torch.float32 and k in g.names}
    batch, nz = g.batch(step), g.noise(step)
    mod.training_step(to_dev(batch, mod.device), 0, noise={k: nz[k] for k in ("eps_plan", "u_plan")}); torch.cuda.synchronize()
    hip = {k: v.detach().cpu().clone() for k, v in mod.named_gradients().items()}
    og, ol = {}, {}
    for dt in (torch.float32, torch.float64):
        torch.set_default_dtype(dt)
        P = O.require_grad_(cast(sd, dt))
        ol[dt], og[dt] = O.playlmp_step(P, O.Adam([n for n in P], 1e-4), cast(batch, dt), cast(nz, dt), cams)
    torch.set_default_dtype(torch.float32)
    rows = sorted(((rel(hip[k], og[torch.float64][k]), rel(og[torch.float32][k], og[torch.float64][k]), rel(hip[k], og[torch.float32][k]), k) for k in og[torch.float32] if k in hip and og[torch.float64][k].norm() > 0), reverse=True)[:6]
    print("step", step, {k: (round(ol[torch.float32][k], 6), round(ol[torch.float64][k], 6), round(mod.logged.get("train/" + k, float("nan")), 6)) for k in ("action_loss", "kl_loss")})
    for a, b, c_, k in rows: print(f"   hip-vs-f64 {a:.2e}   torch32-vs-f64 {b:.2e}   hip-vs-torch32 {c_:.2e}  {k}")
