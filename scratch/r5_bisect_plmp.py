"""which operand rounding moves PlayLMP's encoder gradients (C1 size: B = 32, window 16)?
CPU only: the oracle's playlmp_step gradient evaluated in fp32, with bf16 operand rounding in every contraction, and with
the rounding restricted to one contraction class (operand_rounding(only=...)); per variant the relative distance of the
soft-argmax temperature gradient, the three conv weight gradients and d_emb (the gradient entering the encoder's backward)
from the fp32 evaluation and from the all-rounded one.  Also the reproducibility of the all-rounded evaluation under a
1-ulp relative perturbation of the parameters (the 'floor' of golden_util.gradient_floor)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from oracle import tacorl_oracle as O
from tacorl_amd import synth
from tests.golden_util import Golden
torch.set_num_threads(8)
B = int(os.environ.get("B", 32))
g = Golden("playlmp")
P0 = g.params()
batch = synth.make_play_batch(4400, B, 16, {"rgb_static": (84, 84)})
gen = torch.Generator().manual_seed(9)
nz = dict(eps_plan=torch.randn(B, 16, generator=gen), u_plan=torch.rand(B, 16, generator=gen), u_goal=torch.rand(B, 32, generator=gen),
          rand=[torch.rand(B, 15, 6, 10, generator=gen), torch.rand(B, 15, 6, generator=gen), torch.rand(B, 15, 6, 10, generator=gen), torch.rand(B, 15, 6, generator=gen)])
E = "perceptual_encoder.networks.rgb_static."
KEYS = [E + "model.6.temperature", E + "model.0.weight", E + "model.2.weight", E + "model.4.weight", E + "fc_layers.0.weight", "d_emb",
        "action_decoder.rnn.weight_hh_l0", "plan_recognition.transformer_encoder.layers.0.linear1.weight", "plan_proposal.policy.fc_layers.0.weight"]
def ev(only, perturb=None):
    P = {k: v.clone() for k, v in P0.items()}
    if perturb is not None:
        gp = torch.Generator().manual_seed(perturb)
        P = {k: v * (1 + 1e-7 * torch.randn(v.shape, generator=gp)) for k, v in P.items()}
    O.require_grad_(P)
    ex = {}
    t0 = time.time()
    if only == "fp32":
        _, gr = O.playlmp_step(P, None, batch, nz, ["rgb_static"], step=False, extra=ex)
    else:
        with O.operand_rounding(torch.bfloat16, only=None if only == "all" else only):
            _, gr = O.playlmp_step(P, None, batch, nz, ["rgb_static"], step=False, extra=ex)
    gr["d_emb"] = ex["d_emb"]
    return {k: gr[k].detach().double() for k in KEYS}, time.time() - t0
rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-300)).item()
ref, dt = ev("fp32"); print(f"(one evaluation: {dt:.1f} s)", flush=True)
allr, _ = ev("all")
rows = []
for name, only in [("all contractions", "all"), ("convolutions only", {"conv"}), ("RNN + heads only", {"rnn"}), ("transformer only", {"attn"}),
                   ("other Linear only (goal encoder, plan proposal, encoder FC)", {"linear"}), ("all but RNN", {"conv", "attn", "linear"}), ("all but conv", {"rnn", "attn", "linear"})]:
    got = allr if only == "all" else ev(only)[0]
    rows.append((name, [rel(got[k], ref[k]) for k in KEYS], [rel(got[k], allr[k]) for k in KEYS]))
    print(name, " vs fp32:", " ".join(f"{x:.3g}" for x in rows[-1][1]), flush=True)
fl = [0.0] * len(KEYS)
for seed in (1000, 1001):
    got = ev("all", perturb=seed)[0]
    fl = [max(a, rel(got[k], allr[k])) for a, k in zip(fl, KEYS)]
print("all contractions, parameters perturbed by 1e-7 relative, vs unperturbed:", " ".join(f"{x:.3g}" for x in fl))
short = ["temperature", "conv1.w", "conv2.w", "conv3.w", "enc fc1.w", "d_emb", "rnn W_hh0", "PR linear1.w", "plan-proposal fc0.w"]
with open(os.environ.get("OUT", "/tmp/bisect.md"), "w") as f:
    f.write(f"| bf16 operand rounding in | " + " | ".join(short) + " |\n|---|" + "---|" * len(short) + "\n")
    for name, a, _ in rows:
        f.write(f"| {name} (vs fp32) | " + " | ".join(f"{x:.2e}" for x in a) + " |\n")
    for name, _, b in rows[1:]:
        f.write(f"| {name} (vs all-rounded) | " + " | ".join(f"{x:.2e}" for x in b) + " |\n")
    f.write(f"| all, parameters x (1 + 1e-7 N(0,1)) (vs all-rounded) | " + " | ".join(f"{x:.2e}" for x in fl) + " |\n")
print(open(os.environ.get("OUT", "/tmp/bisect.md")).read())
