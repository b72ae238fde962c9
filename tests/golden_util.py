"""Load tests/golden/*.npz (made by oracle/gen_golden.py from the unmodified reference)
and re-derive the inputs they were computed on."""
import json
import os

import numpy as np
import torch

from tacorl_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.cfg = json.loads(str(self.z["config"]))
        self.names = [str(n) for n in self.z["param_names"]]
        self.shapes = json.loads(str(self.z["param_shapes"]))
        self.requires_grad = [bool(b) for b in self.z["param_requires_grad"]]
        self.cams = {k: tuple(v) for k, v in self.cfg["cams"].items()}

    # ---- inputs
    def params(self):
        """Initial parameters {reference state-dict name: tensor} (logical shapes)."""
        return {n: synth.param_values(n, s, self.cfg["seed"]) for n, s in zip(self.names, self.shapes)}

    def batch(self, step):
        bseed = self.cfg["seed"] * 100 + step
        if self.cfg["kind"] in ("cql", "rollout_cql"):
            return synth.make_transition_batch(bseed, self.cfg["B"], self.cams)
        return synth.make_play_batch(bseed, self.cfg["B"], self.cfg["T"], self.cams)

    def tape(self, step):
        pre = f"s{step}/noise/"
        keys = sorted(k for k in self.z.files if k.startswith(pre))
        return [(k[len(pre) + 3:], torch.from_numpy(self.z[k])) for k in keys]

    def noise(self, step):
        """Name the recorded draws (order: SURVEY 8a note 1)."""
        t = self.tape(step)
        kind = self.cfg["kind"]
        nz = {}
        if kind == "playlmp":
            # train-mode dropout masks of the plan recognition come first (1 embedding mask + 4 per encoder layer:
            # attention probabilities, dropout1, FFN dropout, dropout2 - reference plan_recognition_transformer.py:87-88)
            nd = sum(1 for k, _ in t if k == "dropout")
            assert all(k == "dropout" for k, _ in t[:nd])
            drop, t = [m for _, m in t[:nd]], t[nd:]
            if self.cfg.get("validate"):  # validation_step ends with pp_dist.sample() for its output dict (:341-344)
                assert t[-1][0] == "normal"
                t = t[:-1]
            assert [k for k, _ in t] == ["normal", "rand", "rand", "uniform01", "uniform01", "rand", "rand"]
            nz = dict(eps_plan=t[0][1], rand=[t[1][1], t[2][1], t[5][1], t[6][1]], u_plan=t[3][1], u_goal=t[4][1])
            if drop:
                nz["dropout"] = drop
            return nz
        i = 0
        if kind == "tacorl":
            nz["eps_pr"] = t[0][1]
            i = 1
        if kind == "cql":  # discrete gripper draws interleave (actor.py:83-97,118-132)
            order = ["eps_pi", "g_pi", "eps_next", "g_next", "u_rand", "eps_cur", "g_cur", "eps_nxt", "g_nxt"]
        else:
            order = ["eps_pi", "eps_next", "u_rand", "eps_cur", "eps_nxt"]
        assert len(t) - i == len(order), (len(t), order)
        for k, (_, v) in zip(order, t[i:]):
            nz[k] = v
        return nz

    # ---- expected outputs
    def logged(self, step):
        d = json.loads(str(self.z[f"s{step}/logged"]))
        return {k.split("/", 1)[1]: v for k, v in d.items()}

    def latent_plan(self, step):
        return torch.from_numpy(self.z[f"s{step}/latent_plan"])

    def stats(self, step, what):
        pre = f"s{step}/{what}/"
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre)}


def spec_for(g):
    """ACSpec equal to what the reference module was built with (gen_golden.CASES +
    config/module/{tacorl,cql_offline_goal_cond}.yaml)."""
    from oracle import tacorl_oracle as O

    c = g.cfg
    cams = sorted(g.cams)
    ov = c.get("overrides", {})
    if c["kind"] == "tacorl":
        base = dict(n=4, discount=0.95, actor_lr=1e-4, critic_lr=3e-4, deterministic_backup=True,
                    reward_scale=10.0, bc_epochs=5, with_lagrange=True, action_decoder_lr=3e-4)
        base.update({("n" if k == "n_action_samples" else k): v for k, v in ov.items()})
        return O.ACSpec(cams=cams, goal_cams=cams, action_dim=c["latent"], discrete_gripper=False,
                        target_entropy=-7.0, finetune_action_decoder=c["finetune_ad"], ac_cams=cams,
                        pr_cams=cams, **base)
    base = dict(n=4, discount=0.99, actor_lr=1e-4, critic_lr=3e-4, deterministic_backup=False,
                reward_scale=10.0, bc_epochs=5, with_lagrange=True)
    base.update({("n" if k == "n_action_samples" else k): v for k, v in ov.items()})
    return O.ACSpec(cams=cams, goal_cams=cams, action_dim=7, discrete_gripper=True,
                    target_entropy=-7.0, **base)


def check_stats(got_named, expected_stats, rtol, atol=0.0, what=""):
    """Compare tensors against (l2, sum, 16 samples) fingerprints.
    ``atol`` (per element) absorbs Adam's g/(|g|+eps) amplification on near-zero grads
    when comparing post-step parameters (a few % of one lr-sized update)."""
    bad = []
    for name, exp in expected_stats.items():
        if name not in got_named:
            bad.append(f"{what}{name}: missing")
            continue
        got = synth.tensor_stats(got_named[name])
        scale = max(abs(exp[0]), 1e-30)
        n = got_named[name].numel()
        # l2: relative; sum & samples: absolute against the tensor's l2 scale
        if abs(got[0] - exp[0]) > rtol * scale + atol * np.sqrt(n) + 1e-12:
            bad.append(f"{what}{name}: l2 {got[0]:.8g} vs {exp[0]:.8g}")
        elif abs(got[1] - exp[1]) > rtol * scale * np.sqrt(n) + atol * n + 1e-12:
            bad.append(f"{what}{name}: sum {got[1]:.8g} vs {exp[1]:.8g}")
        elif np.max(np.abs(got[2:] - exp[2:])) > rtol * max(np.max(np.abs(exp[2:])), scale / np.sqrt(n)) + atol + 1e-12:
            bad.append(f"{what}{name}: samples max|d|={np.max(np.abs(got[2:] - exp[2:])):.3g}")
    return bad


def resync_oracle(mod, P, opts):
    """Copy the module's parameters and Adam state (moments, step counters) into the oracle's.

    Multi-step comparisons restart the oracle from the module's own state before every step after the first:
    after ONE Adam step two implementations that agree to 1e-7 on a gradient can differ by 2*lr in the elements
    whose gradient is ~0 (the first update is lr * sign(g)), and such a difference can flip a ReLU gate in the
    next step - a 1e-3..1e-2 event in a small-batch gradient that says nothing about either implementation
    (the fp32 and fp64 evaluations of the oracle differ by as much, scratch/dbg_c4.py).  `opts`: the oracle's
    optimisers by name (make_opts) or a single oracle Adam (PlayLMP)."""
    import torch

    sd = mod.state_dict()
    with torch.no_grad():
        for k, v in P.items():
            if k in sd and sd[k].dtype == torch.float32:
                v.copy_(sd[k].detach().cpu())
    name_of = {id(p): n for n, p in mod.named_parameters()}
    mopts = mod.configure_optimizers()
    mopts = list(mopts) if isinstance(mopts, (list, tuple)) else [mopts]
    for o in mopts:
        tgt = opts[o.name] if isinstance(opts, dict) else opts
        for blk, p, m, v in o._triples():
            n = name_of[id(p)]
            tgt.m[n], tgt.v[n] = m.detach().cpu().clone().contiguous(), v.detach().cpu().clone().contiguous()
            tgt.t = int(blk.step.item())


def gradient_floor(grad_fn, P, ref_grads, runs=4, eps=1e-7):
    """Per-tensor reproducibility of a gradient evaluated with bf16 operand rounding: re-evaluate it with every
    parameter perturbed by a relative N(0, eps) noise (eps = 1e-7: one fp32 ulp, what a different summation order
    does upstream of a rounding point) and return {name: worst relative change}.  A rounding to bf16 is a step
    function, so a 1-ulp change that crosses a boundary moves that operand by 2^-8; through the 30 sequential layers
    of the RNN's BPTT, or the (dp - <p, dp>) cancellation of the soft-argmax temperature, this is a 1-2.5 % effect
    on small-batch gradients (scratch/chaos_floor.py).  No bf16 kernel can be held tighter than this floor.
    grad_fn(P_perturbed) -> {name: grad}; it must not modify its argument's originals."""
    import torch

    floor = {}
    for r in range(runs):
        gen = torch.Generator().manual_seed(1000 + r)
        Pp = {k: (v.detach() * (1 + eps * torch.randn(v.shape, generator=gen))).requires_grad_(v.requires_grad)
              for k, v in P.items()}
        got = grad_fn(Pp)
        for k, g0 in ref_grads.items():
            if k in got and g0 is not None and g0.norm() > 0:
                e = ((got[k] - g0).norm() / g0.norm()).item()
                floor[k] = max(floor.get(k, 0.0), e)
    return floor


def record_margin(tensor, err, tol, floor=None, kind="grad"):
    """One line per (test, tensor) comparison into gpurun_out/parity_margins.jsonl (or $TACORL_MARGINS): the measured
    error, the tolerance it was held to and, where the tolerance was widened, the reproducibility floor that widened
    it - scratch/margins_md.py turns the file into profiles/r0N_parity_margins.md, so that a pass at 1.1e-2 and a pass at
    2.4e-1 can be told apart.  Best effort: never fails a test."""
    try:
        path = os.environ.get("TACORL_MARGINS")
        if path is None:
            d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
            if not os.path.isdir(d):
                return
            path = os.path.join(d, "parity_margins.jsonl")
        test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
        with open(path, "a") as f:
            f.write(json.dumps({"test": test, "kind": kind, "tensor": tensor, "err": float(err), "tol": float(tol),
                                "floor": None if floor is None else float(floor)}) + "\n")
    except Exception:
        pass
