"""Generate tests/golden/*.npz by running the UNMODIFIED reference on CPU.

Run in the build container only (needs /root/reference):
    python oracle/gen_golden.py            # all cases
    python oracle/gen_golden.py tacorl_q   # one case

A fixture holds: the case config + seeds (inputs and parameters are re-derived
from the seed with tacorl_amd.synth), the recorded noise tape, every scalar the
reference logged, the sampled latent plans, and per-parameter fingerprints
(l2, sum, 16 samples - synth.tensor_stats) of the gradients seen by each
optimiser and of the parameters after each step.  No reference source text is
stored.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as H  # noqa: E402
from tacorl_amd import synth  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # TACORL, frozen LMP (BASELINE config 2 shape, tiny batch), Q phase
    "tacorl_q": dict(kind="tacorl", B=3, T=16, cams={"rgb_static": (84, 84)}, latent=16,
                     epoch=5, finetune_ad=False, steps=2, seed=11),
    # TACORL reference default (AD fine-tune on), BC phase
    "tacorl_bc_ad": dict(kind="tacorl", B=2, T=16, cams={"rgb_static": (84, 84)}, latent=16,
                         epoch=0, finetune_ad=True, steps=2, seed=12),
    # dual camera real-world-like (BASELINE config 4 shape): latent 32, 128^2 + 84^2
    "tacorl_dualcam": dict(kind="tacorl", B=2, T=8, latent=32, epoch=5, finetune_ad=False,
                           cams={"rgb_static": (128, 128), "rgb_gripper": (84, 84)},
                           steps=1, seed=13, overrides=dict(deterministic_backup=False)),
    # BASELINE config 4 as stated: both cameras 128x128, window 32 (config/datamodule/play_lmp_real_world.yaml:17-18),
    # latent plan 32 (config/experiment/play_lmp_real_world.yaml:10), real-world entropy backup
    "tacorl_c4": dict(kind="tacorl", B=2, T=32, latent=32, epoch=5, finetune_ad=False,
                      cams={"rgb_static": (128, 128), "rgb_gripper": (128, 128)}, steps=2, seed=17),
    # BASELINE config 3 as stated: the full default TACORL step - Q phase WITH action-decoder fine-tuning
    "tacorl_q_ad": dict(kind="tacorl", B=2, T=16, cams={"rgb_static": (84, 84)}, latent=16,
                        epoch=5, finetune_ad=True, steps=2, seed=18),
    # flat CQL baseline, discrete gripper (BASELINE config 5 shape)
    "cql_q": dict(kind="cql", B=3, cams={"rgb_static": (84, 84)}, epoch=5, steps=2, seed=14),
    # BASELINE config 5 as stated: 32 action samples in the logsumexp
    "cql_n32": dict(kind="cql", B=3, cams={"rgb_static": (84, 84)}, epoch=5, steps=2, seed=19,
                    overrides=dict(n_action_samples=32)),
    "cql_bc": dict(kind="cql", B=3, cams={"rgb_static": (84, 84)}, epoch=0, steps=1, seed=15,
                   overrides=dict(n_action_samples=2)),
    # PlayLMP seq-VAE step (BASELINE config 1 shape)
    "playlmp": dict(kind="playlmp", B=3, T=16, cams={"rgb_static": (84, 84)}, latent=16,
                    steps=2, seed=16),
    # PlayLMP as the reference trains it: plan-recognition dropout 0.1 in train mode
    # (config/networks/plan_recognition/transformer.yaml:9); the keep masks are part of the noise tape
    # Rollout-time surface (evaluation/rollout_manager.py:310-431): deterministic and sampled
    # actor.get_actions, perceptual_encoder.get_state_from_observation, three action_decoder.act steps with the
    # carried hidden state
    "rollout_tacorl": dict(kind="rollout", B=2, T=4, cams={"rgb_static": (84, 84)}, latent=16, seed=21),
    "rollout_cql": dict(kind="rollout_cql", B=3, cams={"rgb_static": (84, 84)}, seed=22),
    "playlmp_dropout": dict(kind="playlmp", B=3, T=16, cams={"rgb_static": (84, 84)}, latent=16,
                            steps=2, seed=20, dropout_p=0.1),
    # validation_step (tacorl.py:275-287, cql_offline_lightning.py:234-236, play_lmp_for_rl.py:319-348): the same
    # computation with optimize=False, eval mode, no_grad - every `validation/*` scalar (q1_data is what
    # config/callbacks/checkpoint/rl_real_world.yaml:5 monitors) and the fact that nothing moves
    "val_tacorl": dict(kind="tacorl", B=3, T=16, cams={"rgb_static": (84, 84)}, latent=16, epoch=5, finetune_ad=False,
                       steps=1, seed=31, validate=True),
    "val_tacorl_ad": dict(kind="tacorl", B=2, T=16, cams={"rgb_static": (84, 84)}, latent=16, epoch=5, finetune_ad=True,
                          steps=1, seed=32, validate=True),
    "val_cql": dict(kind="cql", B=3, cams={"rgb_static": (84, 84)}, epoch=5, steps=1, seed=33, validate=True),
    "val_playlmp": dict(kind="playlmp", B=3, T=16, cams={"rgb_static": (84, 84)}, latent=16, steps=1, seed=34,
                        validate=True),
    # FREE-RUNNING trajectories (round 5): many optimiser steps of the reference from one initial state with nothing
    # re-synchronised in between - Adam's bias-correction counters and moments, Polyak drift of the targets, and the
    # BC -> Q switch of the actor loss when current_epoch reaches bc_epochs (cql_offline_lightning.py:459-466) in the
    # middle of the run.  Per step: the logged scalars, the latent plan and the noise tape; parameter fingerprints at
    # the steps listed in `param_steps`; torch's Adam step counters at the end.
    "tacorl_traj": dict(kind="tacorl", B=4, T=16, cams={"rgb_static": (84, 84)}, latent=16, finetune_ad=True,
                        steps=12, seed=41, epochs=[4] * 6 + [5] * 6, param_steps=[0, 5, 6, 11], traj=True),
    "playlmp_traj": dict(kind="playlmp", B=4, T=16, cams={"rgb_static": (84, 84)}, latent=16, steps=8, seed=42,
                         param_steps=[0, 3, 7], traj=True),
}


def _stats_dict(prefix, named, out):
    for n, t in named:
        out[f"{prefix}/{n}"] = synth.tensor_stats(t)


def _group_of(name):
    if name.startswith("actor."):
        return "actor"
    for g in ("q1", "q2", "action_decoder", "log_alpha_prime", "log_alpha"):
        if name.startswith(g):
            return g
    return None


def run_rollout(name, c):
    """Inference goldens: no optimiser, modules in eval mode, under no_grad."""
    torch.manual_seed(c["seed"])
    out, cams = {}, c["cams"]
    names = tuple(sorted(cams))
    tape = H.NoiseTape()
    if c["kind"] == "rollout":
        lmp = H.build_play_lmp(cams=names, latent_plan_dim=c["latent"], seq_len=16)
        mod = H.build_tacorl(lmp, finetune_action_decoder=False)
        synth.fill_params_(mod, c["seed"])
        mod.eval()
        batch = synth.make_play_batch(c["seed"] * 100, c["B"], c["T"], cams)
        obs0 = {"observation": {k: v[:, 0] for k, v in batch["states"].items()}, "goal": batch["goal"]}
        with torch.no_grad(), H.record_noise(tape):
            plan, lp0 = mod.actor.get_actions(obs0, deterministic=True, reparameterize=False)
            plan_s, lp_s = mod.actor.get_actions(obs0, deterministic=False, reparameterize=False)
            mod.action_decoder.clear_hidden_state()
            acts, embs = [], []
            for t in range(3):
                ad_state = mod.perceptual_encoder.get_state_from_observation(
                    observation={k: v[:, t] for k, v in batch["states"].items()}, modalities=mod.action_decoder_modalities)
                embs.append(ad_state.clone())
                acts.append(mod.action_decoder.act(latent_plan=plan, perceptual_emb=ad_state.unsqueeze(1)).clone())
            hidden = mod.action_decoder.hidden_state
        assert float(lp0.abs().max()) == 0.0 and lp0.shape == plan.shape
        out.update(plan=plan.numpy(), plan_sampled=plan_s.numpy(), logpi_sampled=lp_s.numpy(),
                   ad_state=torch.stack(embs).numpy(), actions=torch.stack(acts).numpy(), hidden=hidden.numpy())
    else:
        mod = H.build_cql(cams=names)
        synth.fill_params_(mod, c["seed"])
        mod.eval()
        batch = synth.make_transition_batch(c["seed"] * 100, c["B"], cams)
        obs = batch["observations"]
        with torch.no_grad(), H.record_noise(tape):
            a_det, lp0 = mod.actor.get_actions(obs, deterministic=True, reparameterize=False)
            a_smp, lp_smp = mod.actor.get_actions(obs, deterministic=False, reparameterize=False)
            a_rsm, lp_rsm = mod.actor.get_actions(obs, deterministic=False, reparameterize=True)
        out.update(act_det=a_det.numpy(), act_sample=a_smp.numpy(), logpi_sample=lp_smp.numpy(), act_rsample=a_rsm.numpy(),
                   logpi_rsample=lp_rsm.numpy())
    for i, (kind, t) in enumerate(tape.draws):
        out[f"s0/noise/{i:02d}_{kind}"] = t.numpy()
    out["param_names"] = np.array([n for n, _ in mod.named_parameters()])
    out["param_shapes"] = np.array(json.dumps([list(p.shape) for _, p in mod.named_parameters()]))
    out["param_requires_grad"] = np.array([p.requires_grad for _, p in mod.named_parameters()])
    out["config"] = np.array(json.dumps(dict(c)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"[{name}] wrote {os.path.getsize(os.path.join(OUT, name + '.npz')) / 1e3:.1f} kB; draws: {[k for k, _ in tape.draws]}")


def run_case(name, c):
    if c["kind"].startswith("rollout"):
        return run_rollout(name, c)
    torch.manual_seed(c["seed"])
    torch.set_num_threads(8)
    out = {}
    cams = c["cams"]
    if c["kind"] in ("tacorl", "playlmp"):
        lmp = H.build_play_lmp(cams=tuple(sorted(cams)), latent_plan_dim=c["latent"], seq_len=c["T"],
                               dropout_p=c.get("dropout_p", 0.0))
    if c["kind"] == "tacorl":
        mod = H.build_tacorl(lmp, finetune_action_decoder=c["finetune_ad"], **c.get("overrides", {}))
    elif c["kind"] == "cql":
        mod = H.build_cql(cams=tuple(sorted(cams)), **c.get("overrides", {}))
    else:
        mod = lmp
    synth.fill_params_(mod, c["seed"])
    mod.train()
    if c.get("validate"):
        mod.eval()  # the Trainer's validation loop: eval mode, no_grad
    mod.current_epoch = c.get("epoch", 0)
    names = [n for n, _ in mod.named_parameters()]
    out["param_names"] = np.array(names)
    out["param_shapes"] = np.array(json.dumps([list(p.shape) for _, p in mod.named_parameters()]))
    out["param_requires_grad"] = np.array([p.requires_grad for _, p in mod.named_parameters()])

    for step in range(c["steps"]):
        if c.get("epochs"):
            mod.current_epoch = c["epochs"][step]
        bseed = c["seed"] * 100 + step
        if c["kind"] == "cql":
            batch = synth.make_transition_batch(bseed, c["B"], cams)
        else:
            batch = synth.make_play_batch(bseed, c["B"], c["T"], cams)
        tape = H.NoiseTape()
        mod.logged = {}
        mod.grad_log = []
        if c.get("validate"):
            before = {n: p.detach().clone() for n, p in mod.named_parameters()}
            with torch.no_grad(), H.record_noise(tape):
                if c["kind"] == "tacorl":
                    orig, cap = mod.get_pr_latent_plan, {}

                    def wrapped_v(b, return_emb_states=True, _o=orig, _c=cap):
                        r = _o(b, return_emb_states=return_emb_states)
                        _c["plan"] = (r[0] if return_emb_states else r).detach().clone()
                        return r

                    mod.get_pr_latent_plan = wrapped_v
                    mod.validation_step(batch)
                    mod.get_pr_latent_plan = orig
                    out[f"s{step}/latent_plan"] = cap["plan"].numpy()
                else:
                    mod.validation_step(batch, 0)
            assert all(torch.equal(before[n], p) for n, p in mod.named_parameters()), "validation_step moved a parameter"
            for i, (kind, t) in enumerate(tape.draws):
                out[f"s{step}/noise/{i:02d}_{kind}"] = t.numpy()
            out[f"s{step}/logged"] = np.array(json.dumps(mod.logged))
            print(f"[{name}] validation: " + ", ".join(f"{k}={v:.5g}" for k, v in sorted(mod.logged.items())))
            print(f"[{name}] draws: {[k for k, _ in tape.draws]}")
            continue
        with H.record_noise(tape):
            if c["kind"] == "tacorl":
                # capture the sampled latent plan (north-star parity item)
                orig = mod.get_pr_latent_plan
                cap = {}

                def wrapped(b, return_emb_states=True, _o=orig, _c=cap):
                    r = _o(b, return_emb_states=return_emb_states)
                    _c["plan"] = (r[0] if return_emb_states else r).detach().clone()
                    return r

                mod.get_pr_latent_plan = wrapped
                mod.training_step(batch)
                mod.get_pr_latent_plan = orig
                out[f"s{step}/latent_plan"] = cap["plan"].numpy()
            elif c["kind"] == "cql":
                mod.training_step(batch, 0)
            else:
                opt = mod.optimizers()[0]
                loss = mod.training_step(batch, 0)
                opt.zero_grad()
                mod.manual_backward(loss)
                opt.step()
        for i, (kind, t) in enumerate(tape.draws):
            out[f"s{step}/noise/{i:02d}_{kind}"] = t.numpy()
        out[f"s{step}/logged"] = np.array(json.dumps(mod.logged))
        # grads as each optimiser saw them: the LAST backward that touched a group
        # before its step is the one right after its zero_grad (reference
        # cql_offline_lightning.py:452-454,401-404,520-538; tacorl.py:231-233).
        if c.get("traj"):
            if step in c["param_steps"]:
                _stats_dict(f"s{step}/param", mod.named_parameters(), out)
            print(f"[{name}] step {step} (epoch {mod.current_epoch}): " +
                  ", ".join(f"{k.split('/')[-1]}={v:.5g}" for k, v in sorted(mod.logged.items())))
            continue
        if c["kind"] == "playlmp":
            _stats_dict(f"s{step}/grad", mod.grad_log[0].items(), out)
        else:
            order = (["action_decoder"] if c["kind"] == "tacorl" and c["finetune_ad"] else []) + [
                "log_alpha", "log_alpha_prime", "actor", "q1", "q2"]
            assert len(mod.grad_log) == len(order), (len(mod.grad_log), order)
            for g, gl in zip(order, mod.grad_log):
                _stats_dict(f"s{step}/grad", [(n, t) for n, t in gl.items() if _group_of(n) == g], out)
        _stats_dict(f"s{step}/param", mod.named_parameters(), out)
        print(f"[{name}] step {step}: " + ", ".join(f"{k.split('/')[-1]}={v:.5g}" for k, v in sorted(mod.logged.items())))

    if c.get("traj"):
        # torch.optim.Adam's own step counters after the run, by optimiser (reference order: configure_optimizers)
        opts = mod.optimizers() if c["kind"] != "playlmp" else [mod.optimizers()[0]]
        counts = []
        for o in opts:
            ts = {int(st["step"]) for st in o.state.values() if "step" in st}
            counts.append(sorted(ts))
        out["adam_steps"] = np.array(json.dumps(counts))
        print(f"[{name}] Adam step counters per optimiser: {counts}")
    cfg = {k: v for k, v in c.items()}
    out["config"] = np.array(json.dumps(cfg))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"[{name}] wrote {os.path.getsize(os.path.join(OUT, name + '.npz')) / 1e3:.1f} kB")


if __name__ == "__main__":
    assert H.check_sdpa_standin() < 1e-6, "recording stand-in for scaled_dot_product_attention drifted from torch's"
    which = sys.argv[1:] or list(CASES)
    for n in which:
        run_case(n, CASES[n])
