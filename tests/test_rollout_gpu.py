"""SURVEY 8f N4 on the GPU: the rollout-time surface of the modules against goldens recorded from the reference
(evaluation/rollout_manager.py:310-431 call sequence), with the TACORL module built the reference's default way -
from a PlayLMP run directory (Hydra config + PL checkpoint, utils/networks.py:90-142)."""
import pytest
import torch

from tests import cfg_util as C
from tests.golden_util import Golden

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("compute,tol", [("f32", 1e-4), ("bf16", 3e-2)])
def test_tacorl_rollout_from_reference_checkpoint(tmp_path, compute, tol):
    from tacorl_amd.modules.tacorl.tacorl import TACORL

    g = Golden("rollout_tacorl")
    z, params = g.z, g.params()
    C.write_reference_run_dir(str(tmp_path), C.lmp_state_dict_from_tacorl(params))
    strip = lambda c: {k: v for k, v in c.items() if k not in ("_target_", "_recursive_")}  # noqa: E731
    mod = TACORL(play_lmp_dir=str(tmp_path), compute_dtype=compute, image_dtype=compute,
                 **strip(C.tacorl_cfg(device="cuda:0", finetune_action_decoder=False)))
    missing, unexpected = mod.load_state_dict(params, strict=False)
    bufs = ("one_hot_embedding_eye", "ones", "gripper_bounds", "action_max_bound", "action_min_bound")
    assert not unexpected and all(m.endswith(bufs) for m in missing), (missing, unexpected)
    mod.eval()
    batch, tape = g.batch(0), g.tape(0)
    obs0 = {"observation": {c: v[:, 0] for c, v in batch["states"].items()}, "goal": batch["goal"]}
    plan, lp = mod.actor.get_actions(obs0, deterministic=True, reparameterize=False)
    assert plan.shape == (2, 16) and lp.shape == plan.shape and float(lp.abs().max()) == 0.0
    assert _rel(plan, z["plan"]) < tol, _rel(plan, z["plan"])
    ps, lps = mod.actor.get_actions(obs0, deterministic=False, reparameterize=False, noise={"eps": tape[0][1]})
    assert lps.shape == (2, 1) and _rel(ps, z["plan_sampled"]) < tol and _rel(lps, z["logpi_sampled"]) < 10 * tol
    # the same plan as get_pr_latent_plan's encoder path sees it: the frozen LMP encoder's embeddings
    mod.action_decoder.clear_hidden_state()
    assert mod.ad.hidden_state is None
    plan_ref = torch.from_numpy(z["plan"]).to(mod.device)  # decode from the golden plan: errors do not compound
    for t in range(3):
        st = mod.perceptual_encoder.get_state_from_observation(
            observation={c: v[:, t] for c, v in batch["states"].items()}, modalities=mod.action_decoder_modalities)
        assert st.shape == (2, 32) and _rel(st, z["ad_state"][t]) < tol
        a = mod.action_decoder.act(latent_plan=plan_ref, perceptual_emb=st.unsqueeze(1),
                                   noise=(tape[1 + 2 * t][1], tape[2 + 2 * t][1]))
        assert a.shape == (2, 1, 7)
        # the gripper command is a class (+-1): exact; the sampled arm action through exp(log_scale) * logit(u)
        assert torch.equal(a[..., -1].cpu(), torch.from_numpy(z["actions"][t])[..., -1]), t
        assert _rel(a[..., :-1], z["actions"][t][..., :-1]) < (10 * tol if compute == "f32" else 0.2), (t, _rel(a, z["actions"][t]))
    assert _rel(mod.ad.hidden_state, z["hidden"]) < tol
    # a cleared state starts a new plan from h_0 = 0: step 0 is reproduced
    mod.action_decoder.clear_hidden_state()
    st = mod.perceptual_encoder.get_state_from_observation({c: v[:, 0] for c, v in batch["states"].items()},
                                                           mod.action_decoder_modalities)
    a0 = mod.action_decoder.act(plan_ref, st.unsqueeze(1), noise=(tape[1][1], tape[2][1]))
    assert _rel(a0[..., :-1], z["actions"][0][..., :-1]) < (10 * tol if compute == "f32" else 0.2)


def test_cql_rollout_discrete_gripper():
    from tacorl_amd.lightning import instantiate

    g = Golden("rollout_cql")
    z = g.z
    mod = instantiate(C.cql_cfg(device="cuda:0"))
    mod.load_state_dict(g.params())
    mod.eval()
    obs, tape = g.batch(0)["observations"], g.tape(0)
    a, lp = mod.actor.get_actions(obs, deterministic=True)
    assert a.shape == (3, 7) and float(lp.abs().max()) == 0.0
    assert _rel(a, z["act_det"]) < 1e-4 and torch.equal(a[:, -1].cpu().abs(), torch.ones(3))
    a, lp = mod.actor.get_actions(obs, deterministic=False, reparameterize=False, noise={"eps": tape[0][1], "gumbel_u": tape[1][1]})
    assert _rel(a, z["act_sample"]) < 1e-4 and _rel(lp, z["logpi_sample"]) < 1e-4
    a, lp = mod.actor.get_actions(obs, deterministic=False, reparameterize=True, noise={"eps": tape[2][1], "gumbel_u": tape[3][1]})
    assert _rel(a, z["act_rsample"]) < 1e-4 and _rel(lp, z["logpi_rsample"]) < 1e-4
    # without injected noise: fresh draws, still a valid action
    a, lp = mod.actor.get_actions(obs)
    assert torch.isfinite(a).all() and torch.isfinite(lp).all() and bool((a[:, :6].abs() <= 1).all())


def test_state_from_observation_two_cameras_of_equal_geometry():
    """Two cameras of the same size share the encoder runner's (n, H, W) buffers: the fused state must still be
    [enc_static(static) | enc_gripper(gripper)], not the last camera twice (C4: both cameras 128x128)."""
    from tests.test_step_gpu import build_tacorl, to_dev

    g = Golden("tacorl_c4")
    mod = build_tacorl(g)
    mod.load_state_dict(g.params(), strict=False)
    mod.eval()
    cams = sorted(g.cams)
    assert len(cams) == 2
    states = to_dev(g.batch(0), mod.device)["states"]
    obs = {c: states[c][:, 0] for c in cams}
    assert obs[cams[0]].shape == obs[cams[1]].shape
    both = mod.perceptual_encoder.get_state_from_observation(obs, cams)
    one = [mod.perceptual_encoder.get_state_from_observation({c: obs[c]}, [c]) for c in cams]
    assert both.shape == (obs[cams[0]].shape[0], 64)
    assert torch.equal(both, torch.cat(one, dim=-1))
    assert not torch.equal(one[0], one[1])
    # the actor's [obs | goal] representation runs obs and goal through the same runner as well
    rep = mod.actor.get_emb_representation({"observation": obs, "goal": to_dev(g.batch(0), mod.device)["goal"]})
    rep_obs_only = mod.actor.get_emb_representation({"observation": obs})
    assert torch.equal(rep[:, :64], rep_obs_only)
