"""Checkpoint interchange with the reference on the host (utils/networks.py:90-142): a
reference-style PlayLMP run directory - Hydra config with unresolved interpolations, PL checkpoints - is found,
resolved and loaded into the HIP-side PlayLMP (built on the CPU: parameters are host-testable)."""
import os

import pytest
import torch

from tests import cfg_util as C
from tests.golden_util import Golden


def _sd():
    g = Golden("rollout_tacorl")
    return g, C.lmp_state_dict_from_tacorl(g.params())


def test_load_reference_run_directory(tmp_path):
    from tacorl_amd.modules.play_lmp.play_lmp_for_rl import load_play_lmp

    g, sd = _sd()
    other = {k: v + 1.0 for k, v in sd.items()}
    C.write_reference_run_dir(str(tmp_path), sd, latent=16, T=16,
                              extra_ckpts=[("PlayLMP_epoch_1_step_10.ckpt", other), ("PlayLMP_epoch_10_step_100.ckpt", sd)])
    lmp = load_play_lmp(str(tmp_path), device="cpu")  # last.ckpt; `${latent_plan_dim}` etc. arrive resolved
    assert lmp.pr.latent_plan_dim == 16 and lmp.pr.T_max == 16 and lmp.ad.P == 16
    got = lmp.state_dict()
    missing = [k for k in sd if k not in got]
    assert not missing, missing
    assert all(torch.equal(got[k].cpu(), sd[k]) for k in sd)
    # epoch N means exactly N (the reference parses the integer after "epoch"): 1 must not pick epoch_10
    lmp1 = load_play_lmp(str(tmp_path), epoch=1, device="cpu")
    k0 = next(iter(sd))
    assert torch.equal(lmp1.state_dict()[k0].cpu(), other[k0])
    lmp10 = load_play_lmp(str(tmp_path), epoch=10, device="cpu")
    assert torch.equal(lmp10.state_dict()[k0].cpu(), sd[k0])
    # a .ckpt path works as well (utils/networks.py:104-107); anything else is refused
    lmpf = load_play_lmp(str(tmp_path / "model_ckpts" / "last.ckpt"), device="cpu")
    assert torch.equal(lmpf.state_dict()[k0].cpu(), sd[k0])
    with pytest.raises(ValueError):
        load_play_lmp(str(tmp_path / ".hydra" / "config.yaml"), device="cpu")


def test_interpolation_resolver_without_omegaconf(tmp_path):
    from tacorl_amd.modules.common import load_resolved_yaml

    p = tmp_path / "config.yaml"
    p.write_text("a: 3\nb:\n  c: ${a}\n  d: x_${a}_${b.c}\n  e: [1, '${b.c}']\nf: ${b}\n")
    cfg = load_resolved_yaml(str(p))
    assert cfg["b"] == {"c": 3, "d": "x_3_3", "e": [1, 3]} and cfg["f"] == cfg["b"]
    p.write_text("a: ${oc.env:HOME}\n")
    try:
        import omegaconf  # noqa: F401
    except ImportError:
        with pytest.raises(NotImplementedError):
            load_resolved_yaml(str(p))


def test_tacorl_builds_from_a_run_directory(tmp_path):
    """TACORL(play_lmp_dir=...) - the reference's default construction path (tacorl.py:44-53) - on the CPU."""
    from tacorl_amd.modules.tacorl.tacorl import TACORL

    g, sd = _sd()
    C.write_reference_run_dir(str(tmp_path), sd)
    strip = lambda c: {k: v for k, v in c.items() if k not in ("_target_", "_recursive_")}  # noqa: E731
    mod = TACORL(play_lmp_dir=str(tmp_path), **strip(C.tacorl_cfg(device="cpu", finetune_action_decoder=False)))
    assert sorted(n for n, _ in mod.named_parameters()) == sorted(g.names)
    got = mod.state_dict()
    for k in ("perceptual_encoder.networks.rgb_static.model.0.weight", "plan_recognition.fc.weight",
              "action_decoder.rnn.weight_hh_l1"):
        assert torch.equal(got[k].cpu(), sd[k]), k
    # the actor starts as the LMP's encoder + goal encoder + plan proposal (tacorl.py:63-70)
    assert torch.equal(got["actor.actor.policy.fc_mean.weight"].cpu(), sd["plan_proposal.policy.fc_mean.weight"])
    assert torch.equal(got["actor.encoder.networks.rgb_static.model.2.weight"].cpu(),
                       sd["perceptual_encoder.networks.rgb_static.model.2.weight"])
