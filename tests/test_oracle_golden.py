"""Pin the CPU oracle (oracle/tacorl_oracle.py) against goldens produced by the
unmodified reference (oracle/gen_golden.py).  CPU-only; no reference access needed."""
import numpy as np
import pytest
import torch

from oracle import tacorl_oracle as O
from tests.golden_util import Golden, check_stats, spec_for

PARAM_ATOL = 1e-5  # ~3% of one Adam update (lr 3e-4): near-zero grads flip under g/(|g|+eps)
RTOL = 2e-5  # fp32 re-association (de-duplicated encoders, fused grads) stays below this


def _check_logs(got, exp, rtol=RTOL):
    bad = []
    for k, v in exp.items():
        if k not in got:
            continue
        if abs(got[k] - v) > rtol * max(abs(v), 1e-3):
            bad.append(f"{k}: {got[k]:.8g} vs {v:.8g}")
    return bad


@pytest.mark.parametrize("name", ["tacorl_q", "tacorl_bc_ad", "tacorl_dualcam", "tacorl_c4", "tacorl_q_ad"])
@pytest.mark.parametrize("faithful", [False, True])
def test_tacorl_step_matches_reference(name, faithful):
    if faithful and name != "tacorl_q":
        pytest.skip("faithful schedule checked once")
    g = Golden(name)
    spec = spec_for(g)
    P = O.require_grad_(g.params(), frozen_prefixes=("perceptual_encoder.", "plan_recognition."))
    opts = O.make_opts(P, spec)
    for step in range(g.cfg["steps"]):
        logs, plan, grads = O.tacorl_step(P, opts, spec, g.batch(step), g.noise(step), g.cfg["epoch"],
                                          faithful=faithful)
        exp = g.logged(step)
        assert set(exp) - set(logs) == set(), set(exp) - set(logs)
        bad = _check_logs(logs, exp)
        assert torch.allclose(plan, g.latent_plan(step), rtol=1e-5, atol=1e-6)
        bad += check_stats(grads, g.stats(step, "grad"), rtol=5e-5, what="grad ")
        bad += check_stats(P, g.stats(step, "param"), rtol=RTOL, atol=PARAM_ATOL, what="param ")
        assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("name", ["cql_q", "cql_bc", "cql_n32"])
def test_cql_step_matches_reference(name):
    g = Golden(name)
    spec = spec_for(g)
    P = O.require_grad_(g.params())
    opts = O.make_opts(P, spec)
    for step in range(g.cfg["steps"]):
        logs, grads = O.cql_step(P, opts, spec, g.batch(step), g.noise(step), g.cfg["epoch"])
        bad = _check_logs(logs, g.logged(step))
        bad += check_stats(grads, g.stats(step, "grad"), rtol=5e-5, what="grad ")
        bad += check_stats(P, g.stats(step, "param"), rtol=RTOL, atol=PARAM_ATOL, what="param ")
        assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("name", ["playlmp", "playlmp_dropout"])
def test_playlmp_step_matches_reference(name):
    g = Golden(name)
    P = O.require_grad_(g.params())
    opt = O.Adam([n for n in P], 1e-4)
    for step in range(g.cfg["steps"]):
        logs, grads = O.playlmp_step(P, opt, g.batch(step), g.noise(step), sorted(g.cams),
                                     dropout_p=g.cfg.get("dropout_p", 0.0))
        bad = _check_logs(logs, g.logged(step))
        bad += check_stats(grads, g.stats(step, "grad"), rtol=5e-5, what="grad ")
        bad += check_stats(P, g.stats(step, "param"), rtol=RTOL, atol=PARAM_ATOL, what="param ")
        assert not bad, "\n".join(bad[:20])


def test_golden_param_layout_is_reference_layout():
    """State-dict key layout the boundary must keep (SURVEY 8a note 9)."""
    g = Golden("tacorl_q")
    names = set(g.names)
    for k in ("actor.actor.policy.fc_layers.0.weight", "actor.encoder.networks.rgb_static.model.0.weight",
              "actor.encoder.networks.rgb_static.model.6.temperature", "q1.critic.Q.out.weight",
              "target_q2.goal_encoder.mlp.4.bias", "perceptual_encoder.networks.rgb_static.fc_layers.3.weight",
              "plan_recognition.transformer_encoder.layers.1.linear2.weight", "action_decoder.rnn.weight_hh_l1",
              "log_alpha", "log_alpha_prime"):
        assert k in names, k
    assert sum(int(np.prod(s)) for n, s in zip(g.names, g.shapes) if n.startswith("q1.")) == 352226
