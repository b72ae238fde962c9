"""tacorl_amd.data.replay.PlayIndex against what the reference's PlayDataset.__getitem__ returned for the same
draws (tests/golden/play_sampler.npz, recorded by oracle/gen_sampler_golden.py): window frames with pad-by-repetition,
variable window sizes, geometric / similar-robot-obs hindsight goals, disp, zero-padded actions with the gripper
action repeated (reference play_dataset.py:115-169,258-310)."""
import json
import os

import numpy as np
import pytest

from tacorl_amd.data.replay import GEOMETRIC, SIMILAR, PlayIndex, pad_actions

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "play_sampler.npz"))
CFG = json.loads(str(G["cfg"]))


def _draws(rec, index, goal_aug):
    """Name the reference's recorded numpy draws of one item (order: window size, strategy, then per strategy)."""
    it = iter(rec)
    d = {"window_size": int(next(it)[1])}
    u = next(it)[1]
    d["strategy"] = GEOMETRIC if u < CFG["strategy"]["geometric"] else SIMILAR
    d.update(disp=1, noise_step=0, u_choice=0.0, u_random_state=0.0)
    if d["strategy"] == GEOMETRIC:
        d["disp"] = int(next(it)[1])
        if goal_aug:
            d["noise_step"] = int(next(it)[1]) - 1
    else:
        kind, u2 = next(it)
        d["u_choice"] = d["u_random_state"] = u2  # one uniform: picks a neighbour, or a random state when there is none
    assert next(it, None) is None
    return d


@pytest.mark.parametrize("variant,goal_aug", [("plain", False), ("goal_aug", True)])
def test_play_index_matches_reference_dataset(variant, goal_aug):
    ix = PlayIndex(G["ep"], CFG["min_ws"], CFG["max_ws"], goal_sampling_prob=CFG["p"], goal_strategy_prob=CFG["strategy"],
                   goal_augmentation=goal_aug, nn_steps_from_step=json.loads(str(G["nn"])))
    assert len(ix) == int(G[f"{variant}/len"])
    recs = json.loads(str(G[f"{variant}/draws"]))
    ds = [_draws(r, ix, goal_aug) for r in recs]
    draws = {k: np.array([d[k] for d in ds]) for k in ds[0]}
    s = ix.sample(G[f"{variant}/idx"], draws)
    assert np.array_equal(s["window_size"], G[f"{variant}/window_size"])
    assert np.array_equal(s["frames"], G[f"{variant}/frames"])
    assert np.array_equal(s["goal"], G[f"{variant}/goal"])
    assert np.array_equal(s["disp"], G[f"{variant}/disp"])
    assert np.array_equal(pad_actions(G["all_actions"], s["frames"], s["padded"]), G[f"{variant}/actions"])
    # both strategies and both pad cases occur in the fixture
    assert (s["disp"] == -1).any() and (s["disp"] > 0).any() and s["padded"].any() and (~s["padded"][:, -1]).any()


def test_draws_follow_the_reference_distributions():
    ix = PlayIndex([[0, 999]], 8, 16, goal_sampling_prob=0.3)
    d = ix.draw(20000, np.random.default_rng(0))
    assert d["window_size"].min() == 8 and d["window_size"].max() == 16
    assert abs(d["strategy"].mean() - 0.5) < 0.02 and abs(d["disp"].mean() - 1 / 0.3) < 0.1
    assert set(np.unique(d["noise_step"])) == {-1, 0, 1}
