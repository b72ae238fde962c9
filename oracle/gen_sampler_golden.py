"""tests/golden/play_sampler.npz: what the reference's PlayDataset.__getitem__ (datamodule/dataset/play_dataset.py:
115-169) returns on a tiny synthetic dataset written to a temp dir, with every numpy draw it consumes recorded.
Frame ids are encoded in the pixel values, so the fixture holds, per item: the draws, the frame ids of the (padded)
window, the goal frame id, disp, window_size and the padded actions.  Build container only (needs /root/reference)."""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as H  # noqa: E402

H.install_shims()
from tacorl.datamodule.dataset.play_dataset import PlayDataset  # noqa: E402

EP = [[0, 69], [70, 159]]
N = 160
MIN_WS, MAX_WS = 8, 16
NN = {str(s): [int(x) for x in np.random.RandomState(s).randint(0, N, size=s % 4)] for s in range(N)}


class IdentityTransforms:  # the transform manager is out of this fixture's scope: frames pass through as tensors
    def __call__(self, inp, transf_type="train", device="cpu"):
        return {k: torch.as_tensor(np.asarray(v)).float() for k, v in inp.items()}


def main():
    with tempfile.TemporaryDirectory() as d:
        d = Path(d)
        rs = np.random.RandomState(0)
        acts = rs.uniform(-1, 1, size=(N, 7)).astype(np.float32)
        for i in range(N):
            img = np.zeros((2, 2, 3), np.uint8)
            img[..., 0], img[..., 1] = i % 256, i // 256
            np.savez(d / f"episode_{i:07d}.npz", rgb_static=img, rel_actions_world=acts[i],
                     robot_obs=np.full(15, i, np.float32), scene_obs=np.full(24, i, np.float32))
        np.save(d / "ep_start_end_ids.npy", np.array(EP))
        with open(d / "nn.json", "w") as f:
            json.dump({"train": NN}, f)
        rec = []
        gen = np.random.RandomState(123)
        o_randint, o_choice, o_rng = np.random.randint, np.random.choice, np.random.default_rng

        def randint(low, high=None, *a, **kw):
            v = gen.randint(low, high)
            rec.append(("randint", float(v)))
            return v

        def choice(options, p=None, *a, **kw):
            u = gen.uniform()
            rec.append(("choice_u", u))
            options = list(options)
            if p is None:
                return options[min(int(u * len(options)), len(options) - 1)]
            return options[int(np.searchsorted(np.cumsum(p), u, side="right").clip(0, len(options) - 1))]

        class FakeRng:
            def geometric(self, p):
                v = gen.geometric(p)
                rec.append(("geometric", float(v)))
                return v

        np.random.randint, np.random.choice, np.random.default_rng = randint, choice, lambda *a, **k: FakeRng()
        try:
            out = {}
            for variant, goal_aug in (("plain", False), ("goal_aug", True)):
                ds = PlayDataset(data_dir=d, modalities=["rgb_static", "rel_actions_world"], train=True, real_world=True,
                                 min_window_size=MIN_WS, max_window_size=MAX_WS, pad=True,
                                 transform_manager=IdentityTransforms(), include_goal=True, goal_augmentation=goal_aug,
                                 goal_sampling_prob=0.3, goal_strategy_prob={"geometric": 0.6, "similar_robot_obs": 0.4},
                                 nn_steps_from_step_path=str(d / "nn.json"))
                idxs = gen.randint(0, len(ds), size=48)
                items = []
                for idx in idxs:
                    rec.clear()
                    it = ds[int(idx)]
                    fid = lambda t: (t[..., 0, 0, 0] + 256 * t[..., 0, 0, 1]).numpy().astype(np.int64)  # noqa: E731
                    items.append(dict(idx=int(idx), draws=list(rec), frames=fid(it["states"]["rgb_static"]),
                                      goal=int(fid(it["goal"]["rgb_static"])), disp=int(it["disp"]),
                                      window_size=int(it["window_size"]), actions=it["actions"].numpy()))
                out[f"{variant}/len"] = np.array(len(ds))
                out[f"{variant}/idx"] = np.array([i["idx"] for i in items])
                out[f"{variant}/frames"] = np.stack([i["frames"] for i in items])
                out[f"{variant}/goal"] = np.array([i["goal"] for i in items])
                out[f"{variant}/disp"] = np.array([i["disp"] for i in items])
                out[f"{variant}/window_size"] = np.array([i["window_size"] for i in items])
                out[f"{variant}/actions"] = np.stack([i["actions"] for i in items])
                out[f"{variant}/draws"] = np.array(json.dumps([i["draws"] for i in items]))
        finally:
            np.random.randint, np.random.choice, np.random.default_rng = o_randint, o_choice, o_rng
        out["all_actions"] = acts
        out["ep"] = np.array(EP)
        out["nn"] = np.array(json.dumps(NN))
        out["cfg"] = np.array(json.dumps(dict(min_ws=MIN_WS, max_ws=MAX_WS, p=0.3, strategy={"geometric": 0.6, "similar_robot_obs": 0.4})))
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "play_sampler.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) / 1e3, "kB")


if __name__ == "__main__":
    main()
