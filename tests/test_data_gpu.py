"""SURVEY 8f N2 / N3 on the GPU: the augmenting pack kernel against the CPU restatement of the reference's train
pipeline (and, for RandomShiftsAug, against the reference class's own recorded outputs), the HBM-resident replay's
gather, the pinned-host feeder, and a TACORL step fed by them."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pack(frames, shift, jitter, pad, dtype):
    from tacorl_amd import _lib, ops

    n, H, W, _ = frames.shape
    out = torch.full((n, H, W, 3), float("nan"), device=DEV, dtype=dtype)
    ops.pack_images_u8_aug_batch([(frames.data_ptr(), H * W * 3, out.data_ptr(), n, shift, jitter)],
                                 _lib.BF16 if dtype == torch.bfloat16 else _lib.F32, H, W, pad)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("hw,pad", [(84, 4), (128, 6), (20, 2)])
def test_augment_pack_matches_oracle(hw, pad):
    from oracle import augment_oracle as A
    from tacorl_amd import _lib
    from tacorl_amd.data.augment import AugmentSpec

    _lib.call("tacorl_hip_init", 0)
    g = torch.Generator(device=DEV).manual_seed(hw)
    n = 9
    frames = torch.randint(0, 256, (n, hw, hw, 3), device=DEV, dtype=torch.uint8, generator=g)
    frames[1] = 200  # a constant grey frame: hue undefined, contrast mean = the value
    shift, jitter = AugmentSpec(pad=pad, brightness=0.3, contrast=0.3, hue=0.3).draw(n, DEV, g)
    jitter[2, 7] = 0.0  # ColorTransform(prob<1): an image that keeps its colours
    ref = A.train_pipeline(frames.cpu(), shift.cpu(), jitter.cpu(), pad)
    got = _pack(frames, shift, jitter, pad, torch.float32)
    assert torch.isfinite(got).all()
    # hue runs through a piecewise (sextant) map: an image value that lands exactly on a sextant border may take the
    # neighbouring branch under 1-ulp differences - both branches agree there, so the error stays at rounding level
    assert (got.cpu() - ref).abs().max().item() < 2e-5, (got.cpu() - ref).abs().max().item()
    got16 = _pack(frames, shift, jitter, pad, torch.bfloat16)
    assert (got16.float().cpu() - ref.to(torch.bfloat16).float()).abs().max().item() <= 2.0 ** -7  # one bf16 ulp below 1
    # stages off: shift only / nothing = the plain normalising pack
    plain = _pack(frames, None, None, pad, torch.float32)
    assert torch.equal(plain.cpu(), A.train_pipeline(frames.cpu()))
    only_shift = _pack(frames, shift, None, pad, torch.float32)
    assert torch.equal(only_shift.cpu(), A.train_pipeline(frames.cpu(), shift.cpu(), None, pad))


def test_random_shift_kernel_matches_reference_outputs():
    """RandomShiftsAug of the reference itself (tests/golden/augment.npz) -> /255 -> Normalize."""
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz"))
    for tag in ("a", "b", "c"):
        n, hw, pad, seed = (int(v) for v in G[f"{tag}/cfg"])
        frames = torch.from_numpy(np.random.RandomState(seed).randint(0, 256, size=(n, hw, hw, 3)).astype(np.uint8)).to(DEV)
        got = _pack(frames, torch.from_numpy(G[f"{tag}/shift"]).to(DEV), None, pad, torch.float32)
        ref = (torch.from_numpy(G[f"{tag}/out"]) / 255.0 - 0.5) / 0.5
        assert (got.cpu() - ref).abs().max().item() < 2e-4  # the reference's bilinear grid weights (~1e-6 of 255)


def _dataset(N=400, hw=84):
    g = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 256, (N, hw, hw, 3), dtype=torch.uint8, generator=g)
    acts = np.random.RandomState(4).uniform(-1, 1, size=(N, 7)).astype(np.float32)
    acts[:, -1] = np.where(acts[:, -1] >= 0, 1.0, -1.0)
    return frames, acts


def test_hbm_and_pinned_replay_feed_the_step():
    from tacorl_amd.data.augment import AugmentSpec, draw_play_batch_augmentation
    from tacorl_amd.data.replay import HbmReplay, PinnedReplay, PlayIndex, pad_actions
    from tests.test_fullsize_gpu import _mod

    frames, acts = _dataset()
    ix = PlayIndex([[0, 199], [200, 399]], 16, 16, goal_sampling_prob=0.3)
    rng = np.random.default_rng(0)
    B, T = 16, 16
    idx = rng.integers(len(ix), size=B)
    draws = ix.draw(B, rng)
    s = ix.sample(idx, draws)
    hbm = HbmReplay({"rgb_static": frames}, acts, ix, device=DEV)
    b = hbm.batch(idx, draws)
    torch.cuda.synchronize()
    assert torch.equal(b["states"]["rgb_static"].cpu(), frames[torch.from_numpy(s["frames"])])
    assert torch.equal(b["goal"]["rgb_static"].cpu(), frames[torch.from_numpy(s["goal"])])
    assert np.array_equal(b["actions"].cpu().numpy(), pad_actions(acts, s["frames"], s["padded"]))
    assert np.array_equal(b["disp"].cpu().numpy(), s["disp"])
    for how in ("host", "device"):  # host gather + H2D copy / GPU gather out of the pinned pages over PCIe
        pin = PinnedReplay({"rgb_static": frames}, acts, ix, DEV, gather=how)
        pin.prefetch(idx, draws)
        bp = pin.next()
        torch.cuda.synchronize()
        for k in ("states", "goal"):
            assert torch.equal(bp[k]["rgb_static"], b[k]["rgb_static"]), how
        assert torch.equal(bp["actions"], b["actions"]) and torch.equal(bp["disp"], b["disp"])
    # the step takes the feeder's batch as it is; without augmentation it is the plain uint8 route bit for bit
    ma, mb = _mod("bf16"), _mod("bf16")
    torch.manual_seed(7); torch.cuda.manual_seed(7)
    ma.training_step(b)
    torch.manual_seed(7); torch.cuda.manual_seed(7)
    mb.training_step({k: ({c: t.clone() for c, t in v.items()} if isinstance(v, dict) else v) for k, v in bp.items()})
    torch.cuda.synchronize()
    assert ma.logged == mb.logged and all(v == v for v in ma.logged.values())
    # with augmentation: finite, different from the plain step, reproducible for the same draws
    g = torch.Generator(device=DEV).manual_seed(11)
    aug = draw_play_batch_augmentation({"rgb_static": AugmentSpec(pad=4)}, B, T, DEV, g)
    outs = []
    for _ in range(2):
        m = _mod("bf16")
        torch.manual_seed(7); torch.cuda.manual_seed(7)
        m.training_step(dict(b, aug=aug))
        torch.cuda.synchronize()
        outs.append(dict(m.logged))
    assert outs[0] == outs[1] and all(v == v for v in outs[0].values()) and outs[0] != ma.logged


def test_fused_replay_batch_equals_gathered_batch():
    """HbmReplay.batch(fused=True): the module's image pack reads the frames by index out of the dataset
    (tacorl_pack_images_u8_gather_batch) - bit for bit the step of the gathered uint8 batch, for TACORL and PlayLMP,
    plain and augmented."""
    from tacorl_amd.data.augment import AugmentSpec, draw_play_batch_augmentation
    from tacorl_amd.data.replay import HbmReplay, PlayIndex
    from tacorl_amd.modules.play_lmp.play_lmp_for_rl import PlayLMP
    from tests import cfg_util as C
    from tests.test_fullsize_gpu import _mod

    frames, acts = _dataset()
    ix = PlayIndex([[0, 199], [200, 399]], 16, 16, goal_sampling_prob=0.3)
    rng = np.random.default_rng(1)
    B, T = 16, 16
    idx, draws = rng.integers(len(ix), size=B), ix.draw(B, rng)
    hbm = HbmReplay({"rgb_static": frames}, acts, ix, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(12)
    aug = draw_play_batch_augmentation({"rgb_static": AugmentSpec(pad=4)}, B, T, DEV, g)
    strip = lambda c: {k: v for k, v in c.items() if k not in ("_target_", "_recursive_")}  # noqa: E731

    def tacorl():
        return _mod("bf16")

    def playlmp():
        torch.manual_seed(3)
        return PlayLMP(**strip(C.playlmp_cfg(device="cuda:0", compute_dtype="bf16", image_dtype="bf16")))

    for build, args in ((tacorl, ()), (playlmp, (0,))):
        for a in (None, aug):
            outs, imgs = [], []
            for fused in (False, True):
                m = build()
                b = hbm.batch(idx, draws, aug=a, fused=fused)
                assert ("replay" in b) == fused and ("states" in b) != fused
                torch.manual_seed(7); torch.cuda.manual_seed(7)
                m.training_step(b, *args)
                torch.cuda.synchronize()
                outs.append(dict(m.logged))
                imgs.append({c: t.clone() for c, t in m.frames.items()})
            assert outs[0] == outs[1] and all(v == v for v in outs[0].values()), (build.__name__, a is not None, outs)
            assert all(torch.equal(imgs[0][c], imgs[1][c]) for c in imgs[0]), "window frames differ"


def _resize_pack(frames, size, shift, jitter, pad, dtype=torch.float32):
    from tacorl_amd import _lib, ops

    n, Hs, Ws, _ = frames.shape
    H, W = size
    out = torch.full((n, H, W, 3), float("nan"), device=DEV, dtype=dtype)
    ops.pack_images_u8_resize_aug_batch([(frames.data_ptr(), Hs * Ws * 3, out.data_ptr(), n, None, 1, shift, jitter)],
                                        _lib.BF16 if dtype == torch.bfloat16 else _lib.F32, (Hs, Ws), H, W, pad)
    torch.cuda.synchronize()
    return out


def test_resize_kernel_matches_interpolate_and_oracle():
    """N3's first stage, torchvision Resize (rl_train.yaml:3-4,16-17): the kernel against F.interpolate's own outputs
    (tests/golden/resize.npz) and the whole pipeline Resize -> shift -> /255 -> ColorJitter -> Normalize against the oracle."""
    from oracle import augment_oracle as A
    from tacorl_amd import _lib
    from tacorl_amd.data.augment import AugmentSpec

    _lib.call("tacorl_hip_init", 0)
    R = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resize.npz"))
    for tag in ("static", "gripper", "rect", "up"):
        n, hs, ws, ht, wt, seed = (int(v) for v in R[f"{tag}/cfg"])
        frames = torch.from_numpy(np.random.RandomState(seed).randint(0, 256, size=(n, hs, ws, 3)).astype(np.uint8)).to(DEV)
        got = _resize_pack(frames, (ht, wt), None, None, 0)
        ref = (torch.from_numpy(R[f"{tag}/out"]) / 255.0 - 0.5) / 0.5
        assert (got.cpu() - ref).abs().max().item() < 1e-5, (tag, (got.cpu() - ref).abs().max().item())
    g = torch.Generator(device=DEV).manual_seed(21)
    frames = torch.randint(0, 256, (5, 200, 200, 3), device=DEV, dtype=torch.uint8, generator=g)
    for size, pad in (((128, 128), 6), ((84, 84), 4)):
        shift, jitter = AugmentSpec(pad=pad, brightness=0.3, contrast=0.3, hue=0.3).draw(5, DEV, g)
        ref = A.train_pipeline(frames.cpu(), shift.cpu(), jitter.cpu(), pad, resize=size)
        got = _resize_pack(frames, size, shift, jitter, pad)
        assert (got.cpu() - ref).abs().max().item() < 5e-5, (size, (got.cpu() - ref).abs().max().item())


def test_color_jitter_kernel_matches_independent_fp64_evaluation():
    J = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jitter.npz"))
    frames = torch.from_numpy(J["frames"]).to(DEV)
    params = torch.from_numpy(J["params"]).to(DEV)
    got = _pack(frames, None, params, 0, torch.float32)
    ref = (torch.from_numpy(J["out"]) - 0.5) / 0.5
    d = (got.cpu().double() - ref).abs()
    assert d.max().item() < 2e-5, d.max().item()


def test_step_with_resizing_augmentation():
    """A TACORL step fed 200x200 uint8 frames with the real-world train pipeline (Resize 84 + shift + jitter): the module
    adopts the resized geometry and the images it encodes are the oracle's."""
    from oracle import augment_oracle as A
    from tacorl_amd.data.augment import AugmentSpec, draw_play_batch_augmentation
    from tests.test_fullsize_gpu import _mod

    B, T = 4, 16
    g = torch.Generator(device=DEV).manual_seed(31)
    st = torch.randint(0, 256, (B, T, 200, 200, 3), device=DEV, dtype=torch.uint8, generator=g)
    gl = torch.randint(0, 256, (B, 200, 200, 3), device=DEV, dtype=torch.uint8, generator=g)
    acts = torch.rand(B, T, 7, device=DEV, generator=g) * 2 - 1
    aug = draw_play_batch_augmentation({"rgb_static": AugmentSpec(pad=4, resize=(84, 84))}, B, T, DEV, g)
    m = _mod("f32")
    m.training_step({"states": {"rgb_static": st}, "goal": {"rgb_static": gl}, "actions": acts,
                     "disp": torch.ones(B, device=DEV).long(), "aug": aug})
    torch.cuda.synchronize()
    assert all(v == v for v in m.logged.values()) and m.frames["rgb_static"].shape == (B * T, 84, 84, 3)
    a = aug["states"]["rgb_static"]
    ref = A.train_pipeline(st.view(B * T, 200, 200, 3).cpu(), a["shift"].view(B * T, 2).cpu(), a["jitter"].view(B * T, 8).cpu(), 4,
                           resize=(84, 84))
    assert (m.frames["rgb_static"].cpu() - ref).abs().max().item() < 5e-5
