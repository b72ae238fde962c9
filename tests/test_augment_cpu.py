"""oracle/augment_oracle.py against the reference's own RandomShiftsAug outputs (tests/golden/augment.npz)."""
import os

import numpy as np
import torch

from oracle import augment_oracle as A

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augment.npz"))


def _frames(n, hw, seed):
    return torch.from_numpy(np.random.RandomState(seed).randint(0, 256, size=(n, hw, hw, 3)).astype(np.uint8))


def test_random_shift_restatement_matches_reference():
    for tag in ("a", "b", "c"):
        n, hw, pad, seed = (int(v) for v in G[f"{tag}/cfg"])
        shift = torch.from_numpy(G[f"{tag}/shift"])
        assert int(shift.min()) >= 0 and int(shift.max()) <= 2 * pad
        got = A.random_shift(_frames(n, hw, seed), shift, pad).float()
        ref = torch.from_numpy(G[f"{tag}/out"])
        # the reference samples its grid bilinearly: ~1e-6 weights on the neighbours (values are 0..255)
        assert (got - ref).abs().max().item() < 2e-2, (tag, (got - ref).abs().max().item())
        assert (got - ref).abs().mean().item() < 1e-3


def test_color_jitter_identities():
    """Sanity of the restated torchvision colour operations: neutral factors are the identity, hue shifts by whole
    turns are the identity, grey images have no hue, brightness scales and clamps."""
    torch.manual_seed(0)
    img = torch.rand(3, 9, 9)
    assert torch.allclose(A.color_jitter(img, 1.0, 1.0, 0.0, [0, 1, 2, 3]), img, atol=1e-6)
    assert torch.allclose(A.adjust_hue(img, 1.0), img, atol=1e-5)
    grey = torch.rand(1, 9, 9).expand(3, 9, 9)
    assert torch.allclose(A.adjust_hue(grey, 0.3), grey, atol=1e-6)
    assert torch.allclose(A.adjust_brightness(img, 2.0), (2 * img).clamp(0, 1))
    a = A.adjust_hue(img, 0.25)
    assert torch.allclose(a.max(0).values, img.max(0).values, atol=1e-6)  # value (max channel) is hue-invariant


def test_resize_restatement_matches_interpolate():
    """torchvision Resize on tensors = F.interpolate(bilinear, align_corners=False): tests/golden/resize.npz holds its
    outputs (oracle/gen_resize_golden.py); the restatement agrees to fp32 rounding of the blend."""
    R = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "resize.npz"))
    for tag in ("static", "gripper", "rect", "up"):
        n, hs, ws, ht, wt, seed = (int(v) for v in R[f"{tag}/cfg"])
        frames = torch.from_numpy(np.random.RandomState(seed).randint(0, 256, size=(n, hs, ws, 3)).astype(np.uint8))
        got = A.resize_bilinear(frames, (ht, wt))
        ref = torch.from_numpy(R[f"{tag}/out"])
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() < 1e-3, (tag, (got - ref).abs().max().item())  # values are 0..255


def test_color_jitter_restatement_matches_independent_fp64_evaluation():
    """tests/golden/jitter.npz: the documented ColorJitter formulas evaluated per pixel in fp64 by a second
    implementation (oracle/gen_jitter_golden.py), grey frames / sextant borders / clamping included."""
    J = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jitter.npz"))
    frames, params, ref = torch.from_numpy(J["frames"]), J["params"], torch.from_numpy(J["out"])
    for i in range(frames.shape[0]):
        img = (frames[i].float() / 255.0).permute(2, 0, 1)
        got = A.color_jitter(img, float(params[i, 0]), float(params[i, 1]), float(params[i, 2]),
                             [int(v) for v in params[i, 3:7]]).permute(1, 2, 0)
        d = (got.double() - ref[i]).abs()
        # fp32 against fp64; a pixel exactly on a sextant border may take the neighbouring branch - both agree there
        assert d.max().item() < 5e-6, (i, d.max().item())
