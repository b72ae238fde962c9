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
