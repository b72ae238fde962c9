"""CPU restatement of the reference's train-time image pipeline.  TEST INFRASTRUCTURE ONLY.

config/datamodule/transform_manager/transforms/rl_train.yaml per camera: torchvision Resize -> RandomShiftsAug(pad) ->
ScaleImageTensor (x / 255) -> ColorTransform = torchvision ColorJitter(contrast, brightness, hue) -> Normalize(0.5, 0.5).

* resize_bilinear: torchvision's tensor Resize is torch.nn.functional.interpolate; PINNED against F.interpolate's own
  outputs (tests/golden/resize.npz, oracle/gen_resize_golden.py).

* random_shift: reference utils/transforms.py:265-299.  The reference builds a sampling grid whose points are, for an
  integer draw `shift = randint(0, 2 * pad + 1)`, exactly the pixel centres of the replicate-padded frame translated by
  the draw, and samples it bilinearly - i.e. an integer translation with clamped borders, up to the ~1e-6
  interpolation weights that fp32 grid arithmetic leaves on the neighbours.  PINNED: tests/golden/augment.npz holds the
  outputs of the reference class itself (oracle/gen_augment_golden.py) and tests/test_augment_cpu.py checks this
  restatement against them.
* color_jitter: torchvision is a third-party dependency that is ABSENT from this image (setup.cfg pins no version; the
  reference imports `torchvision.transforms.ColorJitter`).  The functions below restate the published algorithm of
  torchvision.transforms.v1 ColorJitter.forward and _functional_tensor.py (adjust_brightness / adjust_contrast /
  adjust_hue with _rgb2hsv / _hsv2rgb, float images in [0, 1]).  There is no torchvision output to compare with here
  (PARITY UNPINNED against torchvision itself); tests/golden/jitter.npz pins it against an INDEPENDENT evaluation of the
  documented formulas - per-pixel python / colorsys-style arithmetic in fp64, oracle/gen_jitter_golden.py - including the
  edge cases: grey pixels and frames (hue undefined, saturation 0), pixels on sextant borders, clamping at 0 and 1.
Every random draw is an explicit argument.
"""
import torch


def resize_bilinear(frames, size):
    """torchvision.transforms.Resize(size) on float tensors (rl_train.yaml:3-4,16-17; the reference's torchvision
    generation: bilinear, no antialias) = torch.nn.functional.interpolate(mode="bilinear", align_corners=False), i.e.
    ATen's upsample_bilinear2d: src = max(scale (dst + 0.5) - 0.5, 0) with scale = in / out in fp32, the two
    neighbours i0 = floor(src), i1 = i0 + (i0 < in - 1), weights l1 = src - i0, l0 = 1 - l1, and the blend
    h0 (w0 p00 + w1 p01) + h1 (w0 p10 + w1 p11).  PINNED: tests/golden/resize.npz holds F.interpolate's own outputs
    (oracle/gen_resize_golden.py).  frames (n,H,W,C) uint8 / float -> (n,size[0],size[1],C) fp32 (0..255 scale kept)."""
    x = frames.to(torch.float32)
    n, H, W, C = x.shape
    oh, ow = size

    def axis(out, inn):
        scale = torch.tensor(inn / out, dtype=torch.float32)
        src = torch.clamp(scale * (torch.arange(out, dtype=torch.float32) + 0.5) - 0.5, min=0.0)
        i0 = src.floor().to(torch.int64).clamp(max=inn - 1)
        i1 = i0 + (i0 < inn - 1).to(torch.int64)
        l1 = src - i0.to(torch.float32)
        return i0, i1, 1.0 - l1, l1

    y0, y1, h0, h1 = axis(oh, H)
    x0, x1, w0, w1 = axis(ow, W)
    w0, w1 = w0.view(1, 1, ow, 1), w1.view(1, 1, ow, 1)
    h0, h1 = h0.view(1, oh, 1, 1), h1.view(1, oh, 1, 1)
    top = w0 * x[:, y0][:, :, x0] + w1 * x[:, y0][:, :, x1]
    bot = w0 * x[:, y1][:, :, x0] + w1 * x[:, y1][:, :, x1]
    return h0 * top + h1 * bot


def random_shift(frames, shift, pad):
    """frames (n,H,W,C) any dtype; shift (n,2) integer draws (sx, sy) in [0, 2*pad] -> translated frames, same dtype."""
    n, H, W, _ = frames.shape
    out = torch.empty_like(frames)
    ys, xs = torch.arange(H), torch.arange(W)
    for i in range(n):
        sx, sy = int(shift[i][0]) - pad, int(shift[i][1]) - pad
        yy = (ys + sy).clamp(0, H - 1)
        xx = (xs + sx).clamp(0, W - 1)
        out[i] = frames[i][yy][:, xx]
    return out


def _gray(img):
    r, g, b = img.unbind(-3)
    return (0.2989 * r + 0.587 * g + 0.114 * b).unsqueeze(-3)


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1.0)


def adjust_brightness(img, f):
    return _blend(img, torch.zeros_like(img), f)


def adjust_contrast(img, f):
    mean = torch.mean(_gray(img), dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, f)


def _rgb2hsv(img):
    r, g, b = img.unbind(-3)
    maxc, minc = torch.max(img, dim=-3).values, torch.min(img, dim=-3).values
    eqc = maxc == minc
    cr = maxc - minc
    ones = torch.ones_like(maxc)
    s = cr / torch.where(eqc, ones, maxc)
    div = torch.where(eqc, ones, cr)
    rc, gc, bc = (maxc - r) / div, (maxc - g) / div, (maxc - b) / div
    hr = (maxc == r) * (bc - gc)
    hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
    hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
    h = torch.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0)
    return torch.stack((h, s, maxc), dim=-3)


def _hsv2rgb(img):
    h, s, v = img.unbind(-3)
    i = torch.floor(h * 6.0)
    f = h * 6.0 - i
    i = i.to(torch.int32) % 6
    p = torch.clamp(v * (1.0 - s), 0.0, 1.0)
    q = torch.clamp(v * (1.0 - f * s), 0.0, 1.0)
    t = torch.clamp(v * (1.0 - (1.0 - f) * s), 0.0, 1.0)
    mask = i.unsqueeze(-3) == torch.arange(6).view(-1, 1, 1)
    a1 = torch.stack((v, q, p, p, t, v), dim=-3)
    a2 = torch.stack((t, v, v, q, p, p), dim=-3)
    a3 = torch.stack((p, p, t, v, v, q), dim=-3)
    a4 = torch.stack((a1, a2, a3), dim=-4)
    return torch.einsum("...ijk, ...xijk -> ...xjk", mask.to(img.dtype), a4)


def adjust_hue(img, hf):
    hsv = _rgb2hsv(img)
    h, s, v = hsv.unbind(-3)
    h = (h + hf) % 1.0
    return _hsv2rgb(torch.stack((h, s, v), dim=-3))


def color_jitter(img, brightness_factor, contrast_factor, hue_factor, order):
    """ColorJitter.forward on one CHW float image in [0,1]: the drawn factors applied in the drawn order
    (fn_idx: 0 brightness, 1 contrast, 2 saturation - None in the reference's configs -, 3 hue)."""
    for fn in order:
        if fn == 0:
            img = adjust_brightness(img, brightness_factor)
        elif fn == 1:
            img = adjust_contrast(img, contrast_factor)
        elif fn == 3:
            img = adjust_hue(img, hue_factor)
    return img


def train_pipeline(frames_u8, shift=None, jitter=None, pad=0, resize=None):
    """uint8 HWC frames (n,H,W,3) -> normalised fp32 NHWC, as the dataloader's transform chain produces them
    (channels-last here; the reference's tensors are CHW).  jitter (n,8): {b, c, h, order0..3, apply};
    resize = (H, W): torchvision Resize first, on the 0..255 float frame."""
    x = frames_u8
    if resize is not None:
        x = resize_bilinear(x, resize)
    if shift is not None:
        x = random_shift(x, shift, pad)
    x = x.to(torch.float32) / 255.0
    if jitter is not None:
        out = []
        for i in range(x.shape[0]):
            img = x[i].permute(2, 0, 1)
            if float(jitter[i][7]) != 0.0:
                img = color_jitter(img, float(jitter[i][0]), float(jitter[i][1]), float(jitter[i][2]),
                                   [int(v) for v in jitter[i][3:7]])
            out.append(img.permute(1, 2, 0))
        x = torch.stack(out)
    return (x - 0.5) / 0.5
