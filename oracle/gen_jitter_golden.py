"""tests/golden/jitter.npz: torchvision ColorJitter (brightness, contrast, hue in a drawn order) evaluated INDEPENDENTLY of
oracle/augment_oracle.py: per pixel, python floats (fp64), from the documented definitions - brightness / contrast as
blends clamped to [0, 1] (contrast against the mean of the ITU-R 601-2 grey 0.2989 R + 0.587 G + 0.114 B of the image
as it stands), hue as RGB -> HSV (hexcone model), h <- (h + f) mod 1, HSV -> RGB.  torchvision itself is absent from
this image; this fixture pins the restatement and the kernel against a second implementation, including the edge
cases: a constant grey frame, grey pixels, pure primaries (sextant borders), black and white, values that clamp."""
import os

import numpy as np


def rgb_to_hsv(r, g, b):
    mx, mn = max(r, g, b), min(r, g, b)
    v = mx
    if mx == mn:
        return 0.0, 0.0, v
    s = (mx - mn) / mx
    rc, gc, bc = (mx - r) / (mx - mn), (mx - g) / (mx - mn), (mx - b) / (mx - mn)
    if r == mx:
        h = bc - gc
    elif g == mx:
        h = 2.0 + rc - bc
    else:
        h = 4.0 + gc - rc
    return (h / 6.0) % 1.0, s, v


def hsv_to_rgb(h, s, v):
    i = int(np.floor(h * 6.0))
    f = h * 6.0 - i
    c = lambda x: min(max(x, 0.0), 1.0)  # noqa: E731
    p, q, t = c(v * (1.0 - s)), c(v * (1.0 - s * f)), c(v * (1.0 - s * (1.0 - f)))
    return [(v, t, p), (q, v, p), (p, v, t), (p, q, v), (t, p, v), (v, p, q)][i % 6]


def jitter(img, bf, cf, hf, order):
    img = img.astype(np.float64).copy()
    clip = lambda a: np.clip(a, 0.0, 1.0)  # noqa: E731
    for op in order:
        if op == 0:
            img = clip(bf * img)
        elif op == 1:
            mean = (0.2989 * img[..., 0] + 0.587 * img[..., 1] + 0.114 * img[..., 2]).mean()
            img = clip(cf * img + (1.0 - cf) * mean)
        elif op == 3:
            out = np.empty_like(img)
            for y in range(img.shape[0]):
                for x in range(img.shape[1]):
                    h, s, v = rgb_to_hsv(*img[y, x])
                    out[y, x] = hsv_to_rgb((h + hf) % 1.0, s, v)
            img = out
    return img


rs = np.random.RandomState(5)
n, hw = 8, 12
frames = rs.randint(0, 256, size=(n, hw, hw, 3)).astype(np.uint8)
frames[1] = 200                                   # constant grey frame
frames[2, :, :6] = frames[2, :, :6, :1]           # half the pixels grey (r = g = b)
prim = np.array([[255, 0, 0], [255, 255, 0], [0, 255, 0], [0, 255, 255], [0, 0, 255], [255, 0, 255], [0, 0, 0], [255, 255, 255]])
frames[3] = prim[rs.randint(0, 8, size=(hw, hw))]  # primaries / secondaries: exactly on sextant borders; black, white
frames[4, :, :, 1] = frames[4, :, :, 0]            # r == g ties for the max channel
params = np.zeros((n, 8), np.float32)
orders = [[0, 1, 2, 3], [3, 2, 1, 0], [1, 0, 3, 2], [2, 3, 0, 1], [0, 3, 1, 2], [1, 3, 2, 0], [3, 0, 2, 1], [2, 1, 0, 3]]
for i in range(n):
    params[i] = [rs.uniform(0.6, 1.5), rs.uniform(0.6, 1.5), rs.uniform(-0.4, 0.4), *orders[i], 1.0]
params[3, 2] = 1.0 / 6.0   # a hue shift of exactly one sextant
params[5, 0] = 1.9         # brightness that clamps
out = np.stack([jitter(frames[i] / 255.0, float(params[i, 0]), float(params[i, 1]), float(params[i, 2]),
                       [int(v) for v in params[i, 3:7]]) for i in range(n)])
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "jitter.npz")
np.savez_compressed(path, frames=frames, params=params, out=out.astype(np.float64))
print("wrote", path, os.path.getsize(path) / 1e3, "kB")
