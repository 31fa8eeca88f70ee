"""CPU restatement of the image resize the reference's loaders apply to frames that are not stored at the training resolution.
TEST INFRASTRUCTURE: only tests/ may import it (it pins the behaviour a device-side resize will have to reproduce; round 3 ships
no such kernel yet -- DESIGN.md section 9.4).

Reference: datasets/base_dataset.py:80 `transforms.Resize((height, width), interpolation=BICUBIC)` applied to PIL images at :147 --
torchvision hands a PIL image to `Image.resize((w, h), Image.BICUBIC)`, i.e. Pillow's two-pass separable convolution
(src/libImaging/Resample.c, public source, restated here from its documented algorithm): per output index a window of
`support * max(scale, 1)` input pixels around the centre `(x + 0.5) * scale`, weights from the Keys cubic (a = -0.5) evaluated at
the distances divided by max(scale, 1) (the antialiasing), normalised to sum 1, quantised to 22-bit fixed point; a horizontal
pass then a vertical pass, each rounding to uint8 with saturation.  PINNED: tests/test_resize_oracle.py compares it bit for bit
with Pillow itself (the library the reference calls) on random images and on the tiny_kitti frames, down- and up-scaling."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size, support=2.0, filt=_bicubic):
    """-> (bounds (out,2) [first input index, count], integer coefficients (out, ksize)) -- Resample.c precompute_coeffs +
    normalize_coeffs_8bpc for the full-image box."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - sup + 0.5), 0)
        xmax = min(int(center + sup + 0.5), in_size) - xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One separable pass along `axis` of an (H, W, C) uint8 array: fixed-point sums, + half, >> 22, saturate."""
    out_size, ksize = kk.shape
    idx = np.minimum(bounds[:, :1] + np.arange(ksize)[None, :], img.shape[axis] - 1)          # taps beyond the count have weight 0
    src = img.astype(np.int64)
    if axis == 1:
        acc = (src[:, idx, :] * kk[None, :, :, None]).sum(2)
    else:
        acc = (src[idx, :, :] * kk[:, :, None, None]).sum(1)
    acc = (acc + (1 << (PRECISION_BITS - 1))) >> PRECISION_BITS
    return np.clip(acc, 0, 255).astype(np.uint8)


def resize_bicubic(img, out_h, out_w):
    """(H, W, C) uint8 -> (out_h, out_w, C) uint8, what `PIL.Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC)` holds."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w = img.shape[:2]
    if out_w != w:
        img = _pass(img, *precompute_coeffs(w, out_w), axis=1)
    if out_h != h:
        img = _pass(img, *precompute_coeffs(h, out_h), axis=0)
    return img
