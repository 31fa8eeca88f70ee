"""oracle/ref_resize.py (the resize the reference's loaders apply to frames that are not at the training resolution,
datasets/base_dataset.py:80,147: torchvision Resize(BICUBIC) on PIL images = Pillow's Image.resize) against Pillow itself, bit for
bit.  CPU only; the oracle is what a device-side resize will be checked against."""
import glob
import os

import numpy as np
import pytest
from PIL import Image

from oracle.ref_resize import precompute_coeffs, resize_bicubic


@pytest.mark.parametrize("shape", [(375, 1242, 192, 640), (370, 1226, 192, 640), (900, 1600, 288, 512), (37, 53, 19, 31), (64, 96, 128, 192),
                                   (100, 100, 100, 37), (48, 64, 48, 64), (20, 30, 7, 9), (9, 7, 40, 33), (5, 4, 1, 1)])
def test_bicubic_resize_is_pillows(shape):
    h, w, oh, ow = shape
    rng = np.random.default_rng(h * 131 + w)
    for img in (rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), np.full((h, w, 3), 255, np.uint8), np.zeros((h, w, 3), np.uint8),
                (rng.integers(0, 2, size=(h, w, 3)) * 255).astype(np.uint8)):          # noise, saturated, black, hard edges (overshoot clipping)
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(resize_bicubic(img, oh, ow), want)


def test_kitti_frames_through_the_reference_resize(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "tiny_kitti_jpeg", "*.jpg")))
    assert files
    for path in files:
        img = np.asarray(Image.open(path).convert("RGB"))
        for oh, ow in ((96, 320), (192, 640), (375, 1242), (100, 333)):
            want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
            assert np.array_equal(resize_bicubic(img, oh, ow), want)


def test_coefficients_are_normalised_fixed_point():
    bounds, kk = precompute_coeffs(1242, 640)
    assert kk.shape == (640, 2 * int(np.ceil(2.0 * 1242 / 640)) + 1)
    assert np.all(np.abs(kk.sum(1) - (1 << 22)) <= kk.shape[1])           # each row sums to one up to the rounding of its taps
    assert np.all(bounds[:, 0] >= 0) and np.all(bounds[:, 0] + bounds[:, 1] <= 1242)
