"""dd_resize_bicubic (csrc/dd_resize.hip through hipops.resize): the loaders' transforms.Resize(BICUBIC) on PIL frames
(reference datasets/base_dataset.py:80,147) on the device.  CPU: the product's tap tables equal the oracle's, which
tests/test_resize_oracle.py holds bit-identical to Pillow.  GPU: the resized frames equal Pillow's own, bit for bit -- KITTI's
raw sizes down to the training resolution, up-scaling, one unchanged axis (Pillow skips that pass), slots of a larger batch."""
import glob
import os

import numpy as np
import pytest
import torch
from PIL import Image

from oracle.ref_resize import precompute_coeffs, resize_bicubic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sizes", [(1242, 640), (375, 192), (1226, 640), (370, 192), (1600, 512), (900, 288), (96, 192), (53, 31), (7, 40), (4, 1), (640, 640)])
def test_tap_tables_are_the_oracles(sizes):
    from hipops import resize as R
    b, k = R.coefficients(*sizes)
    ob, ok = precompute_coeffs(*sizes)
    assert b.dtype == np.int32 and k.dtype == np.int32
    assert np.array_equal(b, ob) and np.array_equal(k, ok)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(375, 1242, 192, 640), (370, 1226, 192, 640), (376, 1241, 192, 640), (374, 1238, 192, 640), (900, 1600, 288, 512),
                                   (37, 53, 19, 31), (64, 96, 128, 192), (100, 100, 100, 37), (100, 100, 37, 100), (20, 30, 7, 9), (9, 7, 40, 33), (5, 4, 1, 1)])
def test_device_resize_is_pillows(shape):
    from hipops import resize as R
    h, w, oh, ow = shape
    rng = np.random.default_rng(h * 131 + w)
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), np.full((h, w, 3), 255, np.uint8), np.zeros((h, w, 3), np.uint8),
            (rng.integers(0, 2, size=(h, w, 3)) * 255).astype(np.uint8)]          # noise, saturated, black, hard edges (overshoot clipping)
    got = R.resize_batch(torch.from_numpy(np.stack(imgs)).cuda(), oh, ow).cpu().numpy()
    for g, img in zip(got, imgs):
        want = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(g, want), float(np.abs(g.astype(int) - want.astype(int)).max())
        assert np.array_equal(want, resize_bicubic(img, oh, ow))


@pytest.mark.gpu
def test_device_resize_into_slots_of_a_batch(golden_dir):
    """Frames of two source sizes land in their slots of one (n, H, W, 3) batch -- how the loaders' groups of equal geometry are resized."""
    from hipops import resize as R
    rng = np.random.default_rng(7)
    a = [rng.integers(0, 256, size=(375, 1242, 3), dtype=np.uint8) for _ in range(3)]
    b = [rng.integers(0, 256, size=(370, 1226, 3), dtype=np.uint8) for _ in range(2)]
    out = torch.zeros(5, 192, 640, 3, dtype=torch.uint8, device="cuda")
    R.resize_batch(torch.from_numpy(np.stack(a)).cuda(), 192, 640, out=out, slots=[0, 3, 4])
    R.resize_batch(torch.from_numpy(np.stack(b)).cuda(), 192, 640, out=out, slots=[2, 1])
    want = {0: a[0], 3: a[1], 4: a[2], 2: b[0], 1: b[1]}
    for slot, img in want.items():
        assert np.array_equal(out[slot].cpu().numpy(), np.asarray(Image.fromarray(img).resize((640, 192), Image.BICUBIC))), slot
    files = sorted(glob.glob(os.path.join(golden_dir, "tiny_kitti_jpeg", "*.jpg")))
    frames = np.stack([np.asarray(Image.open(f).convert("RGB")) for f in files])
    got = R.resize_batch(torch.from_numpy(frames).cuda(), 96, 320).cpu().numpy()
    for g, f in zip(got, frames):
        assert np.array_equal(g, np.asarray(Image.fromarray(f).resize((320, 96), Image.BICUBIC)))
