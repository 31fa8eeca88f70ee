"""SURVEY.md 8(f) row 1 -- the input side of a step: ToTensor / flip / per-frame ColorJitter / target pyramid.

CPU: the host loaders' restatement of torchvision's ColorJitter against the oracle's (oracle/ref_input.py), and the loader
contract in both modes on a small on-disk KITTI-layout fixture written by the test.  GPU: dd_prepare_frames / dd_pyramid_down2
(through hipops.inputs) against the oracle, Trainer.process_inputs end to end, the double-buffered prefetcher."""
import itertools
import os
import random

import numpy as np
import pytest
import torch

import oracle.ref_input as ori


def random_params(rng, B, F, force_orders=None):
    rows = torch.zeros(B, F, 9)
    orders = list(itertools.permutations(range(4)))
    k = 0
    for b in range(B):
        for f in range(F):
            order = force_orders[k % len(force_orders)] if force_orders else orders[rng.randrange(24)]
            k += 1
            rows[b, f] = torch.tensor([1.0 if rng.random() < 0.8 else 0.0] + [float(o) for o in order] +
                                      [rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2), rng.uniform(-0.1, 0.1)])
    return rows


def smooth_u8(gen, B, F, H, W):
    """photo-like frames: low-frequency colour + noise, with flat patches (grey pixels: the max == min branch of rgb->hsv)."""
    base = torch.rand(B * F, 3, H // 8 + 1, W // 8 + 1, generator=gen)
    img = torch.nn.functional.interpolate(base, (H, W), mode="bilinear", align_corners=False) + 0.1 * torch.rand(B * F, 3, H, W, generator=gen)
    img[:, :, : H // 4, : W // 4] = img[:, :1, : H // 4, : W // 4]
    return (img.clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).reshape(B, F, H, W, 3).contiguous()


def test_host_color_jitter_matches_oracle():
    from datasets.base_dataset import ColorJitter
    rng = random.Random(3)
    gen = torch.Generator().manual_seed(1)
    cj = ColorJitter()
    x = smooth_u8(gen, 1, 1, 32, 48)[0, 0].permute(2, 0, 1).float().div(255)
    for order in itertools.permutations(range(4)):
        vals = [rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2), rng.uniform(0.8, 1.2), rng.uniform(-0.1, 0.1)]
        got = cj.apply(x, (list(order), vals))
        want = ori.color_jitter(x, order, *vals)
        assert float((got - want).abs().max()) < 2e-6, order


def write_kitti_fixture(root, frames=4, size=(96, 64)):
    from PIL import Image
    folder = os.path.join(root, "2011_09_26", "2011_09_26_drive_0001_sync")
    rgb = os.path.join(folder, "image_02", "rgb", "downsample")
    os.makedirs(rgb)
    os.makedirs(os.path.join(folder, "image_02", "depth"))
    gen = torch.Generator().manual_seed(5)
    for i in range(frames):
        arr = smooth_u8(gen, 1, 1, size[1], size[0])[0, 0].numpy()
        Image.fromarray(arr).save(os.path.join(rgb, "{:010}.png".format(i)))
        lidar = np.stack([np.random.RandomState(i).uniform(0, 374, 50), np.random.RandomState(i + 9).uniform(0, 1241, 50),
                          np.random.RandomState(i + 99).uniform(2, 60, 50)], 1)
        np.save(os.path.join(folder, "image_02", "depth", "{:010}.npy".format(i)), lidar)
    with open(os.path.join(folder, "calib_cam_to_cam.txt"), "w") as fh:
        fh.write("S_rect_02: 1.242000e+03 3.750000e+02\nS_rect_03: 1.242000e+03 3.750000e+02\n")
    return "2011_09_26/2011_09_26_drive_0001_sync"


def test_loader_contract_host_and_device_modes(tmp_path):
    """Both modes of the loader describe the same sample: what the device kernels are asked to compute (oracle on the uint8
    frames + drawn parameters) equals what the host path returns for the same random draws."""
    from datasets import KITTIDataset
    folder = write_kitti_fixture(str(tmp_path))
    files = ["{} {} l".format(folder, i) for i in (1, 2)]
    common = dict(data_path=str(tmp_path), filenames=files, height=64, width=96, cam_name="image_02", img_type="downsample",
                  frame_idxs=[0, -1, 1], num_scales=3, is_train=True, img_ext=".png", load_depth=True)
    seen_aug = seen_flip = False
    for seed in range(12):
        items = {}
        for mode in (False, True):
            random.seed(seed)
            items[mode] = KITTIDataset(device_preprocess=mode, **common)[seed % 2]
        host, dev = items[False], items[True]
        assert dev["frames_u8"].shape == (3, 64, 96, 3) and dev["frames_u8"].dtype == torch.uint8
        assert dev["jitter"].shape == (3, 9) and ("color", 0, 0) not in dev
        out = ori.prepare_inputs({f: dev["frames_u8"][i][None] for i, f in enumerate([0, -1, 1])},
                                 {f: dev["jitter"][i][None] for i, f in enumerate([0, -1, 1])}, dev["flip"][None], [0])
        for f in (0, -1, 1):
            assert torch.equal(out[("color", f, 0)][0], host[("color", f, 0)])
            assert float((out[("color_aug", f, 0)][0] - host[("color_aug", f, 0)]).abs().max()) < 2e-6
        assert torch.equal(host["depth_gt"], dev["depth_gt"])          # the LiDAR flip stays on the host in both modes
        seen_aug |= bool(dev["jitter"][:, 0].max() > 0)
        seen_flip |= bool(dev["flip"])
        if dev["jitter"][0, 0] > 0:                                    # per-frame draws, as the reference's tensor ColorJitter behaves
            assert not torch.equal(dev["jitter"][0], dev["jitter"][1])
    assert seen_aug and seen_flip


def test_pyramid_weights_match_torch_on_cpu():
    """The tap set / weights dd_pyramid_down2 uses (8 taps 2i-3..2i+4, Keys a=-0.5 at (t-3.5)/2, clipped taps renormalised),
    restated in numpy, reproduce F.interpolate(bicubic, antialias=True) on the CPU -- borders included."""
    def cubic(x):
        x = abs(x)
        return (1.5 * x - 2.5) * x * x + 1 if x < 1 else (((-0.5 * x + 2.5) * x - 4) * x + 2 if x < 2 else 0.0)

    def down(v):
        n = v.shape[-1]
        out = np.zeros(v.shape[:-1] + (n // 2,))
        for i in range(n // 2):
            taps = [(2 * i - 3 + t, cubic((t - 3.5) * 0.5)) for t in range(8) if 0 <= 2 * i - 3 + t < n]
            tot = sum(w for _, w in taps)
            out[..., i] = sum(v[..., j] * w for j, w in taps) / tot
        return out

    x = torch.rand(1, 1, 12, 20, dtype=torch.float64)
    want = torch.nn.functional.interpolate(x, (6, 10), mode="bicubic", align_corners=False, antialias=True).numpy()
    got = down(down(x.numpy()).swapaxes(-1, -2)).swapaxes(-1, -2)       # horizontal, then vertical
    assert np.abs(got - want).max() < 1e-12


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_prepare_frames_matches_oracle():
    from hipops.inputs import prepare_frames
    B, F, H, W = 8, 3, 64, 96
    gen = torch.Generator().manual_seed(11)
    frames = smooth_u8(gen, B, F, H, W)
    params = random_params(random.Random(2), B, F, force_orders=list(itertools.permutations(range(4))))    # all 24 orders
    flip = torch.tensor([0, 1, 0, 1, 1, 0, 0, 1], dtype=torch.int32)
    color, aug = prepare_frames(frames.cuda(), params.cuda(), flip.cuda())
    want = ori.prepare_inputs({f: frames[:, f] for f in range(F)}, {f: params[:, f] for f in range(F)}, flip, [0])
    for f in range(F):
        assert torch.equal(color[f].cpu(), want[("color", f, 0)])                 # ToTensor + flip: exact
        err = (aug[f].cpu() - want[("color_aug", f, 0)]).abs()
        print("frame %d: color_aug max|err| %.2e" % (f, float(err.max())))
        assert float(err.max()) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(192, 640), (64, 96), (8, 12)])
def test_pyramid_matches_torch(H, W):
    from hipops.inputs import pyramid_down2
    x = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(4))
    got = pyramid_down2(x.cuda()).cpu()
    want = torch.clamp(torch.nn.functional.interpolate(x, (H // 2, W // 2), mode="bicubic", align_corners=False, antialias=True), 0, 1)
    assert float((got - want).abs().max()) < 2e-6
    got2 = pyramid_down2(pyramid_down2(x.cuda())).cpu() if H >= 16 else None
    if got2 is not None:
        want2 = torch.clamp(torch.nn.functional.interpolate(want, (H // 4, W // 4), mode="bicubic", align_corners=False, antialias=True), 0, 1)
        assert float((got2 - want2).abs().max()) < 3e-6


@pytest.mark.gpu
def test_process_inputs_and_prefetcher(tmp_path):
    """Trainer.process_inputs on the loader's uint8 batches == the oracle; the prefetcher yields the same batches in order."""
    from options import DynamoOptions
    from Trainer import Trainer
    from hipops.inputs import DevicePrefetcher
    from torch.utils.data import DataLoader
    folder = write_kitti_fixture(str(tmp_path))
    opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "2", "--height", "64", "--width", "96", "--weights_init", "scratch",
                                      "--num_workers", "0", "--log_dir", str(tmp_path / "logs"), "--data_path", str(tmp_path), "--img_ext", ".png"])
    opt.print_opt = False
    tr = Trainer(opt)
    files = ["{} {} l".format(folder, i) for i in (1, 2, 1, 2)]
    ds = tr.get_dataset(files, is_train=True)
    assert ds.device_preprocess
    random.seed(21)
    raw = list(DataLoader(ds, batch_size=2))
    random.seed(21)
    direct = []
    for b in DataLoader(ds, batch_size=2):
        tr.process_inputs(b)
        direct.append(b)
    for b, r in zip(direct, raw):
        want = ori.prepare_inputs({f: r["frames_u8"][:, i] for i, f in enumerate(opt.frame_ids)},
                                  {f: r["jitter"][:, i] for i, f in enumerate(opt.frame_ids)}, r["flip"], opt.scales)
        for k, v in want.items():
            assert float((b[k].cpu() - v).abs().max()) < 5e-6, k
        assert "frames_u8" not in b
    random.seed(21)
    fetched = list(DevicePrefetcher(DataLoader(ds, batch_size=2, pin_memory=True), tr.process_inputs, tr.device))
    torch.cuda.synchronize()
    assert len(fetched) == len(direct)
    for a, b in zip(fetched, direct):
        for k in b:
            if torch.is_tensor(b[k]):
                assert torch.equal(a[k], b[k]), k
