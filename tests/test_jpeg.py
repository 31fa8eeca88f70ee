"""Baseline JPEG decoding (first stage): the numpy restatement (oracle/ref_jpeg.py) is PINNED against PIL's
own decode -- what the reference's loader executes (datasets/base_dataset.py:13-18) -- bit for bit, on the six tiny_kitti frames
and on JPEGs encoded here with every chroma sub-sampling, several qualities, odd sizes and restart markers; the device decoder
(csrc/dd_jpeg.hip through hipops.jpeg) is then checked against PIL on the same files, bit for bit, on the GPU."""
import glob
import io
import os

import numpy as np
import pytest
import torch
from PIL import Image

import oracle.ref_jpeg as rj

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KITTI = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "tiny_kitti_jpeg", "*.jpg")))


def pil_decode(data):
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))


def synthetic(w, h, subsampling, quality, restart=0, seed=0, grey=False):
    rng = np.random.default_rng(seed)
    base = rng.random((h // 8 + 2, w // 8 + 2, 3))
    img = np.kron(base, np.ones((8, 8, 1)))[:h, :w] * 200 + rng.random((h, w, 3)) * 55
    im = Image.fromarray(img.astype(np.uint8))
    if grey:
        im = im.convert("L")
    buf = io.BytesIO()
    kw = dict(quality=quality)
    if not grey:
        kw["subsampling"] = subsampling
    if restart:
        kw["restart_marker_blocks"] = restart
    im.save(buf, "JPEG", **kw)
    return buf.getvalue()


CASES = [(96, 64, 2, 75, 0), (96, 64, 0, 90, 0), (96, 64, 1, 60, 0), (101, 67, 2, 85, 0), (101, 67, 1, 85, 0), (99, 35, 0, 95, 0),
         (96, 64, 2, 75, 4), (160, 48, 2, 30, 1), (640, 192, 2, 92, 0), (33, 17, 2, 50, 0)]


def test_kitti_fixtures_are_the_reference_frames(golden_dir):
    """The six committed JPEGs decode (with PIL) to the frames the process_batch golden of the unmodified reference was run on."""
    z = np.load(os.path.join(golden_dir, "net_tiny_kitti.npz"))
    assert len(KITTI) == 6
    frames = {f: z["in/color|{}".format(f)] for f in (0, -1, 1)}            # (B,3,H,W) uint8, B = [image_02, image_03]
    for cam, b in (("image_02", 0), ("image_03", 1)):
        for f, idx in ((-1, 0), (0, 1), (1, 2)):
            data = open(os.path.join(ROOT, "tests", "golden", "tiny_kitti_jpeg", "{}_{}.jpg".format(cam, idx)), "rb").read()
            assert np.array_equal(pil_decode(data).transpose(2, 0, 1), frames[f][b]), (cam, f)


@pytest.mark.parametrize("path", KITTI, ids=[os.path.basename(p) for p in KITTI])
def test_oracle_equals_pil_on_tiny_kitti(path):
    data = open(path, "rb").read()
    assert np.array_equal(rj.decode(data), pil_decode(data))


@pytest.mark.parametrize("w,h,ss,q,rst", CASES)
def test_oracle_equals_pil_on_encoded_images(w, h, ss, q, rst):
    data = synthetic(w, h, ss, q, rst)
    assert (rj.JpegHeader(data).restart > 0) == (rst > 0)
    assert np.array_equal(rj.decode(data), pil_decode(data))


def test_oracle_greyscale():
    data = synthetic(72, 40, 0, 80, grey=True)
    assert np.array_equal(rj.decode(data), pil_decode(data))


def test_header_record_matches_the_oracles_parse():
    from hipops import abi, jpeg
    for data in [open(KITTI[0], "rb").read(), synthetic(101, 67, 1, 85), synthetic(96, 64, 2, 75, 4)]:
        rec, geom = jpeg.parse_header(data)
        want = rj.JpegHeader(data)
        hd = abi.DDJpegHeader.from_buffer_copy(rec.tobytes())
        assert (hd.width, hd.height, hd.data_offset, hd.data_end, hd.restart_interval) == (want.width, want.height, want.data_offset, len(data), want.restart)
        assert geom[:3] == (want.width, want.height, len(want.components))
        for k, (_, h, v, tq) in enumerate(want.components):
            assert (hd.h[k], hd.v[k], hd.tq[k]) == (h, v, tq)
            assert np.array_equal(np.array(hd.qt[tq]), want.qt[tq])
        for (tc, th), (bits, vals) in want.huff.items():
            t = 2 * tc + th
            assert list(hd.bits[t]) == bits and list(hd.vals[t])[:len(vals)] == vals


def test_progressive_and_cmyk_are_refused():
    from hipops import jpeg
    im = Image.fromarray((np.random.default_rng(0).random((32, 48, 3)) * 255).astype(np.uint8))
    buf = io.BytesIO()
    im.save(buf, "JPEG", progressive=True)
    with pytest.raises(jpeg.UnsupportedJpeg):
        jpeg.parse_header(buf.getvalue())
    buf = io.BytesIO()
    im.convert("CMYK").save(buf, "JPEG")
    with pytest.raises(jpeg.UnsupportedJpeg):
        jpeg.parse_header(buf.getvalue())
    with pytest.raises(jpeg.UnsupportedJpeg):
        jpeg.parse_header(b"\x89PNG\r\n\x1a\n" + b"\0" * 32)


# ---- the device decoder ------------------------------------------------------------------------------------------------------
def device_decode(datas):
    from hipops import jpeg
    recs, geoms = zip(*[jpeg.parse_header(d) for d in datas])
    assert len(set(geoms)) == 1
    w, h, nc, hs, vs = geoms[0]
    cap = (max(len(d) for d in datas) + 4095) // 4096 * 4096
    buf = np.zeros((len(datas), cap), np.uint8)
    for i, d in enumerate(datas):
        buf[i, :len(d)] = np.frombuffer(d, np.uint8)
    out = jpeg.decode_batch(torch.from_numpy(buf).cuda(), torch.from_numpy(np.stack(recs)).cuda(), h, w, nc, hs, vs)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.gpu
def test_device_decoder_equals_pil_on_tiny_kitti():
    datas = [open(p, "rb").read() for p in KITTI]
    got = device_decode(datas)
    for g, d in zip(got, datas):
        assert np.array_equal(g, pil_decode(d))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,ss,q,rst", CASES)
def test_device_decoder_equals_pil_on_encoded_images(w, h, ss, q, rst):
    datas = [synthetic(w, h, ss, q, rst, seed=s) for s in range(3)]
    got = device_decode(datas)
    for g, d in zip(got, datas):
        want = pil_decode(d)
        assert g.shape == want.shape
        assert np.array_equal(g, want), (int(np.abs(g.astype(int) - want.astype(int)).max()), float((g != want).mean()))


@pytest.mark.gpu
def test_device_decoder_greyscale_and_large_batch():
    datas = [synthetic(72, 40, 0, 80, seed=s, grey=True) for s in range(2)]
    for g, d in zip(device_decode(datas), datas):
        assert np.array_equal(g, pil_decode(d))
    datas = [synthetic(640, 192, 2, 90, seed=s) for s in range(36)]          # one training batch: 12 triplets
    for g, d in zip(device_decode(datas), datas):
        assert np.array_equal(g, pil_decode(d))


def write_kitti_jpeg_fixture(root):
    """The six tiny_kitti frames in the reference's processed layout (<folder>/image_0{2,3}/rgb/downsample/<frame:010>.jpg)."""
    import shutil
    folder = os.path.join(root, "2011_09_26", "2011_09_26_drive_0001_sync")
    for cam in ("image_02", "image_03"):
        rgb = os.path.join(folder, cam, "rgb", "downsample")
        os.makedirs(rgb)
        for i in range(3):
            shutil.copy(os.path.join(ROOT, "tests", "golden", "tiny_kitti_jpeg", "{}_{}.jpg".format(cam, i)), os.path.join(rgb, "{:010}.jpg".format(i)))
    with open(os.path.join(folder, "calib_cam_to_cam.txt"), "w") as fh:
        fh.write("S_rect_02: 1.242000e+03 3.750000e+02\nS_rect_03: 1.242000e+03 3.750000e+02\n")
    return "2011_09_26/2011_09_26_drive_0001_sync"


def test_loader_hands_over_compressed_frames(tmp_path):
    """device_decode: the item carries the file bytes + header records instead of decoded pixels; a dataset whose files are not at
    the training resolution (or are no baseline JPEGs) keeps the PIL path."""
    from datasets import KITTIDataset
    from hipops import jpeg
    folder = write_kitti_jpeg_fixture(str(tmp_path))
    files = ["{} 1 l".format(folder), "{} 1 r".format(folder)]
    kw = dict(data_path=str(tmp_path), filenames=files, cam_name="image_02", img_type="downsample", frame_idxs=[0, -1, 1], num_scales=3,
              is_train=True, img_ext=".jpg", device_preprocess=True, device_decode=True)
    ds = KITTIDataset(height=192, width=640, **kw)
    assert ds.device_decode
    item = ds[0]
    assert "frames_u8" not in item and item["jpeg_bytes"].shape == (3, ds._jpeg_cap) and item["jpeg_hdr"].shape == (3, jpeg.HEADER_BYTES)
    for k, f in enumerate([0, -1, 1]):
        data = open(os.path.join(str(tmp_path), folder, "image_02", "rgb", "downsample", "{:010}.jpg".format(1 + f)), "rb").read()
        assert bytes(item["jpeg_bytes"][k, :len(data)].numpy()) == data
    other = KITTIDataset(height=96, width=320, device_resize=False, **kw)      # needs a resize and the device resize is off: stays on the host
    assert not other.device_decode and "frames_u8" in other[0] and other[0]["frames_u8"].shape == (3, 96, 320, 3)
    third = KITTIDataset(height=96, width=320, **kw)              # device resize (default): the compressed frames travel, whatever their size
    assert third.device_decode and "jpeg_bytes" in third[0] and "frames_u8" not in third[0]


def test_one_unusual_file_does_not_end_the_run(tmp_path):
    """a progressive file (or one with another sampling, or more bytes than the fixed record) among baseline frames: that
    sample travels as PIL-decoded pixels, `collate` turns the rest of its batch into pixels as well (from the bytes they carry --
    the pixels the device decoder would have produced), and batches without such a sample stay compressed."""
    from PIL import Image
    from datasets import KITTIDataset
    from torch.utils.data import DataLoader
    folder = write_kitti_jpeg_fixture(str(tmp_path))
    odd = os.path.join(str(tmp_path), folder, "image_03", "rgb", "downsample", "{:010}.jpg".format(2))
    with Image.open(odd) as img:
        img.convert("RGB").save(odd, "JPEG", quality=90, progressive=True)
    files = ["{} 1 l".format(folder), "{} 1 r".format(folder)]
    ds = KITTIDataset(data_path=str(tmp_path), filenames=files, cam_name="image_02", img_type="downsample", frame_idxs=[0, -1, 1], num_scales=3,
                      is_train=False, img_ext=".jpg", device_preprocess=True, device_decode=True, height=192, width=640)
    assert ds.device_decode
    assert "jpeg_bytes" in ds[0] and "frames_u8" in ds[1] and "jpeg_bytes" not in ds[1]
    (mixed,) = list(DataLoader(ds, batch_size=2, collate_fn=ds.collate))
    assert "jpeg_bytes" not in mixed and mixed["frames_u8"].shape == (2, 3, 192, 640, 3)
    for i, cam in enumerate(("image_02", "image_03")):
        for k, f in enumerate((0, -1, 1)):
            data = open(os.path.join(str(tmp_path), folder, cam, "rgb", "downsample", "{:010}.jpg".format(1 + f)), "rb").read()
            assert np.array_equal(mixed["frames_u8"][i, k].numpy(), pil_decode(data)), (cam, f)
    first, second = list(DataLoader(ds, batch_size=1, collate_fn=ds.collate))
    assert "jpeg_bytes" in first and "frames_u8" not in first and "frames_u8" in second


def test_one_unusual_file_of_another_size_does_not_end_the_run(tmp_path):
    """the mixed batch with the device resize ON and files that are NOT at the training resolution (KITTI's
    `original` image type; here the 640x192 fixture frames for a 320x96 run).  The host-decoded sample arrives at (H,W) through
    _host_frame; the samples `collate` decodes from their bytes must be resized the same way (PIL bicubic, the reference's loader
    arithmetic, datasets/base_dataset.py:140-147), or default_collate fails on mismatched shapes and the run aborts."""
    from PIL import Image
    from datasets import KITTIDataset
    from torch.utils.data import DataLoader
    folder = write_kitti_jpeg_fixture(str(tmp_path))
    odd = os.path.join(str(tmp_path), folder, "image_03", "rgb", "downsample", "{:010}.jpg".format(2))
    with Image.open(odd) as img:
        img.convert("RGB").save(odd, "JPEG", quality=90, progressive=True)
    files = ["{} 1 l".format(folder), "{} 1 r".format(folder)]
    ds = KITTIDataset(data_path=str(tmp_path), filenames=files, cam_name="image_02", img_type="downsample", frame_idxs=[0, -1, 1], num_scales=3,
                      is_train=False, img_ext=".jpg", device_preprocess=True, device_decode=True, device_resize=True, height=96, width=320)
    assert ds.device_decode and ds.device_resize
    assert "jpeg_bytes" in ds[0] and "frames_u8" in ds[1] and ds[1]["frames_u8"].shape == (3, 96, 320, 3)
    (mixed,) = list(DataLoader(ds, batch_size=2, collate_fn=ds.collate))
    assert "jpeg_bytes" not in mixed and mixed["frames_u8"].shape == (2, 3, 96, 320, 3)
    for i, cam in enumerate(("image_02", "image_03")):
        for k, f in enumerate((0, -1, 1)):
            path = os.path.join(str(tmp_path), folder, cam, "rgb", "downsample", "{:010}.jpg".format(1 + f))
            with Image.open(path) as img:
                want = np.asarray(img.convert("RGB").resize((320, 96), Image.BICUBIC))
            assert np.array_equal(mixed["frames_u8"][i, k].numpy(), want), (cam, f)


@pytest.mark.gpu
def test_training_inputs_from_compressed_frames(tmp_path):
    """Trainer.process_inputs on compressed batches == the PIL-decoded frames / 255 (flip applied), through DataLoader + prefetcher."""
    import random
    from options import DynamoOptions
    from Trainer import Trainer
    from hipops.inputs import DevicePrefetcher
    from torch.utils.data import DataLoader
    folder = write_kitti_jpeg_fixture(str(tmp_path))
    opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "2", "--weights_init", "scratch", "--num_workers", "0",
                                      "--log_dir", str(tmp_path / "logs"), "--data_path", str(tmp_path), "--no_hip_graph"])
    opt.print_opt = False
    tr = Trainer(opt)
    files = ["{} 1 l".format(folder), "{} 1 r".format(folder)]
    ds = tr.get_dataset(files, is_train=True)
    assert ds.device_preprocess and ds.device_decode
    random.seed(3)
    raw = list(DataLoader(ds, batch_size=2))
    random.seed(3)
    batches = list(DevicePrefetcher(DataLoader(ds, batch_size=2, pin_memory=True), tr.process_inputs, tr.device))
    torch.cuda.synchronize()
    assert len(batches) == 1 and "jpeg_bytes" not in batches[0] and "frames_u8" not in batches[0]
    b, r = batches[0], raw[0]
    for i, cam in enumerate(("image_02", "image_03")):
        for f in (0, -1, 1):
            data = open(os.path.join(str(tmp_path), folder, cam, "rgb", "downsample", "{:010}.jpg".format(1 + f)), "rb").read()
            want = torch.from_numpy(pil_decode(data).copy()).permute(2, 0, 1).float().div(255)
            if int(r["flip"][i]):
                want = want.flip(-1)
            assert torch.equal(b[("color", f, 0)][i].cpu(), want), (cam, f)
    assert b[("color", 0, 1)].shape == (2, 3, 96, 320)


@pytest.mark.gpu
def test_training_inputs_from_compressed_frames_of_another_size(tmp_path):
    """SURVEY 8(f) row 1, last residue: files that are NOT at the training resolution (here the 640x192 fixture frames for a 320x96
    run; KITTI's originals for a 640x192 run) decode AND resize on the device: Trainer.process_inputs == ToTensor(PIL decode ->
    PIL resize(BICUBIC)), the reference's loader arithmetic (datasets/base_dataset.py:80,140-147), bit for bit."""
    from PIL import Image
    from options import DynamoOptions
    from Trainer import Trainer
    from torch.utils.data import DataLoader
    folder = write_kitti_jpeg_fixture(str(tmp_path))
    opt = DynamoOptions().parse(args=["-d", "kitti", "--depth_model", "litemono", "-b", "2", "--weights_init", "scratch", "--num_workers", "0", "--height", "96",
                                      "--width", "320", "--log_dir", str(tmp_path / "logs"), "--data_path", str(tmp_path), "--no_hip_graph"])
    opt.print_opt = False
    tr = Trainer(opt)
    files = ["{} 1 l".format(folder), "{} 1 r".format(folder)]
    ds = tr.get_dataset(files, is_train=False)
    assert ds.device_preprocess and ds.device_decode and ds.device_resize
    (batch,) = list(DataLoader(ds, batch_size=2, collate_fn=ds.collate))
    assert "jpeg_bytes" in batch
    tr.process_inputs(batch)
    torch.cuda.synchronize()
    for i, cam in enumerate(("image_02", "image_03")):
        for f in (0, -1, 1):
            path = os.path.join(str(tmp_path), folder, cam, "rgb", "downsample", "{:010}.jpg".format(1 + f))
            with Image.open(path) as img:
                want = np.asarray(img.convert("RGB").resize((320, 96), Image.BICUBIC))
            want = torch.from_numpy(want.copy()).permute(2, 0, 1).float().div(255)
            assert torch.equal(batch[("color", f, 0)][i].cpu(), want), (cam, f)
    assert batch[("color", 0, 1)].shape == (2, 3, 48, 160)
