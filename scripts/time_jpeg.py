"""Device JPEG decode of one training batch (12 triplets = 36 frames of 640x192, 4:2:0): time per batch from HIP events, next to
PIL's decode of the same files on one host core.  GPU box:  python scripts/time_jpeg.py"""
import io
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
from hipops import jpeg  # noqa: E402

files = sorted(os.path.join(ROOT, "tests", "golden", "tiny_kitti_jpeg", f) for f in os.listdir(os.path.join(ROOT, "tests", "golden", "tiny_kitti_jpeg")))
datas = [open(f, "rb").read() for f in files] * 6
recs, geoms = zip(*[jpeg.parse_header(d) for d in datas])
w, h, nc, hs, vs = geoms[0]
cap = (max(len(d) for d in datas) + 4095) // 4096 * 4096
buf = np.zeros((len(datas), cap), np.uint8)
for i, d in enumerate(datas):
    buf[i, :len(d)] = np.frombuffer(d, np.uint8)
dev_b, dev_h = torch.from_numpy(buf).cuda(), torch.from_numpy(np.stack(recs)).cuda()
for _ in range(3):
    out = jpeg.decode_batch(dev_b, dev_h, h, w, nc, hs, vs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    out = jpeg.decode_batch(dev_b, dev_h, h, w, nc, hs, vs)
e1.record()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ref = [np.asarray(Image.open(io.BytesIO(d)).convert("RGB")) for d in datas]
t_pil = (time.perf_counter() - t0) / 3
t0 = time.perf_counter()
for _ in range(3):
    [jpeg.parse_header(d) for d in datas]
t_parse = (time.perf_counter() - t0) / 3
ok = all(np.array_equal(o, r) for o, r in zip(out.cpu().numpy(), ref))
print("%d frames %dx%d (%.1f KB compressed each): device decode %.2f ms per batch (Huffman + IDCT + up-sampling + colour); PIL on one core %.1f ms; "
      "header parsing on the host %.2f ms; identical to PIL: %s" % (len(datas), w, h, np.mean([len(d) for d in datas]) / 1024, e0.elapsed_time(e1) / 20, t_pil * 1e3, t_parse * 1e3, ok))
