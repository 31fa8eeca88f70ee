"""Input side of a step at the KITTI bench shape (B=12, 3 frames, 192x640): the device path (upload of the uint8 triplets from
pinned memory + dd_prepare_frames + the two pyramid levels) against the loader-side torch path on this host's CPU
(ToTensor + per-frame ColorJitter per sample, one process), and the fp32 upload the loader-side path needs afterwards."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamo-depth_amd"))
from datasets.base_dataset import ColorJitter, to_tensor  # noqa: E402
from hipops import inputs as I  # noqa: E402
import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

B, F, H, W = 12, 3, 192, 640
g = torch.Generator().manual_seed(0)
u8 = torch.randint(0, 256, (B, F, H, W, 3), dtype=torch.uint8, generator=g).pin_memory()
cj = ColorJitter()
params = torch.stack([torch.stack([ColorJitter.row(cj.draw()) for _ in range(F)]) for _ in range(B)]).pin_memory()
flip = torch.randint(0, 2, (B,), dtype=torch.int32, generator=g).pin_memory()


def device_side():
    d = u8.cuda(non_blocking=True)
    color, aug = I.prepare_frames(d, params.cuda(non_blocking=True), flip.cuda(non_blocking=True))
    l1 = I.pyramid_down2(color[0])
    l2 = I.pyramid_down2(l1)
    return aug, l2


for _ in range(3):
    device_side()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    device_side()
e1.record()
torch.cuda.synchronize()
print("device path: upload %.1f MB uint8 + prepare + pyramid: %.1f us per batch" % (u8.numel() / 1e6, e0.elapsed_time(e1) * 1e3 / n))

d = u8.cuda()
p, f = params.cuda(), flip.cuda()
e0.record()
for _ in range(n):
    color, aug = I.prepare_frames(d, p, f)
    I.pyramid_down2(I.pyramid_down2(color[0]))
e1.record()
torch.cuda.synchronize()
print("   kernels only (frames resident): %.1f us per batch" % (e0.elapsed_time(e1) * 1e3 / n))

# loader-side path on the host (what a DataLoader worker does per sample, reference datasets/base_dataset.py:83-95)
torch.set_num_threads(1)
imgs = [Image.fromarray(u8[0, k].numpy()) for k in range(F)]
t = time.perf_counter()
reps = 4
for _ in range(reps):
    for k in range(F):
        x = to_tensor(imgs[k])
        cj.apply(x, cj.draw())
host = (time.perf_counter() - t) / reps
print("host path: ToTensor + ColorJitter of one triplet on one core: %.1f ms -> %.1f ms per batch of %d per worker" % (host * 1e3, host * 1e3 * B, B))
f32 = torch.rand(2 * F * B, 3, H, W).pin_memory()          # color + color_aug, fp32
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    f32.cuda(non_blocking=True)
e1.record()
torch.cuda.synchronize()
print("   fp32 upload of color + color_aug (%.1f MB): %.1f us per batch" % (f32.numel() * 4 / 1e6, e0.elapsed_time(e1) * 1e3 / 20))
