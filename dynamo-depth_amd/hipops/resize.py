"""Host side of the device bicubic resize (csrc/dd_resize.hip): Pillow's tap tables and the batch call.

Replaces `transforms.Resize((height, width), interpolation=BICUBIC)` on PIL frames in the loaders (reference
datasets/base_dataset.py:80,147) for frames that are not stored at the training resolution (KITTI's `original` image type: four
sizes around 1242x375): the workers hand over the compressed files, the GPU decodes (hipops.jpeg) and resizes -- the result is
Pillow's, bit for bit.

The tables follow Pillow's src/libImaging/Resample.c (public source; `precompute_coeffs` + `normalize_coeffs_8bpc`), restated:
output index xx looks at the input window [center - support, center + support) around center = (xx + 0.5) * scale with
support = 2 * max(scale, 1) (bicubic, stretched when down-scaling: the antialiasing), weights = Keys cubic (a = -0.5) of the
distances / max(scale, 1), normalised to sum 1, rounded half away from zero to 22-bit fixed point."""
import ctypes as C
import math

import numpy as np
import torch

from . import abi
from . import lib as L

PRECISION_BITS = 32 - 8 - 2
_tables = {}


def _cubic(x):
    x = np.abs(x)
    a = -0.5
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    far = (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))


def coefficients(in_size, out_size):
    """(bounds (out, 2) int32 [first input index, tap count], coef (out, ksize) int32) of one axis."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    inv = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        count = min(int(center + support + 0.5), in_size) - xmin
        w = _cubic((np.arange(count, dtype=np.float64) + xmin - center + 0.5) * inv)
        total = 0.0
        for v in w:                      # (left-to-right sum in double, as the C loop accumulates it)
            total += float(v)
        if total != 0.0:
            w = w / total
        q = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))      # C casts truncate
        coef[xx, :count] = q
        bounds[xx] = (xmin, count)
    return bounds, coef


def _device_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    t = _tables.get(key)
    if t is None:
        b, k = coefficients(in_size, out_size)
        t = _tables[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), k.shape[1])
    return t


def resize_batch(src, out_h, out_w, out=None, slots=None):
    """src (n, Hs, Ws, 3) uint8 on the GPU -> (n, out_h, out_w, 3) uint8 (or image i into out[slots[i]] of a caller's (m, out_h, out_w, 3)
    buffer): what `Image.fromarray(frame).resize((out_w, out_h), Image.BICUBIC)` holds for every frame."""
    if not src.is_cuda or src.dtype != torch.uint8 or src.dim() != 4 or src.shape[-1] != 3:
        raise L.DynamoHipError("resize_batch takes (n, H, W, 3) uint8 frames on the GPU")
    lib = L.load()
    src = src.contiguous()
    n, hs, ws = int(src.shape[0]), int(src.shape[1]), int(src.shape[2])
    if out is None:
        out = torch.empty((n, out_h, out_w, 3), dtype=torch.uint8, device=src.device)
    slot_t = None
    if slots is not None:
        slot_t = torch.as_tensor(slots, dtype=torch.int32).to(src.device).contiguous()
    if hs == out_h and ws == out_w:
        if slot_t is None:
            out.copy_(src)
        else:
            out[slot_t.long()] = src
        return out
    hb = hk = vb = vk = None
    hks = vks = 0
    if ws != out_w:
        hb, hk, hks = _device_tables(ws, out_w, src.device)
    if hs != out_h:
        vb, vk, vks = _device_tables(hs, out_h, src.device)
    need = int(lib.dd_resize_workspace_bytes(n, hs, ws, out_h, out_w))
    work = torch.empty(max(need, 1), dtype=torch.uint8, device=src.device)
    L.check(lib.dd_resize_bicubic(abi.ptr(src), n, hs, ws, abi.ptr(out), abi.ptr(slot_t), out_h, out_w, abi.ptr(hb), abi.ptr(hk), hks, abi.ptr(vb), abi.ptr(vk), vks,
                                  abi.ptr(work), need, L.current_stream()), "dd_resize_bicubic")
    return out
