"""Host side of the device JPEG decoder (csrc/dd_jpeg.hip): marker-segment parsing into the DDJpegHeader record the kernels read,
and the batch decode call.  Replaces the PIL decode of datasets/base_dataset.py:13-18 (`pil_loader`) for files that are already at
the training resolution (the reference's `downsample` image type): the DataLoader workers then only READ the files."""
import ctypes as C
import struct

import numpy as np
import torch

from . import abi
from . import lib as L

HEADER_BYTES = C.sizeof(abi.DDJpegHeader)
_ZIGZAG = (0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
           35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63)


class UnsupportedJpeg(ValueError):
    """The file is a valid image but not one the device decoder handles (progressive, arithmetic-coded, 12-bit, CMYK, sub-sampled
    luma, several scans ...): decode it on the host instead."""


def parse_header(data):
    """JPEG file bytes -> (header record as a uint8 array of HEADER_BYTES, geometry (width, height, ncomp, h[3], v[3])).
    Walks the marker segments up to the first start-of-scan (ITU T.81 B.2)."""
    if len(data) < 4 or data[0] != 0xFF or data[1] != 0xD8:
        raise UnsupportedJpeg("not a JPEG stream")
    hd = abi.DDJpegHeader()
    have_q, have_h = set(), set()
    comps, i, n = None, 2, len(data)
    while i + 4 <= n:
        if data[i] != 0xFF:
            raise UnsupportedJpeg("marker expected at byte %d" % i)
        m = data[i + 1]
        if m == 0xFF:
            i += 1
            continue
        if m == 0xD8 or m == 0x01 or 0xD0 <= m <= 0xD7:
            i += 2
            continue
        (seg_len,) = struct.unpack_from(">H", data, i + 2)
        seg = memoryview(data)[i + 4:i + 2 + seg_len]
        if m == 0xDB:
            j = 0
            while j < len(seg):
                pq, tq = seg[j] >> 4, seg[j] & 15
                if tq > 3:
                    raise UnsupportedJpeg("quantisation table id %d" % tq)
                for k in range(64):
                    hd.qt[tq][_ZIGZAG[k]] = (seg[j + 1 + 2 * k] << 8 | seg[j + 2 + 2 * k]) if pq else seg[j + 1 + k]
                j += 129 if pq else 65
                have_q.add(tq)
        elif m == 0xC4:
            j = 0
            while j < len(seg):
                tc, th = seg[j] >> 4, seg[j] & 15
                if tc > 1 or th > 1:
                    raise UnsupportedJpeg("Huffman table class/id %d/%d (baseline allows two of each)" % (tc, th))
                t = 2 * tc + th
                count = 0
                for k in range(16):
                    hd.bits[t][k] = seg[j + 1 + k]
                    count += seg[j + 1 + k]
                if count > 256:
                    raise UnsupportedJpeg("corrupt Huffman table")
                for k in range(count):
                    hd.vals[t][k] = seg[j + 17 + k]
                j += 17 + count
                have_h.add(t)
        elif m in (0xC0, 0xC1):
            p, hd.height, hd.width, nc = struct.unpack_from(">BHHB", seg, 0)
            if p != 8:
                raise UnsupportedJpeg("%d-bit samples" % p)
            if nc not in (1, 3):
                raise UnsupportedJpeg("%d components" % nc)
            hd.ncomp = nc
            comps = [(seg[6 + 3 * k], seg[7 + 3 * k] >> 4, seg[7 + 3 * k] & 15, seg[8 + 3 * k]) for k in range(nc)]
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise UnsupportedJpeg("not a baseline Huffman JPEG (SOF%d)" % (m - 0xC0))
        elif m == 0xDD:
            (hd.restart_interval,) = struct.unpack_from(">H", seg, 0)
        elif m == 0xDA:
            if comps is None:
                raise UnsupportedJpeg("scan before frame header")
            ns = seg[0]
            if ns != len(comps):
                raise UnsupportedJpeg("%d of %d components in the first scan (non-interleaved)" % (ns, len(comps)))
            ids = [c[0] for c in comps]
            for k in range(ns):
                if seg[1 + 2 * k] != ids[k]:
                    raise UnsupportedJpeg("scan component order differs from the frame's")
                hd.td[k], hd.ta[k] = seg[2 + 2 * k] >> 4, seg[2 + 2 * k] & 15
                if hd.td[k] not in have_h or (2 + hd.ta[k]) not in have_h:
                    raise UnsupportedJpeg("scan refers to an undefined Huffman table")
            for k, (_, h, v, tq) in enumerate(comps):
                hd.h[k], hd.v[k], hd.tq[k] = h, v, tq
                if tq not in have_q:
                    raise UnsupportedJpeg("component refers to an undefined quantisation table")
            if len(comps) == 3:
                if (comps[1][1], comps[1][2], comps[2][1], comps[2][2]) != (1, 1, 1, 1) or comps[0][1] not in (1, 2) or comps[0][2] not in (1, 2):
                    raise UnsupportedJpeg("sampling factors %r" % ([c[1:3] for c in comps],))
            hd.data_offset, hd.data_end = i + 2 + seg_len, n
            geom = (int(hd.width), int(hd.height), int(hd.ncomp), tuple(int(x) for x in hd.h), tuple(int(x) for x in hd.v))
            return np.frombuffer(bytes(hd), dtype=np.uint8).copy(), geom
        i += 2 + seg_len
    raise UnsupportedJpeg("no start-of-scan marker")


def decode_batch(data, headers, height, width, ncomp=3, h=(2, 1, 1), v=(2, 1, 1)):
    """data (..., cap) uint8 and headers (..., HEADER_BYTES) uint8 on the GPU, every image `width` x `height` with the given
    sampling -> (..., height, width, 3) uint8 RGB, bit for bit what PIL's decode returns.  Three launches for the whole batch."""
    if not data.is_cuda:
        raise L.DynamoHipError("decode_batch runs on the GPU (hosts decode with PIL)")
    lib = L.load()
    lead = data.shape[:-1]
    data = data.reshape(-1, data.shape[-1]).contiguous()
    headers = headers.reshape(-1, HEADER_BYTES).contiguous()
    n = data.shape[0]
    hv = (C.c_int * 3)(*h), (C.c_int * 3)(*v)
    need = lib.dd_jpeg_workspace_bytes(n, height, width, ncomp, hv[0], hv[1])
    if need == 0:
        raise L.DynamoHipError("unsupported JPEG geometry %r" % ((height, width, ncomp, h, v),))
    ws = torch.empty(need, dtype=torch.uint8, device=data.device)
    rgb = torch.empty((n, height, width, 3), dtype=torch.uint8, device=data.device)
    L.check(lib.dd_jpeg_decode(abi.ptr(data), data.shape[1], abi.ptr(headers), n, height, width, ncomp, hv[0], hv[1], abi.ptr(rgb), abi.ptr(ws), need,
                               L.current_stream()), "dd_jpeg_decode")
    return rgb.view(*lead, height, width, 3)
