"""ORACLE (test infrastructure, not product code) -- baseline JPEG decoding restated in numpy / plain Python, arithmetic for
arithmetic what libjpeg(-turbo) executes under PIL's defaults, which is what the reference's loader runs
(datasets/base_dataset.py:13-18 `pil_loader`: Image.open(...).convert('RGB')):

  * sequential Huffman entropy decoding of one interleaved scan (ITU T.81 F.2.2; libjpeg jdhuff.c), restart intervals honoured;
  * de-quantisation and the "ISLOW" integer inverse DCT (jidctint.c: CONST_BITS 13, PASS1_BITS 2, the default dct_method);
  * "fancy" (triangle-filter) chroma up-sampling for 2x1 / 2x2 sub-sampled chroma (jdsample.c h2v1 / h2v2_fancy_upsample, the
    default do_fancy_upsampling) -- edge rows / columns replicated as libjpeg's context rows do;
  * YCbCr -> RGB with the 16-bit fixed-point tables of jdcolor.c.

Third-party algorithm, not under /root/reference: libjpeg-turbo (bundled with Pillow 12.2 in this image; the reference pins no
version).  PINNED here by execution: tests/test_jpeg.py compares this restatement with PIL's own decode, bit for bit, on
JPEGs encoded in the test (4:2:0 / 4:2:2 / 4:4:4, several qualities, with and without restart markers, odd sizes) and on the
six tiny_kitti frames (tests/golden/tiny_kitti_jpeg/).

Only tests/ may import this file; the product path is dynamo-depth_amd/csrc/dd_jpeg.hip behind hipops.jpeg.
"""
import struct

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63])


class JpegHeader:
    """What the decoder needs from the marker segments (T.81 B.2): frame size, per-component sampling / table selectors,
    quantisation tables in natural order, Huffman table specifications, restart interval, offset of the entropy-coded data."""

    def __init__(self, data):
        self.qt = {}
        self.huff = {}          # (class, id) -> (bits[16], vals)
        self.restart = 0
        self.components = []    # (id, h, v, tq)
        self.scan = None        # [(component index, td, ta)]
        i = 2
        if data[:2] != b"\xff\xd8":
            raise ValueError("not a JPEG")
        while i < len(data):
            if data[i] != 0xFF:
                raise ValueError("marker expected at %d" % i)
            m = data[i + 1]
            if m == 0xFF:
                i += 1
                continue
            if m == 0xD8 or 0xD0 <= m <= 0xD7 or m == 0x01:
                i += 2
                continue
            (L,) = struct.unpack(">H", data[i + 2:i + 4])
            seg = data[i + 4:i + 2 + L]
            if m == 0xDB:
                j = 0
                while j < len(seg):
                    pq, tq = seg[j] >> 4, seg[j] & 15
                    if pq:
                        vals = np.frombuffer(seg[j + 1:j + 129], dtype=">u2").astype(np.int32)
                        j += 129
                    else:
                        vals = np.frombuffer(seg[j + 1:j + 65], dtype=np.uint8).astype(np.int32)
                        j += 65
                    nat = np.zeros(64, np.int32)
                    nat[ZIGZAG] = vals
                    self.qt[tq] = nat
            elif m == 0xC4:
                j = 0
                while j < len(seg):
                    tc, th = seg[j] >> 4, seg[j] & 15
                    bits = list(seg[j + 1:j + 17])
                    n = sum(bits)
                    self.huff[(tc, th)] = (bits, list(seg[j + 17:j + 17 + n]))
                    j += 17 + n
            elif m == 0xC0 or m == 0xC1:
                p, self.height, self.width, nc = struct.unpack(">BHHB", seg[:6])
                if p != 8:
                    raise ValueError("only 8-bit samples")
                self.components = [(seg[6 + 3 * k], seg[7 + 3 * k] >> 4, seg[7 + 3 * k] & 15, seg[8 + 3 * k]) for k in range(nc)]
            elif m in (0xC2, 0xC3, 0xC5, 0xC6, 0xC7, 0xC9, 0xCA, 0xCB, 0xCD, 0xCE, 0xCF):
                raise ValueError("not a baseline JPEG (SOF%d)" % (m - 0xC0))
            elif m == 0xDD:
                (self.restart,) = struct.unpack(">H", seg[:2])
            elif m == 0xDA:
                ns = seg[0]
                ids = [c[0] for c in self.components]
                self.scan = [(ids.index(seg[1 + 2 * k]), seg[2 + 2 * k] >> 4, seg[2 + 2 * k] & 15) for k in range(ns)]
                self.data_offset = i + 2 + L
                return
            i += 2 + L
        raise ValueError("no scan")


def _derived(bits, vals):
    """jdhuff.c jpeg_make_d_derived_tbl: canonical codes -> (mincode, maxcode, valptr) per length."""
    code, k = 0, 0
    mincode, maxcode, valptr = [0] * 17, [-1] * 18, [0] * 17
    for l in range(1, 17):
        if bits[l - 1]:
            valptr[l] = k
            mincode[l] = code
            code += bits[l - 1]
            k += bits[l - 1]
            maxcode[l] = code - 1
        code <<= 1
    maxcode[17] = 0xFFFFF
    return mincode, maxcode, valptr, vals


class _Bits:
    def __init__(self, data, pos):
        self.d, self.p, self.buf, self.n = data, pos, 0, 0
        self.marker = None

    def _fill(self):
        while self.n <= 24:
            if self.marker is not None or self.p >= len(self.d):
                b = 0
            else:
                b = self.d[self.p]
                self.p += 1
                if b == 0xFF:
                    nxt = self.d[self.p] if self.p < len(self.d) else 0xD9
                    self.p += 1
                    if nxt != 0:
                        self.marker = nxt          # a marker: feed zeros from here (jdhuff.c does the same)
                        b = 0
            self.buf = (self.buf << 8) | b
            self.n += 8

    def get(self, k):
        if k == 0:
            return 0
        if self.n < k:
            self._fill()
        v = (self.buf >> (self.n - k)) & ((1 << k) - 1)
        self.n -= k
        self.buf &= (1 << self.n) - 1
        return v

    def decode(self, tbl):
        mincode, maxcode, valptr, vals = tbl
        code, l = self.get(1), 1
        while code > maxcode[l]:
            code = (code << 1) | self.get(1)
            l += 1
            if l > 16:
                return 0
        return vals[valptr[l] + code - mincode[l]]

    def restart(self):
        """Discard the remaining bits, step over the RSTn marker."""
        self.buf, self.n = 0, 0
        if self.marker is None:
            while self.p + 1 < len(self.d) and not (self.d[self.p] == 0xFF and 0xD0 <= self.d[self.p + 1] <= 0xD7):
                self.p += 1
            self.p += 2
        self.marker = None


def _extend(v, s):
    return v - ((1 << s) - 1) if s and v < (1 << (s - 1)) else v


def decode_coefficients(data, hdr):
    """-> per component an int32 array (blocks_y, blocks_x, 64) of QUANTISED coefficients in natural order (padded to whole MCUs)."""
    hmax = max(c[1] for c in hdr.components)
    vmax = max(c[2] for c in hdr.components)
    mcux, mcuy = -(-hdr.width // (8 * hmax)), -(-hdr.height // (8 * vmax))
    if len(hdr.scan) != len(hdr.components):
        raise ValueError("only one interleaved scan with all components")
    coefs = [np.zeros((mcuy * c[2], mcux * c[1], 64), np.int32) for c in hdr.components]
    dc_t = {k[1]: _derived(*v) for k, v in hdr.huff.items() if k[0] == 0}
    ac_t = {k[1]: _derived(*v) for k, v in hdr.huff.items() if k[0] == 1}
    br = _Bits(data, hdr.data_offset)
    pred = [0] * len(hdr.components)
    count = 0
    for my in range(mcuy):
        for mx in range(mcux):
            if hdr.restart and count and count % hdr.restart == 0:
                br.restart()
                pred = [0] * len(hdr.components)
            count += 1
            for ci, td, ta in hdr.scan:
                _, h, v, _ = hdr.components[ci]
                for by in range(v):
                    for bx in range(h):
                        blk = coefs[ci][my * v + by, mx * h + bx]
                        s = br.decode(dc_t[td])
                        pred[ci] += _extend(br.get(s), s)
                        blk[0] = pred[ci]
                        k = 1
                        while k < 64:
                            rs = br.decode(ac_t[ta])
                            r, s = rs >> 4, rs & 15
                            if s:
                                k += r
                                blk[ZIGZAG[k]] = _extend(br.get(s), s)
                                k += 1
                            elif r == 15:
                                k += 16
                            else:
                                break
    return coefs


# ---- jidctint.c ------------------------------------------------------------------------------------------------------------
CONST_BITS, PASS1_BITS = 13, 2
F = dict(f0_298=2446, f0_390=3196, f0_541=4433, f0_765=6270, f0_899=7373, f1_175=9633, f1_501=12299, f1_847=15137, f1_961=16069,
         f2_053=16819, f2_562=20995, f3_072=25172)


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _idct_1d(v, shift_even):
    """One pass of jidctint.c over the LAST axis of v (int64 arrays); returns the eight outputs before the descale."""
    z2, z3 = v[..., 2], v[..., 6]
    z1 = (z2 + z3) * F["f0_541"]
    tmp2 = z1 + z3 * (-F["f1_847"])
    tmp3 = z1 + z2 * F["f0_765"]
    z2, z3 = v[..., 0], v[..., 4]
    tmp0 = (z2 + z3) << shift_even
    tmp1 = (z2 - z3) << shift_even
    tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
    tmp0, tmp1, tmp2, tmp3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
    z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
    z5 = (z3 + z4) * F["f1_175"]
    tmp0, tmp1, tmp2, tmp3 = tmp0 * F["f0_298"], tmp1 * F["f2_053"], tmp2 * F["f3_072"], tmp3 * F["f1_501"]
    z1, z2, z3, z4 = z1 * (-F["f0_899"]), z2 * (-F["f2_562"]), z3 * (-F["f1_961"]) + z5, z4 * (-F["f0_390"]) + z5
    tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
    return np.stack([tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3], -1)


def idct_islow(coefs, qt):
    """(by, bx, 64) quantised coefficients -> (by*8, bx*8) uint8 samples."""
    by, bx, _ = coefs.shape
    blk = (coefs.astype(np.int64) * qt.astype(np.int64)).reshape(by, bx, 8, 8)           # [row u][col v]
    # pass 1: columns (the 1-D transform runs along the row index u for every column v)
    cols = np.swapaxes(blk, -1, -2)                                                          # [col][row]
    ws = _descale(_idct_1d(cols, CONST_BITS), CONST_BITS - PASS1_BITS)                       # [col][out row]
    ws = np.swapaxes(ws, -1, -2)                                                             # [out row][col]
    # pass 2: rows
    out = _descale(_idct_1d(ws, CONST_BITS), CONST_BITS + PASS1_BITS + 3)
    out = np.clip(out + 128, 0, 255).astype(np.uint8)
    return out.transpose(0, 2, 1, 3).reshape(by * 8, bx * 8)


# ---- jdsample.c --------------------------------------------------------------------------------------------------------------
def _h2_fancy(rows, bias_even, bias_odd, centre_weight_shift):
    raise NotImplementedError


def upsample_h2v1_fancy(plane):
    p = plane.astype(np.int32)
    h, w = p.shape
    out = np.empty((h, 2 * w), np.int32)
    left = np.concatenate([p[:, :1], p[:, :-1]], 1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], 1)
    out[:, 0::2] = (p * 3 + left + 1) >> 2
    out[:, 1::2] = (p * 3 + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out.astype(np.uint8)


def upsample_h2v2_fancy(plane):
    p = plane.astype(np.int32)
    h, w = p.shape
    above = np.concatenate([p[:1], p[:-1]], 0)
    below = np.concatenate([p[1:], p[-1:]], 0)
    out = np.empty((2 * h, 2 * w), np.int32)
    for v, other in ((0, above), (1, below)):
        colsum = p * 3 + other                                       # thiscolsum per input column
        last = np.concatenate([colsum[:, :1], colsum[:, :-1]], 1)
        nxt = np.concatenate([colsum[:, 1:], colsum[:, -1:]], 1)
        even = (colsum * 3 + last + 8) >> 4
        odd = (colsum * 3 + nxt + 7) >> 4
        even[:, 0] = (colsum[:, 0] * 4 + 8) >> 4
        odd[:, -1] = (colsum[:, -1] * 4 + 7) >> 4
        out[v::2, 0::2] = even
        out[v::2, 1::2] = odd
    return out.astype(np.uint8)


def upsample_h1v2_fancy(plane):
    p = plane.astype(np.int32)
    above = np.concatenate([p[:1], p[:-1]], 0)
    below = np.concatenate([p[1:], p[-1:]], 0)
    out = np.empty((2 * p.shape[0], p.shape[1]), np.int32)
    out[0::2] = (p * 3 + above + 1) >> 2
    out[1::2] = (p * 3 + below + 2) >> 2
    return out.astype(np.uint8)


# ---- jdcolor.c ---------------------------------------------------------------------------------------------------------------
def ycc_to_rgb(y, cb, cr):
    y, cb, cr = y.astype(np.int32), cb.astype(np.int32) - 128, cr.astype(np.int32) - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8)


def decode(data):
    """Baseline JPEG bytes -> (H, W, 3) uint8 RGB (or (H, W) for greyscale), as PIL's Image.open(...).convert('RGB') returns."""
    hdr = JpegHeader(data)
    coefs = decode_coefficients(data, hdr)
    hmax = max(c[1] for c in hdr.components)
    vmax = max(c[2] for c in hdr.components)
    planes = []
    for (cid, h, v, tq), cf in zip(hdr.components, coefs):
        plane = idct_islow(cf, hdr.qt[tq])
        # crop the component to its true down-sampled size before up-sampling (libjpeg up-samples `downsampled_width` columns;
        # the edge replication of the triangle filter acts at the true edge)
        cw, ch = -(-hdr.width * h // hmax), -(-hdr.height * v // vmax)
        plane = plane[:ch, :cw]
        if (h, v) == (hmax, vmax):
            pass
        elif hmax == 2 * h and vmax == 2 * v:
            plane = upsample_h2v2_fancy(plane)
        elif hmax == 2 * h and vmax == v:
            plane = upsample_h2v1_fancy(plane)
        elif hmax == h and vmax == 2 * v:
            plane = upsample_h1v2_fancy(plane)
        else:
            raise ValueError("sampling factors %r not supported" % ((h, v, hmax, vmax),))
        planes.append(plane[:hdr.height, :hdr.width])
    if len(planes) == 1:
        return np.stack([planes[0]] * 3, -1)
    return ycc_to_rgb(*planes)
