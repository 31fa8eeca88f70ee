"""Test infrastructure (oracle): the arithmetic of csrc/dd_conv_mfma.hip restated in NumPy.

dd_conv3x3_mfma computes the reference's fp32 convolutions (reference networks/motion_decoder.py:24-33,57-66, torch.nn.Conv2d ->
F.conv2d in fp32) on the bf16 matrix pipe: every fp32 operand is split into three bf16 pieces and every product is formed from six
partial products accumulated in fp32.  This file states that arithmetic on the CPU so that its two claims can be checked without a
GPU (tests/test_split_bf16.py): the split is EXACT (x == x1 + x2 + x3 in fp32), and a dot product formed from the six partial
products is as close to the float64 result as an fp32 dot product.  Only tests/ may import this module.
"""
import numpy as np


def bf16_round(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 -- what v_cvt_pk_bf16_f32 does on finite values."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x):
    """x (fp32) -> (x1, x2, x3), each representable in bf16, with x1 + x2 + x3 == x exactly (csrc/dd_conv_mfma.hip: split2)."""
    x = np.asarray(x, dtype=np.float32)
    x1 = bf16_round(x)
    r1 = (x - x1).astype(np.float32)            # exact: at most 16 significant bits
    x2 = bf16_round(r1)
    r2 = (r1 - x2).astype(np.float32)           # exact: at most 8 significant bits
    x3 = bf16_round(r2)                         # == r2
    return x1, x2, x3


# the six partial products of the kernel, small ones first (piece index of a, piece index of b)
PRODUCTS = ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0))


def dot_split(a, b, k_block=16):
    """sum_k a[..., k] * b[..., k] the way the kernel forms it: K in blocks of 16 (one MFMA), per block six partial products, each
    block product added to an fp32 accumulator.  The sum INSIDE a block is taken in float64 and rounded once (the matrix instruction
    keeps more than fp32 inside a block; the accumulator between instructions is fp32)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    pa, pb = split3(a), split3(b)
    acc = np.zeros(a.shape[:-1], dtype=np.float32)
    K = a.shape[-1]
    for k0 in range(0, K, k_block):
        sl = slice(k0, min(k0 + k_block, K))
        for ia, ib in PRODUCTS:
            part = (pa[ia][..., sl].astype(np.float64) * pb[ib][..., sl].astype(np.float64)).sum(-1)
            acc = (acc.astype(np.float64) + part).astype(np.float32)
    return acc


def dot_fp32(a, b):
    """The yardstick: an fp32 FMA chain over k (what an fp32 convolution kernel does per output element)."""
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    acc = np.zeros(a.shape[:-1], dtype=np.float32)
    for k in range(a.shape[-1]):
        acc = (acc.astype(np.float64) + a[..., k].astype(np.float64) * b[..., k].astype(np.float64)).astype(np.float32)
    return acc
