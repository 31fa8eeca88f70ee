"""dd_pw_gemm / dd_mlp_pack / dd_gelu_pair (csrc/dd_pw_gemm.hip) against float64: LiteMono's point-wise Linears with the exact GELU
between them (reference networks/depth_encoder.py:200-203,216-224,262-272) computed on the bf16 matrix pipe from three bf16 pieces per
fp32 operand must be as accurate as the fp32 BLAS path -- the yardstick is torch's own fp32 result on the same inputs."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
os.environ.setdefault("DD_MLP", "1")          # the path is opt-in (hipops/functions.py: mlp_ok): these tests are about it


def _err(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


def _gemm(x, w, b, gelu):
    """raw C ABI: y = act(x) . w^T + b through dd_mlp_pack (w as the first operand's slot) + dd_pw_gemm"""
    from hipops import lib as L
    lib = L.load()
    M, K = x.shape
    N = w.shape[0]
    nb = int(lib.dd_pw_gemm_pack_bytes(N, K))
    pack = torch.empty(nb // 4, dtype=torch.float32, device="cuda")
    st = L.current_stream()
    # dd_mlp_pack's first slot packs a (hidden, C) matrix for N = hidden, K = C: use it for any (N, K)
    dummy = torch.zeros(K, N, device="cuda")
    L.check(lib.dd_mlp_pack(w.data_ptr(), w.stride(0), w.stride(1), dummy.data_ptr(), dummy.stride(0), dummy.stride(1), K, N, pack.data_ptr(), None, None,
                            None, None, st), "dd_mlp_pack")
    y = torch.empty(M, N, device="cuda")
    L.check(lib.dd_pw_gemm(x.data_ptr(), pack.data_ptr(), None if b is None else b.data_ptr(), M, K, N, int(gelu), y.data_ptr(), st), "dd_pw_gemm")
    return y


# (M, K, N, gelu): every dispatch of csrc/dd_pw_gemm.hip -- narrow with all column blocks per wave (2x2, 1x4, 1x7), narrow spread over
# workgroups (few rows), wide in groups of three / two column blocks; ragged rows, a ragged last column block
CASES = [(32768, 384, 64, 1), (32768, 384, 64, 0), (16384, 768, 128, 1), (16400, 1344, 224, 1), (16384, 1344, 224, 0), (1000, 384, 64, 1), (5760, 1344, 224, 1),
         (33000, 64, 384, 0), (23040, 128, 768, 0), (5000, 224, 1344, 0), (4097, 64, 80, 0), (777, 32, 40, 1), (3000, 96, 288, 0)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_gemm_matches_float64(case):
    M, K, N, gelu = case
    g = torch.Generator().manual_seed(sum(case))
    x = (torch.randn(M, K, generator=g) * 1.5).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    y = _gemm(x, w, b, gelu)
    xa = F.gelu(x.double()) if gelu else x.double()
    ref = xa @ w.double().t() + b.double()
    lib32 = (F.gelu(x) if gelu else x) @ w.t() + b
    e_own, e_lib = _err(y, ref), _err(lib32, ref)
    print("gemm %-28s own %.2e  torch fp32 %.2e" % (case, e_own, e_lib))
    assert e_own <= max(2.0 * e_lib, 2e-6), (e_own, e_lib)


def test_transposed_packs_give_the_data_gradients():
    """dd_mlp_pack's third and fourth operand: g . w2 and g_pre . w1 from the weights where they lie"""
    from hipops import lib as L
    lib = L.load()
    C, hid, M = 64, 384, 4096
    g = torch.Generator().manual_seed(5)
    w1 = (torch.randn(hid, C, generator=g) / 8).cuda()
    w2 = (torch.randn(C, hid, generator=g) / 20).cuda()
    nb1, nb2 = int(lib.dd_pw_gemm_pack_bytes(hid, C)), int(lib.dd_pw_gemm_pack_bytes(C, hid))
    packs = torch.empty((2 * (nb1 + nb2)) // 4, device="cuda")
    p0, st = packs.data_ptr(), L.current_stream()
    L.check(lib.dd_mlp_pack(w1.data_ptr(), w1.stride(0), w1.stride(1), w2.data_ptr(), w2.stride(0), w2.stride(1), C, hid, p0, p0 + nb1, p0 + nb1 + nb2,
                            p0 + 2 * nb1 + nb2, None, st), "dd_mlp_pack")
    go = torch.randn(M, C, generator=g).cuda()
    gp = torch.empty(M, hid, device="cuda")
    L.check(lib.dd_pw_gemm(go.data_ptr(), p0 + nb1 + nb2, None, M, C, hid, 0, gp.data_ptr(), st), "g . w2")
    assert _err(gp, go.double() @ w2.double()) < 2e-6
    gpre = torch.randn(M, hid, generator=g).cuda()
    gy = torch.empty(M, C, device="cuda")
    L.check(lib.dd_pw_gemm(gpre.data_ptr(), p0 + 2 * nb1 + nb2, None, M, hid, C, 0, gy.data_ptr(), st), "g_pre . w1")
    assert _err(gy, gpre.double() @ w1.double()) < 2e-6


def test_gelu_pair_is_atens_arithmetic():
    from hipops import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(9)
    pre = (torch.randn(1 << 16, generator=g) * 3).cuda()
    pre[:8] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 20.0, -20.0, 5.5, -5.5])
    go = torch.randn(1 << 16, generator=g).cuda()
    gi, post = go.clone(), torch.empty_like(pre)
    L.check(lib.dd_gelu_pair(pre.data_ptr(), gi.data_ptr(), post.data_ptr(), pre.numel(), L.current_stream()), "dd_gelu_pair")
    p = pre.clone().requires_grad_(True)
    ref = F.gelu(p)
    (gref,) = torch.autograd.grad(ref, p, go)
    assert float((post - ref.detach()).abs().max()) <= 2e-7 * float(ref.abs().max())
    assert float((gi - gref).abs().max()) <= 1e-6 * float(gref.abs().max())
    ref64 = F.gelu(pre.double())
    assert _err(post, ref64) < 3e-7


class _Block(torch.nn.Module):
    def __init__(self, C):
        super().__init__()
        self.pwconv1 = torch.nn.Linear(C, 6 * C)
        self.act = torch.nn.GELU()
        self.pwconv2 = torch.nn.Linear(6 * C, C)

    def forward(self, y):
        return self.pwconv2(self.act(self.pwconv1(y)))


@pytest.mark.parametrize("shape", [(3, 48, 160, 64), (9, 24, 80, 128), (1, 130, 129, 64)], ids=lambda s: "x".join(map(str, s)))
def test_block_forward_and_gradients_match_float64(shape):
    """MlpFn (what networks/depth_encoder.py:_mlp_residual calls) against the stock module in float64, torch's fp32 result as the yardstick"""
    from hipops.functions import mlp, mlp_ok
    B, H, W, C = shape
    torch.manual_seed(B + C)
    blk = _Block(C).cuda()
    y = torch.randn(B, H, W, C, device="cuda", requires_grad=True)
    assert mlp_ok(y, blk)
    go = torch.randn(B, H, W, C, device="cuda")
    out = mlp(y, blk)
    own = torch.autograd.grad(out, [y] + list(blk.parameters()), go)
    lib = torch.autograd.grad(blk(y), [y] + list(blk.parameters()), go)
    blk64 = _Block(C).cuda().double()
    blk64.load_state_dict({k: v.double() for k, v in blk.state_dict().items()})
    y64 = y.detach().double().requires_grad_(True)
    out64 = blk64(y64)
    ref = torch.autograd.grad(out64, [y64] + list(blk64.parameters()), go.double())
    e_own, e_lib = _err(out, out64.detach()), _err(blk(y).detach(), out64.detach())
    print("forward %-18s own %.2e  torch fp32 %.2e" % (shape, e_own, e_lib))
    assert e_own <= max(2.0 * e_lib, 2e-6)
    for name, a, b, r in zip(["g_y", "g_w1", "g_b1", "g_w2", "g_b2"], own, lib, ref):
        e_own, e_lib = _err(a, r), _err(b, r)
        print("%-5s %-18s own %.2e  torch fp32 %.2e" % (name, shape, e_own, e_lib))
        assert a.shape == r.shape
        assert e_own <= max(3.0 * e_lib, 3e-6), (name, e_own, e_lib)


def test_forward_only_pass_keeps_nothing_and_matches():
    from hipops.functions import mlp
    torch.manual_seed(3)
    blk = _Block(64).cuda()
    y = torch.randn(2, 48, 160, 64, device="cuda")
    with torch.no_grad():
        a, b = mlp(y, blk), blk(y)
    assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    y2 = y.clone().requires_grad_(True)
    c = mlp(y2, blk)
    assert torch.equal(a, c.detach())          # bit-reproducible, with or without the tape


@pytest.mark.parametrize("shape", [(3, 48, 160, 64), (9, 24, 80, 128), (1, 131, 127, 64), (2, 67, 129, 128)], ids=lambda s: "x".join(map(str, s)))
def test_fused_block_forward_matches_float64(shape):
    """dd_mlp_fwd: both Linears and the GELU in one kernel, the hidden tile in accumulator registers (the contraction order of the
    second GEMM is the accumulator's register order: every hidden unit must still meet its own weight column) -- ragged row counts"""
    from hipops.functions import mlp_fused, mlp_fused_ok
    B, H, W, C = shape
    torch.manual_seed(B * C)
    blk = _Block(C).cuda()
    with torch.no_grad():
        blk.pwconv1.bias.mul_(3.0)
        y = torch.randn(B, H, W, C, device="cuda") * 1.5
        assert mlp_fused_ok(y, blk)
        own, lib = mlp_fused(y, blk), blk(y)
        blk64 = _Block(C).cuda().double()
        blk64.load_state_dict({k: v.double() for k, v in blk.state_dict().items()})
        ref = blk64(y.double())
    e_own, e_lib = _err(own, ref), _err(lib, ref)
    print("fused forward %-18s own %.2e  torch fp32 %.2e" % (shape, e_own, e_lib))
    assert own.shape == ref.shape
    assert e_own <= max(2.0 * e_lib, 2e-6), (e_own, e_lib)
    assert not mlp_fused_ok(y, blk)            # with the tape on it is not this path


@pytest.mark.parametrize("shape", [(2, 48, 160, 64), (3, 24, 80, 128), (1, 47, 161, 64)], ids=lambda s: "x".join(map(str, s)))
def test_training_block_with_the_hidden_tensor_recomputed_matches_float64(shape):
    """MlpRecomputeFn (round 6): the TRAINING pass of a block through dd_mlp_fwd, nothing but the input kept, the pre-activation rebuilt by
    one library GEMM in the backward -- output and all five gradients against the float64 block, beside the stock fp32 autograd path."""
    from hipops import functions as Fn
    B, H, W, Cc = shape
    g0 = torch.Generator().manual_seed(sum(shape))
    y = torch.randn(B, H, W, Cc, generator=g0).cuda().requires_grad_()
    w1 = (torch.randn(6 * Cc, Cc, generator=g0) / Cc ** 0.5).cuda().requires_grad_()
    b1 = (0.1 * torch.randn(6 * Cc, generator=g0)).cuda().requires_grad_()
    w2 = (torch.randn(Cc, 6 * Cc, generator=g0) / (6 * Cc) ** 0.5).cuda().requires_grad_()
    b2 = (0.1 * torch.randn(Cc, generator=g0)).cuda().requires_grad_()
    go = torch.randn(B, H, W, Cc, generator=g0).cuda()
    leaves = (y, w1, b1, w2, b2)
    before = Fn.mlp_recompute_calls()
    out = Fn.MlpRecomputeFn.apply(*leaves)
    assert Fn.mlp_recompute_calls() == before + 1
    own = (out.detach(),) + torch.autograd.grad(out, leaves, go)
    d = [t.detach().double().requires_grad_() for t in leaves]
    ref_out = F.linear(F.gelu(F.linear(d[0], d[1], d[2])), d[3], d[4])
    ref = (ref_out.detach(),) + torch.autograd.grad(ref_out, d, go.double())
    s_out = F.linear(F.gelu(F.linear(y, w1, b1)), w2, b2)
    stock = (s_out.detach(),) + torch.autograd.grad(s_out, leaves, go)
    for name, a, s, r in zip(("out", "g_y", "g_w1", "g_b1", "g_w2", "g_b2"), own, stock, ref):
        e_own, e_stock = _err(a, r), _err(s, r)
        print("%-5s %-14s own %.2e  torch fp32 %.2e" % (name, shape, e_own, e_stock))
        assert e_own <= max(3.0 * e_stock, 3e-6), (name, e_own, e_stock)
