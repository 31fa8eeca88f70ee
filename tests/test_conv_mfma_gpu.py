"""dd_conv3x3_mfma (csrc/dd_conv_mfma.hip) against float64: the motion decoders' 3x3 convolutions (reference
networks/motion_decoder.py:24-33,57-66) computed on the bf16 matrix pipe from three bf16 pieces per fp32 operand must be as accurate
as an fp32 convolution -- the yardstick is MIOpen's fp32 result on the same inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(B, cin, cout, H, W, pad, seed, bias=True, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, cin, H, W, generator=g) * scale).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).cuda()
    b = torch.randn(cout, generator=g).cuda() if bias else None
    return x, w, b


def _err(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


# (B, cin, cout, H, W, pad): the half- and quarter-resolution levels of the motion decoders (64 + 3 / 64 + 1 channels padded to 72),
# a ragged image (partial tiles on both axes), a pre-padded input (pad 0), more than 96 output channels (two N tiles), few channels
CASES = [(2, 64, 64, 96, 320, 1), (2, 72, 64, 96, 320, 1), (1, 64, 72, 48, 160, 1), (3, 128, 128, 24, 80, 1), (2, 32, 16, 19, 45, 1),
         (2, 64, 64, 50, 66, 0), (1, 256, 256, 12, 40, 1), (1, 16, 160, 17, 33, 1), (1, 20, 24, 9, 40, 1)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_forward_and_data_gradient_match_float64(case):
    from hipops.functions import mfma_conv
    B, cin, cout, H, W, pad = case
    x, w, b = _case(B, cin, cout, H, W, pad, seed=sum(case))
    x.requires_grad_(True)
    y = mfma_conv(x, w, b, pad)
    ref = F.conv2d(x.detach().double(), w.double(), b.double(), padding=pad)
    assert y.shape == ref.shape
    lib32 = F.conv2d(x.detach(), w, b, padding=pad)
    e_own, e_lib = _err(y, ref), _err(lib32, ref)
    print("forward  %-22s own %.2e  library fp32 %.2e" % (case, e_own, e_lib))
    assert e_own <= max(2.0 * e_lib, 2e-6), (e_own, e_lib)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).cuda().contiguous(memory_format=torch.channels_last)
    (gx,) = torch.autograd.grad(y, x, g)
    gref = torch.nn.grad.conv2d_input(x.shape, w.double(), g.double(), padding=pad)
    glib = torch.nn.grad.conv2d_input(x.shape, w, g, padding=pad)
    e_own, e_lib = _err(gx, gref), _err(glib, gref)
    print("data grad %-22s own %.2e  library fp32 %.2e" % (case, e_own, e_lib))
    assert gx.shape == x.shape
    assert e_own <= max(2.0 * e_lib, 2e-6), (e_own, e_lib)


# small images through dd_conv3x3_mfma_flat (flat 256-pixel tiles of the whole batch, split contraction): the deep levels of the encoders
# and motion decoders (B=12 / 24 at 12x40 and 6x20), a tile that ends inside an image, one image smaller than a tile, the widest
# image the window takes (W = 40), a single row, fewer pixels than one M block
# ... and the concatenated inputs of the motion decoders (512 + 3 -> 520, 256 + 3 -> 264 channels: a partial last chunk forward, a
# partial last tile of output channels in the data gradient)
FLAT_CASES = [(12, 512, 512, 6, 20), (12, 256, 256, 12, 40), (24, 256, 256, 12, 40), (3, 64, 128, 7, 13), (1, 128, 64, 5, 40), (2, 64, 64, 1, 9),
              (5, 320, 64, 3, 3), (2, 520, 512, 6, 20), (2, 264, 256, 12, 40)]


@pytest.mark.parametrize("splits", ["auto", "1", "2"])
@pytest.mark.parametrize("case", FLAT_CASES, ids=lambda c: "x".join(map(str, c)))
def test_flat_tiles_forward_and_data_gradient_match_float64(case, splits, monkeypatch):
    from hipops import functions as Fn
    from hipops import lib as L
    B, cin, cout, H, W = case
    if splits != "auto":
        monkeypatch.setenv("DD_FLAT_SPLITS", splits)
        Fn._WS_BYTES.clear()
    lib = L.load()
    assert lib.dd_conv3x3_mfma_flat_supported(B, H, W, cin, cout) and lib.dd_conv3x3_mfma_flat_supported(B, H, W, cout, cin)
    assert Fn._flat_shape(B, H, W, 1, cin, cout)
    x, w, b = _case(B, cin, cout, H, W, 1, seed=sum(case))
    x.requires_grad_(True)
    before = Fn._FLAT_CONV_CALLS[0]
    y = Fn.mfma_conv(x, w, b, 1)
    assert Fn._FLAT_CONV_CALLS[0] == before + 1
    ref = F.conv2d(x.detach().double(), w.double(), b.double(), padding=1)
    lib32 = F.conv2d(x.detach(), w, b, padding=1)
    e_own, e_lib = _err(y, ref), _err(lib32, ref)
    print("flat forward  %-24s splits %-4s own %.2e  library fp32 %.2e" % (case, splits, e_own, e_lib))
    # (one split = one fp32 accumulator over all 9 C terms: 2e-6 at C = 256 -- the chain the tile kernel has too; the default splits it)
    assert y.shape == ref.shape and e_own <= max(2.0 * e_lib, 3e-6), (e_own, e_lib)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).cuda().contiguous(memory_format=torch.channels_last)
    (gx,) = torch.autograd.grad(y, x, g)
    gref = torch.nn.grad.conv2d_input(x.shape, w.double(), g.double(), padding=1)
    glib = torch.nn.grad.conv2d_input(x.shape, w, g, padding=1)
    e_own, e_lib = _err(gx, gref), _err(glib, gref)
    print("flat data grad %-24s splits %-4s own %.2e  library fp32 %.2e" % (case, splits, e_own, e_lib))
    assert gx.shape == x.shape and e_own <= max(2.0 * e_lib, 3e-6), (e_own, e_lib)
    # bit-reproducible (the partial sums of the splits are added in split order)
    y2 = Fn.mfma_conv(x.detach(), w, b, 1)
    assert torch.equal(y2, y.detach())
    Fn._WS_BYTES.clear()


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_weight_gradient_matches_float64(case):
    """dd_conv3x3_mfma_bwd_weight through the C ABI: the pixels are the contraction (tens of thousands of terms per element)."""
    from hipops import lib as L
    from hipops.functions import _p, _dense_nhwc
    B, cin, cout, H, W, pad = case
    if cout % 4:
        pytest.skip("the weight gradient kernel takes output channels in fours")
    x, w, _ = _case(B, cin, cout, H, W, pad, seed=sum(case) + 1)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    g = torch.randn(B, cout, Ho, Wo, generator=torch.Generator().manual_seed(3)).cuda().contiguous(memory_format=torch.channels_last)
    lib = L.load()
    flat = torch.full((cout * 9 * cin,), float("nan"), device="cuda")
    nbytes = int(lib.dd_conv3x3_mfma_wgrad_workspace_bytes(B, Ho, Wo, cin, cout))
    ws = torch.empty(nbytes // 4, device="cuda")
    L.check(lib.dd_conv3x3_mfma_bwd_weight(_p(_dense_nhwc(x)), _p(_dense_nhwc(g)), B, H, W, cin, cout, pad, _p(flat), _p(ws), nbytes, L.current_stream()),
            "dd_conv3x3_mfma_bwd_weight")
    gw = flat.view(cout, 3, 3, cin).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, g.double(), padding=pad)
    lib32 = torch.nn.grad.conv2d_weight(x, w.shape, g, padding=pad)
    e_own, e_lib = _err(gw, ref), _err(lib32, ref)
    print("weight grad %-22s own %.2e  library fp32 %.2e" % (case, e_own, e_lib))
    assert e_own <= max(2.0 * e_lib, 2e-6), (e_own, e_lib)
    # bit-reproducible: a second evaluation gives the same bits
    flat2 = torch.empty_like(flat)
    L.check(lib.dd_conv3x3_mfma_bwd_weight(_p(_dense_nhwc(x)), _p(_dense_nhwc(g)), B, H, W, cin, cout, pad, _p(flat2), _p(ws), nbytes, L.current_stream()),
            "dd_conv3x3_mfma_bwd_weight")
    assert torch.equal(flat, flat2)


def test_wide_dynamic_range_and_exact_small_integers():
    """Pieces of very different magnitude in one dot product, and a case every arithmetic gets exactly: small integers."""
    from hipops.functions import mfma_conv
    x, w, b = _case(1, 64, 32, 16, 64, 1, seed=5)
    x = x * torch.exp(4.0 * torch.randn(x.shape, generator=torch.Generator().manual_seed(2))).cuda()
    y = mfma_conv(x, w, b, 1)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    lib32 = F.conv2d(x, w, b, padding=1)
    assert _err(y, ref) <= max(2.0 * _err(lib32, ref), 2e-6)
    xi = torch.randint(-7, 8, (1, 32, 16, 64)).float().cuda().contiguous(memory_format=torch.channels_last)
    wi = torch.randint(-3, 4, (32, 32, 3, 3)).float().cuda()
    yi = mfma_conv(xi, wi, None, 1)
    assert torch.equal(yi, F.conv2d(xi.double(), wi.double(), padding=1).float())


def test_every_piece_counts():
    """Values whose bf16 head is the same and that differ only in the second / third piece must give different results: x = 1 + 2^-10
    (second piece) and x = 1 + 2^-20 (third piece) against weights of ones."""
    from hipops.functions import mfma_conv
    for eps in (2.0 ** -10, 2.0 ** -20):
        x = torch.full((1, 16, 8, 32), 1.0 + eps).cuda().contiguous(memory_format=torch.channels_last)
        w = torch.zeros(16, 16, 3, 3).cuda()
        w[:, :, 1, 1] = 1.0 / 16
        y = mfma_conv(x, w, None, 1)
        assert float((y - (1.0 + eps)).abs().max()) == 0.0, eps
        # and the weight's pieces
        x1 = torch.ones((1, 16, 8, 32)).cuda().contiguous(memory_format=torch.channels_last)
        w2 = torch.zeros(16, 16, 3, 3).cuda()
        w2[:, :, 1, 1] = (1.0 + eps) / 16
        y2 = mfma_conv(x1, w2, None, 1)
        assert float((y2 - (1.0 + eps)).abs().max()) == 0.0, eps


def test_module_gradients_match_library_convolution():
    """Conv2d's forward takes the hook; weight / bias / input gradients against the stock path."""
    import os
    from networks.layers import Conv2d
    torch.manual_seed(0)
    conv = Conv2d(64, 64, 3, padding=1).cuda()
    x = torch.randn(2, 64, 96, 320, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    from hipops.functions import mfma_conv_calls
    n0 = mfma_conv_calls()
    y = conv(x)
    assert mfma_conv_calls() == n0 + 1
    g = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, (x, conv.weight, conv.bias), g)
    os.environ["DD_STOCK_MFMA_CONV"] = "1"
    try:
        y0 = conv(x)
        gx0, gw0, gb0 = torch.autograd.grad(y0, (x, conv.weight, conv.bias), g)
    finally:
        del os.environ["DD_STOCK_MFMA_CONV"]
    assert mfma_conv_calls() == n0 + 1
    for a, r, name in ((y, y0, "y"), (gx, gx0, "gx"), (gw, gw0, "gw"), (gb, gb0, "gb")):
        rel = float((a - r).norm() / r.norm())
        assert rel < 2e-6, (name, rel)


def test_packs_of_many_layers_in_one_launch_are_the_single_packs():
    """dd_conv3x3_mfma_pack_many writes, for every layer of its job table, exactly the bytes dd_conv3x3_mfma_pack writes for that layer:
    layers of different shapes (one, two and three 32-channel blocks per tile, partial chunks), a channels-last and a contiguous weight,
    a layer without a data-gradient pack."""
    from hipops import lib as L
    from hipops.functions import _p
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 64), (32, 72), (96, 20), (256, 264), (16, 160), (512, 512)]
    weights = []
    for n, (cout, cin) in enumerate(shapes):
        w = torch.randn(cout, cin, 3, 3, generator=g).cuda()
        weights.append(w.contiguous(memory_format=torch.channels_last) if n % 2 else w)
    stream = L.current_stream()

    def buf(n_out, k_in, fill):
        return torch.full((lib.dd_conv3x3_mfma_pack_bytes(n_out, k_in) // 4,), fill, dtype=torch.float32, device="cuda")
    single, many, jobs, owner, first = [], [], [], [], 0
    assert lib.dd_conv3x3_mfma_pack_many_job_words() == 10
    for n, w in enumerate(weights):
        cout, cin = w.shape[:2]
        want_b = n != 2
        sf, sb = buf(cout, cin, 1.0), (buf(cin, cout, 1.0) if want_b else None)
        sw = w.stride()
        L.check(lib.dd_conv3x3_mfma_pack(_p(w), sw[0], sw[1], sw[2], sw[3], cout, cin, _p(sf), _p(sb), stream), "pack")
        single.append((sf, sb))
        mf, mb = buf(cout, cin, 2.0), (buf(cin, cout, 2.0) if want_b else None)
        many.append((mf, mb))
        jobs.append([w.data_ptr(), sw[0], sw[1], sw[2], sw[3], cout, cin, mf.data_ptr(), mb.data_ptr() if want_b else 0, first])
        nb = lib.dd_conv3x3_mfma_pack_many_blocks(cout, cin, 1, 1 if want_b else 0)
        owner += [n] * nb
        first += nb
    jobs_t, owner_t = torch.tensor(jobs, dtype=torch.int64).cuda(), torch.tensor(owner, dtype=torch.int32).cuda()
    L.check(lib.dd_conv3x3_mfma_pack_many(_p(jobs_t), _p(owner_t), first, stream), "pack_many")
    torch.cuda.synchronize()
    for n, ((sf, sb), (mf, mb)) in enumerate(zip(single, many)):
        assert torch.equal(sf.view(torch.int32), mf.view(torch.int32)), ("forward pack", shapes[n])
        if sb is not None:
            assert torch.equal(sb.view(torch.int32), mb.view(torch.int32)), ("data-gradient pack", shapes[n])
    assert lib.dd_conv3x3_mfma_pack_many(None, _p(owner_t), first, stream) != 0


def test_a_network_packs_once_per_forward_and_never_uses_a_stale_pack(monkeypatch):
    """PackSet (hipops/functions.py): a module registered with pack_weights_once_per_forward makes the packs of its layers in one launch at
    the top of its forward pass from the second pass on; results and gradients are bit for bit those of the layer-by-layer packs
    (DD_PACK_MANY=0) over passes with weight updates in between, a tape-free pass between a forward and its backward, a layer called
    outside the module's forward, and a layer used twice in one pass."""
    import torch.nn as nn
    from networks.layers import Conv2d
    from hipops import functions as Fn

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.c = Conv2d(32, 64, 3, padding=1), Conv2d(64, 64, 3, padding=1), Conv2d(64, 32, 3, padding=1)

        def forward(self, x):
            y = self.b(torch.relu(self.a(x)))
            return self.c(torch.relu(self.b(torch.relu(y))))          # layer b twice

    def run(batched):
        monkeypatch.setenv("DD_PACK_MANY", "1" if batched else "0")
        torch.manual_seed(3)
        net = Net().cuda().to(memory_format=torch.channels_last)
        if batched:
            Fn.pack_weights_once_per_forward(net)
        x = torch.randn(2, 32, 64, 192, device="cuda").contiguous(memory_format=torch.channels_last)
        out = []
        for step in range(4):
            y = net(x)
            with torch.no_grad():
                side = net(x * 0.5)                                      # a tape-free pass between the forward and its backward
            loss = (y * y).mean()
            grads = torch.autograd.grad(loss, list(net.parameters()))
            alone = net.a(x)                                             # outside the module's forward: packs alone
            out += [y.detach().clone(), side.clone(), alone.detach().clone()] + [g.clone() for g in grads]
            with torch.no_grad():
                for p, gp in zip(net.parameters(), grads):
                    p.add_(gp, alpha=-0.1)                               # the weights move: the next pass must not see this pass's packs
        return out
    n0 = Fn.pack_many_launches()
    got = run(True)
    launches = Fn.pack_many_launches() - n0
    assert launches == 2 * 4 - 1, launches      # a taped and a tape-free pass per step; in the very first pass the layers join
    want = run(False)
    assert Fn.pack_many_launches() - n0 == launches
    assert len(got) == len(want)
    for k, (a, r) in enumerate(zip(got, want)):
        assert torch.equal(a, r), k


def _tf32(t):
    """Operands as TF32 holds them (10 explicit significand bits, round to nearest even): what cuDNN multiplies when
    torch.backends.cudnn.allow_tf32 is on -- PyTorch's default, i.e. the reference's fp32 convolutions on an Ampere-class GPU."""
    bits = t.contiguous().view(torch.int32)
    bits = (bits + 0x0FFF + ((bits >> 13) & 1)) & ~0x1FFF
    return bits.view(torch.float32)


@pytest.mark.parametrize("case", [(2, 64, 64, 96, 320, 1), (1, 72, 64, 50, 66, 0), (1, 256, 256, 12, 40, 1), (3, 128, 96, 24, 80, 1)], ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("precision", ["high", "medium"])
def test_fewer_partial_products_are_what_their_names_say(case, precision):
    """torch.set_float32_matmul_precision("high") -> three partial products per multiply-add (bf16x3): forward, data and weight gradient within
    3e-5 of max|result| of float64 and at least eight times closer to it than the same convolution on TF32-rounded operands (cuDNN's default
    arithmetic for the reference's fp32 convolutions); "medium" -> one product: the float64 convolution of the bf16-ROUNDED operands to fp32
    accumulation error (the operands are rounded, nothing else is lost).  "highest" is every other test of this file."""
    from hipops.functions import mfma_conv, mfma_products
    B, cin, cout, H, W, pad = case
    x, w, b = _case(B, cin, cout, H, W, pad, seed=sum(case) + 1)
    x.requires_grad_(True)
    w.requires_grad_(True)
    before = torch.get_float32_matmul_precision()
    torch.set_float32_matmul_precision(precision)
    try:
        assert mfma_products() == {"high": 3, "medium": 1}[precision]
        y = mfma_conv(x, w, b, pad)
        g = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).cuda().contiguous(memory_format=torch.channels_last)
        gx, gw = torch.autograd.grad(y, (x, w), g)
    finally:
        torch.set_float32_matmul_precision(before)
    assert mfma_products() == 6

    def f64(xx, ww, gg):
        xd, wd = xx.detach().double().requires_grad_(True), ww.detach().double().requires_grad_(True)
        yd = F.conv2d(xd, wd, b.double(), padding=pad)
        gxd, _ = torch.autograd.grad(yd, (xd, wd), gg.double(), retain_graph=True)
        return yd.detach(), gxd, wd, xd
    ref_y, ref_gx, wd, xd = f64(x, w, g)
    ref_gw = torch.autograd.grad(F.conv2d(xd, wd, None, padding=pad), wd, g.double())[0]
    if precision == "high":
        t_y, t_gx, twd, txd = f64(_tf32(x.detach()), _tf32(w.detach()), _tf32(g))
        t_gw = torch.autograd.grad(F.conv2d(txd, twd, None, padding=pad), twd, _tf32(g).double())[0]
        for name, own, ref, tf in (("forward", y, ref_y, t_y), ("data gradient", gx, ref_gx, t_gx), ("weight gradient", gw, ref_gw, t_gw)):
            e_own, e_tf = _err(own, ref), _err(tf, ref)
            print("%-15s %-22s bf16x3 %.2e  TF32 operands %.2e" % (name, case, e_own, e_tf))
            assert e_own < 3e-5 and e_own * 8 < e_tf, (name, e_own, e_tf)
    else:
        bf = lambda t: t.detach().bfloat16().float()
        r_y, r_gx, rwd, rxd = f64(bf(x), bf(w), bf(g))
        r_gw = torch.autograd.grad(F.conv2d(rxd, rwd, None, padding=pad), rwd, bf(g).double())[0]
        # the bias is added in fp32 to the sum of products of rounded operands: it is NOT rounded
        for name, own, ref in (("forward", y, r_y), ("data gradient", gx, r_gx), ("weight gradient", gw, r_gw)):
            e = _err(own, ref)
            print("%-15s %-22s bf16 operands: against their float64 convolution %.2e" % (name, case, e))
            assert e < 2e-6, (name, e)


def test_small_integers_are_exact_at_every_precision():
    """Integers up to 2^8 are one bf16 piece: one, three or six partial products give the same, exact, result (forward, both gradients)."""
    from hipops.functions import mfma_conv
    g = torch.Generator().manual_seed(11)
    x = torch.randint(-8, 9, (2, 64, 40, 96), generator=g).float().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randint(-4, 5, (64, 64, 3, 3), generator=g).float().cuda().requires_grad_(True)
    go = torch.randint(-3, 4, (2, 64, 40, 96), generator=g).float().cuda().contiguous(memory_format=torch.channels_last)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, padding=1)
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), go.double())
    before = torch.get_float32_matmul_precision()
    try:
        for precision in ("highest", "high", "medium"):
            torch.set_float32_matmul_precision(precision)
            y = mfma_conv(x, w, None, 1)
            gx, gw = torch.autograd.grad(y, (x, w), go)
            assert torch.equal(y.double(), yd.detach()) and torch.equal(gx.double(), gxd) and torch.equal(gw.double(), gwd), precision
    finally:
        torch.set_float32_matmul_precision(before)
