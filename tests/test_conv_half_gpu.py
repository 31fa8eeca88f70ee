"""dd_conv3x3_half (csrc/dd_conv_half.hip) against float64: the 3x3 stride-1 convolutions of the half-precision networks (BASELINE.json
config 5, "fp16 (CDNA4 MFMA conv)"; reference networks/resnet_encoder.py:95-135, networks/depth_decoder.py:10-55,
networks/motion_decoder.py:24-33,48-66 under autocast) -- half-precision operands, one MFMA per operand pair, fp32 accumulation, one
rounding on the way out.  The reference for both the kernel and the library is the float64 convolution of the SAME half-precision
operands (the weights rounded to the type, as the pack does and as autocast's cast does); the yardstick is the library's
half-precision result on the same inputs."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = {"fp16": torch.float16, "bf16": torch.bfloat16}


def _case(B, cin, cout, H, W, seed, dtype, bias=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (3.0 * cin ** 0.5)).cuda()
    b = torch.randn(cout, generator=g).cuda() if bias else None
    return x, w, b


def _err(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


# (B, cin, cout, H, W, pad): config 5's ResNet levels (scaled down in batch), a ragged image (partial tiles on both axes, partial
# last chunk: 40 and 72 channels), a pre-padded input (pad 0), one and three tiles of output channels, 16-channel layers
CASES = [(2, 64, 64, 72, 128, 1), (2, 128, 128, 36, 64, 1), (1, 72, 64, 48, 160, 1), (2, 40, 24, 19, 45, 1), (2, 64, 64, 50, 66, 0),
         (1, 16, 160, 17, 33, 1), (1, 32, 16, 9, 40, 1), (1, 256, 128, 18, 32, 1)]


@pytest.mark.parametrize("kind", ["fp16", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_forward_and_data_gradient_match_float64(case, kind):
    from hipops.functions import half_conv
    dtype = DTYPES[kind]
    B, cin, cout, H, W, pad = case
    x, w, b = _case(B, cin, cout, H, W, sum(case), dtype, bias=(cout % 3 != 0))
    wq = w.to(dtype)
    x.requires_grad_(True)
    y = half_conv(x, w, b, pad)
    assert y.dtype == dtype and y.stride(1) == 1
    bd = None if b is None else b.double()
    ref = F.conv2d(x.detach().double(), wq.double(), bd, padding=pad)
    assert y.shape == ref.shape
    lib = F.conv2d(x.detach(), wq, None if b is None else b.to(dtype), padding=pad)
    e_own, e_lib = _err(y, ref), _err(lib, ref)
    print("forward  %-26s %s own %.2e  library %.2e" % (case, kind, e_own, e_lib))
    ulp = 2.0 ** -11 if kind == "fp16" else 2.0 ** -8
    assert e_own <= max(1.5 * e_lib, 1.01 * ulp), (e_own, e_lib)          # one rounding of the largest output, or what the library leaves
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(1)).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    (gx,) = torch.autograd.grad(y, x, g)
    gref = torch.nn.grad.conv2d_input(x.shape, wq.double(), g.double(), padding=pad)
    glib = torch.nn.grad.conv2d_input(x.shape, wq, g, padding=pad)
    e_own, e_lib = _err(gx, gref), _err(glib, gref)
    print("data grad %-25s %s own %.2e  library %.2e" % (case, kind, e_own, e_lib))
    assert gx.shape == x.shape and gx.dtype == dtype
    assert e_own <= max(1.5 * e_lib, 1.01 * ulp), (e_own, e_lib)


def test_small_integers_are_exact_and_results_are_bit_reproducible():
    """Integer operands whose products and sums fit the types: every dot product is exact in fp32 accumulation and the result is an
    integer the half type holds -- any slip in the fragment layout (a channel octet, a tap, a lane half) shows as a wrong integer; and
    two launches on the same inputs agree bit for bit (no atomics, fixed order)."""
    from hipops.functions import half_conv
    for dtype in (torch.float16, torch.bfloat16):
        g = torch.Generator().manual_seed(5)
        B, cin, cout, H, W = 2, 48, 40, 21, 70
        x = torch.randint(-2, 3, (B, cin, H, W), generator=g).float().cuda().to(dtype).contiguous(memory_format=torch.channels_last)
        w = torch.randint(-1, 2, (cout, cin, 3, 3), generator=g).float().cuda()
        w = w * (torch.rand(cout, cin, 3, 3, generator=g).cuda() < 0.08).float()          # sparse: |y| stays below 256 (exact in bf16)
        y1 = half_conv(x, w, None, 1)
        y2 = half_conv(x, w, None, 1)
        ref = F.conv2d(x.double(), w.double(), None, padding=1)
        assert float(ref.abs().max()) <= 256
        assert torch.equal(y1, y2)
        assert torch.equal(y1.double(), ref), float((y1.double() - ref).abs().max())


@pytest.mark.parametrize("kind", ["fp16", "bf16"])
def test_layer_under_autocast_takes_the_kernel_and_matches_the_library_layer(kind, monkeypatch):
    """layers.Conv2d under autocast on a half-precision channels-last input: the hook routes the layer through dd_conv3x3_half (counted),
    output, data gradient, weight gradient (library kernel, promoted to fp32) and bias gradient agree with the stock autocast layer."""
    from hipops import functions as Fn
    from networks.layers import Conv2d
    dtype = DTYPES[kind]
    torch.manual_seed(3)
    conv = Conv2d(64, 48, 3, padding=1).cuda()
    x0 = torch.randn(4, 64, 40, 136, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    g = torch.randn(4, 48, 40, 136, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    outs = {}
    for own in (True, False):
        monkeypatch.setenv("DD_HALF_MFMA_CONV", "1" if own else "0")
        x = x0.clone().requires_grad_(True)
        conv.zero_grad(set_to_none=True)
        before = Fn.half_conv_calls()
        with torch.autocast("cuda", dtype=dtype):
            y = conv(x)
        assert (Fn.half_conv_calls() == before + 1) == own
        y.backward(g)
        assert conv.weight.grad.dtype == torch.float32 and conv.bias.grad.dtype == torch.float32 and x.grad.dtype == dtype
        outs[own] = (y.detach().double(), x.grad.double(), conv.weight.grad.double(), conv.bias.grad.double())
    # both sides round their results to the half type once (and the library's weight gradient adds its split partial sums with atomics:
    # the order differs from run to run): a few units of the type's last place relative to the largest element -- measured 7e-4 .. 2.8e-3
    # (fp16) and 1.6e-3 .. 5.9e-3 (bf16)
    tol = 4 * 2.0 ** -10 if kind == "fp16" else 4 * 2.0 ** -7
    for name, a, b in zip(("y", "g_x", "g_w", "g_b"), outs[True], outs[False]):
        err = float((a - b).abs().max() / b.abs().max())
        print("%s %s own vs stock autocast layer: %.2e" % (kind, name, err))
        assert err <= tol, (name, err)


WG_CASES = [(2, 64, 64, 72, 128, 1), (2, 128, 64, 36, 64, 1), (1, 72, 64, 48, 160, 1), (2, 40, 32, 19, 45, 1), (2, 64, 64, 50, 66, 0), (1, 32, 160, 17, 33, 1),
            (3, 256, 128, 18, 32, 1)]


@pytest.mark.parametrize("kind", ["fp16", "bf16"])
@pytest.mark.parametrize("case", WG_CASES, ids=lambda c: "x".join(map(str, c)))
def test_weight_gradient_matches_float64(case, kind):
    """dd_conv3x3_half_bwd_weight: half x half products with fp32 accumulation and an fp32 result, against the float64 weight gradient of the
    same half-precision operands; the yardstick is the library's half-precision weight gradient (rounded to the half type)."""
    from hipops import lib as L
    from hipops.functions import _p, _ws, _ws_bytes, _dense_nhwc, DTYPE_CODE
    dtype = DTYPES[kind]
    B, cin, cout, H, W, pad = case
    x, w, _ = _case(B, cin, cout, H, W, sum(case) + 7, dtype, bias=False)
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    g = torch.randn(B, cout, Ho, Wo, generator=torch.Generator().manual_seed(2)).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    lib = L.load()
    flat = torch.empty(cout * 9 * cin, dtype=torch.float32, device="cuda")
    nbytes = _ws_bytes("dd_conv3x3_half_wgrad_workspace_bytes", B, Ho, Wo, cin, cout)
    ws = _ws(nbytes, x.device)
    xd, gd = _dense_nhwc(x), _dense_nhwc(g)
    for _ in range(2):
        L.check(lib.dd_conv3x3_half_bwd_weight(_p(xd), _p(gd), B, H, W, cin, cout, pad, DTYPE_CODE[dtype], _p(flat), _p(ws), nbytes, L.current_stream()), "bwd_weight")
        first = flat.clone() if _ == 0 else first
    assert torch.equal(first, flat)                       # bit-reproducible
    gw = flat.view(cout, 3, 3, cin).permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, g.double(), padding=pad)
    wl = torch.empty_like(w, dtype=dtype)
    _, glib, _ = torch.ops.aten.convolution_backward(g, x, wl, None, (1, 1), (pad, pad), (1, 1), False, [0, 0], 1, (False, True, False))
    e_own, e_lib = _err(gw, ref), _err(glib, ref)
    print("weight grad %-26s %s own %.2e  library %.2e" % (case, kind, e_own, e_lib))
    assert e_own <= max(1.0 * e_lib, 3e-6), (e_own, e_lib)          # fp32 out: better than the library's half-rounded result
