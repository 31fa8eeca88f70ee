"""dd_conv_small (csrc/dd_conv_small.hip through hipops.functions.SmallConvFn): the motion decoders' convolutions on 9-12 channels at full
resolution (reference networks/motion_decoder.py:24-33,57-66) -- forward, data gradient, weight gradient (matrix pipe) and bias
gradient against torch's own conv2d evaluated in float64 on the CPU, on ragged sizes (tiles that hang over the image), every
instantiated channel combination, weights in either layout; bit-reproducible; and the decoder itself with and without the hook."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

COMBOS = [(3, 12, 9), (3, 10, 9), (3, 9, 9), (3, 16, 12), (3, 13, 12), (3, 12, 12), (1, 9, 3), (1, 9, 1), (1, 12, 3), (1, 12, 1)]


def _case(ks, cin, cout, B, H, W, seed, bias=True, w_nhwc=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, ks, ks, generator=g) * 0.2
    b = torch.randn(cout, generator=g) if bias else None
    go = torch.randn(B, cout, H, W, generator=g)
    return x, w, b, go


def _reference(x, w, b, go):
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    bd = b.double().requires_grad_() if b is not None else None
    y = F.conv2d(xd, wd, bd, 1, w.shape[-1] // 2)
    y.backward(go.double())
    return y.detach(), xd.grad, wd.grad, (bd.grad if bd is not None else None)


@pytest.mark.parametrize("combo", COMBOS)
@pytest.mark.parametrize("shape", [(2, 37, 70), (1, 8, 32), (3, 5, 131)])
def test_small_conv_against_float64(combo, shape):
    from hipops.functions import SmallConvFn
    from hipops import lib as L
    ks, cin, cout = combo
    assert L.load().dd_conv_small_supported(ks, cin, cout) == 1
    B, H, W = shape
    x, w, b, go = _case(ks, cin, cout, B, H, W, seed=ks * 1000 + cin * 17 + cout)
    y64, gx64, gw64, gb64 = _reference(x, w, b, go)
    xc = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    wc = w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    bc = b.cuda().requires_grad_()
    y = SmallConvFn.apply(xc, wc, bc)
    y.backward(go.cuda())
    torch.cuda.synchronize()

    def close(got, want, rel):
        want = want.float()
        err = float((got.detach().cpu() - want).abs().max())
        assert err <= rel * max(float(want.abs().max()), 1e-6), (err, float(want.abs().max()))
    close(y, y64, 2e-6)
    close(xc.grad, gx64, 2e-6)
    close(wc.grad, gw64, 2e-5)              # sums over B*H*W pixels on the matrix pipe, folded over the workgroups
    close(bc.grad, gb64, 2e-5)
    assert wc.grad.shape == w.shape and y.shape == (B, cout, H, W)


def test_small_conv_layouts_and_repeatability():
    """NCHW input and gradient (converted on the way in), a contiguous weight (addressed through its strides), no bias; twice the same
    bits (no atomics anywhere)."""
    from hipops.functions import SmallConvFn
    ks, cin, cout = 3, 12, 9
    x, w, _, go = _case(ks, cin, cout, 2, 50, 100, seed=5, bias=False)
    y64, gx64, gw64, _ = _reference(x, w, None, go)
    outs = []
    for _ in range(2):
        xc, wc = x.cuda().requires_grad_(), w.cuda().requires_grad_()          # NCHW, contiguous weight
        y = SmallConvFn.apply(xc, wc, None)
        y.backward(go.cuda())
        outs.append((y.detach().clone(), xc.grad.clone(), wc.grad.clone()))
    torch.cuda.synchronize()
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    y, gx, gw = outs[0]
    assert float((y.cpu() - y64.float()).abs().max()) <= 2e-6 * float(y64.abs().max())
    assert float((gx.cpu() - gx64.float()).abs().max()) <= 2e-6 * float(gx64.abs().max())
    assert float((gw.cpu() - gw64.float()).abs().max()) <= 2e-5 * float(gw64.abs().max())


def test_full_resolution_case_against_the_library():
    """The shape of the headline configuration (12 x 192 x 640, 12 -> 9 channels): against MIOpen's convolution on the same tensors."""
    from hipops.functions import SmallConvFn, small_conv_ok
    x, w, b, go = _case(3, 12, 9, 12, 192, 640, seed=9)
    xc = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    wc = w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    bc = b.cuda().requires_grad_()
    assert small_conv_ok(xc, wc, (1, 1), (1, 1), (1, 1), 1)
    y = SmallConvFn.apply(xc, wc, bc)
    y.backward(go.cuda())
    mine = (y.detach(), xc.grad.clone(), wc.grad.clone(), bc.grad.clone())
    xc.grad = wc.grad = bc.grad = None
    y2 = F.conv2d(xc, wc, bc, 1, 1)
    y2.backward(go.cuda())
    theirs = (y2.detach(), xc.grad, wc.grad, bc.grad)
    for name, a, t, rel in zip(("y", "gx", "gw", "gb"), mine, theirs, (1e-5, 1e-5, 2e-4, 2e-4)):
        err, ref = float((a - t).abs().max()), float(t.abs().max())
        assert err <= rel * ref, (name, err, ref)


def test_motion_decoder_with_and_without_the_hook():
    """networks.MotionDecoder at the bench shape, forward and parameter gradients, hook on against DD_STOCK_SMALL_CONV=1."""
    from networks.motion_decoder import MotionDecoder
    torch.manual_seed(0)
    dec = MotionDecoder([64, 64, 128, 256, 512], scales=[0, 1, 2], num_input_images=3, inp_disp=False, out_dim=3).cuda().to(memory_format=torch.channels_last)
    B, H, W = 4, 192, 640
    feats = [torch.randn(B, 9, H, W, device="cuda")] + [torch.randn(B, c, H >> (i + 1), W >> (i + 1), device="cuda").contiguous(memory_format=torch.channels_last)
                                                         for i, c in enumerate([64, 64, 128, 256, 512])]
    ego = torch.randn(B, 6, 1, 1, device="cuda") * 0.01
    res = {}
    for stock in ("1", "0"):
        os.environ["DD_STOCK_SMALL_CONV"] = stock
        try:
            dec.zero_grad(set_to_none=True)
            out = dec(feats, ego)
            loss = sum((v * v).mean() for v in out.values())
            loss.backward()
            res[stock] = ([v.detach().clone() for v in out.values()], {n: p.grad.clone() for n, p in dec.named_parameters() if p.grad is not None})
        finally:
            os.environ.pop("DD_STOCK_SMALL_CONV", None)
    for a, b in zip(res["0"][0], res["1"][0]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    for n, gs in res["1"][1].items():
        err, ref = float((res["0"][1][n] - gs).abs().max()), float(gs.abs().max())
        assert err <= 5e-4 * max(ref, 1e-8), (n, err, ref)


@pytest.mark.parametrize("case", [(2, 32, 20, 40), (3, 64, 13, 37), (1, 32, 3, 3), (2, 32, 11, 70), (12, 32, 98, 322)])
def test_disparity_head_against_float64(case):
    """dd_conv_head (csrc/dd_conv_head.hip through hipops.functions.HeadConvFn): the disparity heads -- 3x3 to one channel on an input
    that carries its own padding (reference networks/depth_decoder.py:49-51,95-97) -- forward, weight + bias gradient, data gradient."""
    from hipops.functions import HeadConvFn, head_conv_ok
    B, Cc, Hp, Wp = case
    g = torch.Generator().manual_seed(B * 100 + Cc + Hp)
    x = torch.randn(B, Cc, Hp, Wp, generator=g)
    w = torch.randn(1, Cc, 3, 3, generator=g) * 0.1
    b = torch.randn(1, generator=g)
    go = torch.randn(B, 1, Hp - 2, Wp - 2, generator=g)
    xd, wd, bd = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    F.conv2d(xd, wd, bd).backward(go.double())
    y64 = F.conv2d(xd, wd, bd).detach()
    outs = []
    for _ in range(2):
        xc = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        wc = w.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        bc = b.cuda().requires_grad_()
        assert head_conv_ok(xc, wc, (1, 1), (0, 0), (1, 1), 1)
        y = HeadConvFn.apply(xc, wc, bc)
        y.backward(go.cuda())
        outs.append((y.detach().clone(), xc.grad.clone(), wc.grad.clone(), bc.grad.clone()))
    torch.cuda.synchronize()
    for a, c in zip(*outs):
        assert torch.equal(a, c)                       # no atomics: the same bits twice
    for got, want, rel in zip(outs[0], (y64, xd.grad, wd.grad, bd.grad), (2e-6, 2e-6, 2e-5, 2e-5)):
        want = want.float()
        err = float((got.cpu() - want).abs().max())
        assert err <= rel * max(float(want.abs().max()), 1e-6), (err, float(want.abs().max()))
    assert outs[0][2].shape == w.shape


@pytest.mark.parametrize("case", [(2, 64, 9, 21, 3), (2, 64, 9, 21, 1), (3, 128, 5, 7, 3), (2, 256, 4, 6, 1), (2, 512, 3, 5, 3), (1, 512, 1, 1, 1), (12, 64, 96, 320, 3)])
def test_reduction_against_float64(case):
    """dd_redu (csrc/dd_redu.hip through hipops.functions.ReduFn): conv1x1(cat(a, b)) + bias of the motion decoders (reference
    networks/motion_decoder.py:33,66) -- forward, both data gradients, weight and bias gradient against float64; twice the same bits."""
    from hipops.functions import ReduFn, redu_ok
    B, Cc, H, W, cout = case
    gen = torch.Generator().manual_seed(Cc + H * 7 + cout)
    a, b = torch.randn(B, Cc, H, W, generator=gen), torch.randn(B, Cc, H, W, generator=gen)
    conv = torch.nn.Conv2d(2 * Cc, cout, 1)
    go = torch.randn(B, cout, H, W, generator=gen)
    ad, bd = a.double().requires_grad_(), b.double().requires_grad_()
    cd = torch.nn.Conv2d(2 * Cc, cout, 1).double()
    cd.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    y64 = cd(torch.cat((ad, bd), 1))
    y64.backward(go.double())
    conv = conv.cuda().to(memory_format=torch.channels_last)
    outs = []
    for _ in range(2):
        conv.zero_grad(set_to_none=True)
        ac = a.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        bc = b.cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
        if Cc > 1 and H * W > 1:
            assert redu_ok(ac, bc, conv)
        y = ReduFn.apply(ac, bc, conv.weight, conv.bias)
        y.backward(go.cuda())
        outs.append((y.detach().clone(), ac.grad.clone(), bc.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone()))
    torch.cuda.synchronize()
    for p, q in zip(*outs):
        assert torch.equal(p, q)
    wants = (y64.detach(), ad.grad, bd.grad, cd.weight.grad, cd.bias.grad)
    for got, want, rel in zip(outs[0], wants, (3e-6, 2e-6, 2e-6, 2e-5, 2e-5)):
        want = want.float()
        assert got.shape == want.shape
        err = float((got.cpu() - want).abs().max())
        assert err <= rel * max(float(want.abs().max()), 1e-6), (err, float(want.abs().max()))
