"""tools.py operators (HIP, through the C ABI) against the reference's operator goldens (tests/golden/ops.npz) and,
for gradients, against the oracle's autograd.  GPU only."""
import os

import numpy as np
import pytest
import torch

import oracle.ref_loss as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "ops.npz"))


def cu(a, grad=False):
    t = torch.from_numpy(np.asarray(a)).cuda()
    return t.requires_grad_() if grad else t


def close(got, want, rtol, atol, name):
    got = got.detach().cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=name)


def test_forward_matches_reference(z):
    import tools
    from networks.layers import transformation_from_parameters
    from utils import interp
    B, _, h, w = z["depth"].shape
    bp = tools.BackprojectDepth(B + 1, h, w).cuda()
    pts = bp(cu(z["depth"]), cu(z["inv_K"]))
    close(pts, z["points"], 1e-5, 1e-5, "backproject")
    pj = tools.Project3D(B + 1, h, w)
    pix, ego = pj(cu(z["points"]), cu(z["K"]), cu(z["T_inv"]))
    close(pix, z["pix_T"], 1e-4, 2e-5, "project pix")
    close(ego, z["ego_T"], 1e-4, 1e-5, "project ego")
    pix, ego = pj(cu(z["points"]), cu(z["K"]), None)
    close(pix, z["pix_N"], 1e-4, 2e-5, "project pix (T=None)")
    assert float(ego.abs().max()) == 0.0
    close(tools.SSIM()(cu(z["x"]), cu(z["y"])), z["ssim"], 1e-4, 2e-6, "ssim")
    sd, dp = tools.disp_to_depth(cu(z["disp"]), 0.1, 100.0)
    close(sd, z["scaled_disp"], 1e-6, 0, "scaled disp")
    close(dp, z["depth_from_disp"], 1e-6, 0, "depth")
    close(tools.depth_to_disp(dp, 0.1, 100.0), z["disp_roundtrip"], 1e-4, 1e-6, "disp roundtrip")
    assert abs(float(tools.compute_smooth_loss(cu(z["smooth_inp"]), cu(z["x"]))) - float(z["smooth_img"])) < 2e-6
    assert abs(float(tools.compute_smooth_loss(cu(z["smooth_inp"]), None)) - float(z["smooth_none"])) < 2e-6
    close(transformation_from_parameters(cu(z["axisangle"]), cu(z["translation"]), invert=True), z["T_inv"], 1e-5, 1e-6, "T inv")
    close(transformation_from_parameters(cu(z["axisangle"]), cu(z["translation"]), invert=False), z["T_fwd"], 1e-5, 1e-6, "T fwd")
    close(interp(cu(z["disp"]), (h * 4, w * 4)), z["interp_up"], 1e-5, 1e-6, "interp")
    gp = tools.GroundPlane(num_points_per_it=5, max_it=100, tol=0.005, g_prior=0.4)
    dist, param = gp(cu(z["ground_points"]), rand_idx=z["ground_rand_idx"])
    close(param, z["ground_param"], 2e-3, 2e-4, "ground plane")
    close(dist, z["ground_dist"], 2e-3, 2e-3, "ground distance")


def test_backward_matches_oracle_autograd(z):
    import tools
    from networks.layers import transformation_from_parameters
    g = torch.Generator().manual_seed(3)
    B, _, h, w = z["depth"].shape

    def both(fn_hip, fn_ref, inputs, name, rtol=2e-3):
        gpu_in = [cu(a, True) for a in inputs]
        cpu_in = [torch.from_numpy(np.asarray(a)).clone().requires_grad_() for a in inputs]
        out_g, out_c = fn_hip(*gpu_in), fn_ref(*cpu_in)
        out_g = out_g if isinstance(out_g, (tuple, list)) else [out_g]
        out_c = out_c if isinstance(out_c, (tuple, list)) else [out_c]
        total_g = total_c = 0
        for a, b in zip(out_g, out_c):
            wgt = torch.randn(b.shape, generator=g)
            total_g = total_g + (a * wgt.cuda()).sum()
            total_c = total_c + (b * wgt).sum()
        total_g.backward()
        total_c.backward()
        for i, (a, b) in enumerate(zip(gpu_in, cpu_in)):
            if b.grad is None:
                continue
            err = (a.grad.cpu() - b.grad).norm() / (b.grad.norm() + 1e-20)
            assert float(err) < rtol, "%s input %d: rel grad err %.3e" % (name, i, float(err))

    K, invK = torch.from_numpy(z["K"]), torch.from_numpy(z["inv_K"])
    bp = tools.BackprojectDepth(B, h, w).cuda()
    both(lambda d: bp(d, invK.cuda()), lambda d: orc.backproject(d, invK), [z["depth"]], "backproject")
    pj = tools.Project3D(B, h, w)
    both(lambda p, T: pj(p, K.cuda(), T), lambda p, T: orc.project(p, K, T, h, w), [z["points"], z["T_inv"]], "project3d")
    both(lambda p: pj(p, K.cuda(), None), lambda p: orc.project(p, K, None, h, w), [z["points"]], "project3d T=None")
    both(lambda x, y: tools.SSIM()(x, y), orc.ssim_map, [z["x"], z["y"]], "ssim")
    both(lambda d: tools.disp_to_depth(d, 0.1, 100.0), lambda d: orc.disp_to_depth(d, 0.1, 100.0), [z["disp"] + 0.05], "disp_to_depth")
    img = torch.from_numpy(z["x"])
    both(lambda a: tools.compute_smooth_loss(a, img.cuda()), lambda a: orc.smooth_loss(a, img), [z["smooth_inp"]], "smooth")
    both(lambda a, t: transformation_from_parameters(a, t, invert=True), lambda a, t: orc.pose_matrix(a, t, invert=True),
         [z["axisangle"], z["translation"]], "pose invert")
    both(lambda a, t: transformation_from_parameters(a, t, invert=False), lambda a, t: orc.pose_matrix(a, t, invert=False),
         [z["axisangle"], z["translation"]], "pose")


@pytest.mark.parametrize("shape", [(12, 9, 192, 640), (2, 1, 7, 5), (3, 256, 6, 20), (4, 33, 16, 16)])
def test_conv_bias_gradient_channel_sum(shape):
    """networks.layers.Conv2d (bias gradient through dd_channel_sum_nhwc) == nn.Conv2d, channels-last and contiguous."""
    import torch.nn as nn
    from networks.layers import Conv2d
    B, C, H, W = shape
    torch.manual_seed(0)
    ref = nn.Conv2d(5, C, 3, padding=1).cuda()
    mine = Conv2d(5, C, 3, padding=1).cuda()
    mine.load_state_dict(ref.state_dict())
    for fmt in (torch.channels_last, torch.contiguous_format):
        x = torch.randn(B, 5, H, W, device="cuda").to(memory_format=fmt)
        xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
        g = torch.randn(B, C, H, W, device="cuda").to(memory_format=fmt)
        ref.zero_grad(); mine.zero_grad()
        ya = ref.to(memory_format=fmt)(xa); ya.backward(g)
        yb = mine.to(memory_format=fmt)(xb); yb.backward(g)
        assert torch.allclose(ya, yb, rtol=1e-5, atol=1e-5)
        scale = ref.bias.grad.abs().max().item() + 1e-6
        assert (ref.bias.grad - mine.bias.grad).abs().max().item() < 2e-4 * scale + 1e-4 * (B * H * W) ** 0.5 * 1e-3
        assert torch.allclose(ref.weight.grad, mine.weight.grad, rtol=1e-3, atol=1e-3 * ref.weight.grad.abs().max().item())
        assert torch.allclose(xa.grad, xb.grad, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("shape", [(12, 32, 96, 320), (2, 3, 4, 5), (3, 17, 12, 40)])
def test_reflection_pad_nhwc(shape):
    import torch.nn.functional as F
    from hipops.functions import reflect_pad1
    x = torch.randn(*shape, device="cuda").to(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = F.pad(xa, (1, 1, 1, 1), mode="reflect"), reflect_pad1(xb)
    assert yb.is_contiguous(memory_format=torch.channels_last) and torch.equal(ya, yb)
    g = torch.randn_like(ya)
    ya.backward(g); yb.backward(g)
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,c2,mode,elu", [
    ((12, 64, 12, 40), 80, "bilinear", True), ((12, 40, 24, 80), 48, "bilinear", True), ((12, 24, 48, 160), 0, "bilinear", True),
    ((12, 24, 96, 320), 0, None, True), ((3, 128, 12, 40), 0, None, False), ((2, 256, 6, 20), 256, "nearest", True),
    ((2, 16, 96, 320), 0, "nearest", True), ((1, 8, 2, 2), 4, "bilinear", True), ((2, 4, 3, 5), 8, "nearest", False), ((1, 12, 4, 4), 0, None, True)])
def test_decoder_glue_is_the_operator_sequence(shape, c2, mode, elu, dtype):
    """dd_up_cat_pad_t / _bwd_t (ELU + x2 up-sampling + skip concatenation + reflection padding in one pass) against the
    reference's operators (networks/layers.py:84-121, networks/depth_decoder.py:40-53,:98-113) in fp32 on the same values."""
    import torch.nn.functional as F
    from hipops.functions import UpCatPadFn, up_cat_pad_ok
    B, C1, h, w = shape
    g = torch.Generator().manual_seed(sum(shape) + c2)
    x = (torch.randn(shape, generator=g) * 1.5).to(dtype).cuda().contiguous(memory_format=torch.channels_last).requires_grad_()
    H, W = (h, w) if mode is None else (2 * h, 2 * w)
    skip = (torch.randn((B, c2, H, W), generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last).requires_grad_() if c2 else None)
    code = {"nearest": 0, "bilinear": 1, None: 2}[mode]
    assert up_cat_pad_ok(x, skip, code)
    out = UpCatPadFn.apply(x, skip, code, elu)
    assert out.shape == (B, C1 + c2, H + 2, W + 2) and out.is_contiguous(memory_format=torch.channels_last) and out.dtype == dtype
    go = torch.randn(out.shape, generator=g).to(dtype).cuda().contiguous(memory_format=torch.channels_last)
    out.backward(go)
    xr = x.detach().float().requires_grad_()
    sr = skip.detach().float().requires_grad_() if c2 else None
    y = F.elu(xr) if elu else xr
    if mode is not None:
        y = F.interpolate(y, scale_factor=2, mode=mode)
    if c2:
        y = torch.cat((y, sr), 1)
    ref = F.pad(y, (1, 1, 1, 1), mode="reflect")
    ref.backward(go.float())
    eps = {torch.float32: 2e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]

    def near(got, want, what):
        err = float((got.float() - want).abs().max())
        assert err <= eps * max(1.0, float(want.abs().max())), (what, err)
    near(out, ref, "forward")
    near(x.grad, xr.grad, "d x")
    if c2:
        near(skip.grad, sr.grad, "d skip")


@pytest.mark.parametrize("shape,dil", [((12, 64, 48, 160), 1), ((3, 128, 24, 80), 2), ((2, 224, 12, 40), 6), ((1, 8, 5, 7), 3), ((2, 16, 9, 3), 1)])
def test_depthwise_dilated_conv_nhwc(shape, dil):
    """CDilated (reference networks/depth_encoder.py:168-181) with groups == channels: forward, data and weight gradients
    against ATen's convolution in float64."""
    import torch.nn.functional as F
    from hipops.functions import DepthwiseConv3x3NHWCFn, depthwise_conv3x3
    g0 = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(*shape, device="cuda", generator=g0).to(memory_format=torch.channels_last)
    w = torch.randn(shape[1], 1, 3, 3, device="cuda", generator=g0) * 0.3
    xa, wa = x.double().requires_grad_(), w.double().requires_grad_()
    xb, wb = x.clone().requires_grad_(), w.clone().requires_grad_()
    ya = F.conv2d(xa, wa, None, 1, dil, dil, shape[1])
    yb = depthwise_conv3x3(xb, wb, dil) if shape[0] > 1 else DepthwiseConv3x3NHWCFn.apply(xb, wb, dil)
    assert yb.grad_fn.name().startswith("DepthwiseConv3x3NHWCFn") and yb.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(ya.float(), yb, rtol=1e-5, atol=1e-5)
    g = torch.randn(*shape, device="cuda", generator=g0).to(memory_format=torch.channels_last)
    ya.backward(g.double()); yb.backward(g)
    assert torch.allclose(xa.grad.float(), xb.grad, rtol=1e-5, atol=1e-5)
    scale = wa.grad.abs().max().item()
    assert (wa.grad.float() - wb.grad).abs().max().item() <= 2e-5 * max(scale, 1.0)
    yb2 = depthwise_conv3x3(x.clone().requires_grad_(), wb.detach().clone().requires_grad_(), dil) if shape[0] > 1 else None
    if yb2 is not None:                                   # fixed-order weight-gradient sums: bit-reproducible
        w2 = wb.detach().clone().requires_grad_()
        y2 = depthwise_conv3x3(x, w2, dil); y2.backward(g)
        assert torch.equal(w2.grad, wb.grad)


@pytest.mark.parametrize("shape,cout", [((12, 48, 160, 64), 384), ((3, 24, 80, 768), 128), ((2, 12, 40, 224), 1344), ((1, 3, 5, 8), 300)])
def test_pointwise_linear_gradients(shape, cout):
    """LiteMono's channels-last Linears (reference networks/depth_encoder.py:200-203): forward and all three gradients of the
    HIP/MIOpen split against ATen's Linear in float64."""
    import torch.nn as nn
    from hipops.functions import pointwise_linear
    g0 = torch.Generator(device="cuda").manual_seed(11)
    layer = nn.Linear(shape[-1], cout).cuda()
    x = torch.randn(*shape, device="cuda", generator=g0)
    ref = nn.Linear(shape[-1], cout).cuda().double()
    ref.load_state_dict({k: v.double() for k, v in layer.state_dict().items()})
    xa, xb = x.double().requires_grad_(), x.clone().requires_grad_()
    ya, yb = ref(xa), pointwise_linear(xb, layer)
    assert yb.grad_fn.name().startswith("PointwiseLinearFn")
    assert torch.allclose(ya.float(), yb, rtol=1e-4, atol=1e-4)
    g = torch.randn(*shape[:-1], cout, device="cuda", generator=g0)
    ya.backward(g.double()); yb.backward(g)
    assert torch.allclose(xa.grad.float(), xb.grad, rtol=1e-4, atol=1e-4)
    for a, b in ((ref.weight.grad, layer.weight.grad), (ref.bias.grad, layer.bias.grad)):
        assert (a.float() - b).abs().max().item() <= 1e-4 * max(a.abs().max().item(), 1.0)


@pytest.mark.parametrize("shape,act,res", [((12, 64, 96, 320), "relu", False), ((12, 64, 48, 160), "relu", True), ((4, 512, 6, 20), "relu", True),
                                           ((12, 224, 12, 40), None, False), ((3, 64, 48, 160), "gelu", False), ((2, 128, 24, 80), None, True),
                                           ((1, 8, 3, 5), "relu", True)])
def test_batchnorm_act_fused(shape, act, res):
    """layers.BatchNorm2d(x, act, residual) on the HIP path against nn.BatchNorm2d + activation in float64: output, running
    statistics, and the gradients of x, residual, weight, bias."""
    import torch.nn as nn
    import torch.nn.functional as F
    from networks.layers import BatchNorm2d
    g0 = torch.Generator(device="cuda").manual_seed(13)
    Cc = shape[1]
    x = (torch.randn(*shape, device="cuda", generator=g0) * 1.7 + 0.4).to(memory_format=torch.channels_last)
    r = torch.randn(*shape, device="cuda", generator=g0).to(memory_format=torch.channels_last) if res else None
    bn = BatchNorm2d(Cc).cuda().train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(Cc, device="cuda", generator=g0) + 0.5); bn.bias.copy_(torch.randn(Cc, device="cuda", generator=g0) * 0.2)
    ref = nn.BatchNorm2d(Cc).cuda().double().train()
    ref.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn.state_dict().items()})
    xa, xb = x.double().requires_grad_(), x.clone().requires_grad_()
    ra, rb = (r.double().requires_grad_(), r.clone().requires_grad_()) if res else (None, None)
    ya = ref(xa)
    if res:
        ya = ya + ra
    ya = F.relu(ya) if act == "relu" else (F.gelu(ya) if act == "gelu" else ya)
    yb = bn(xb, act=act, residual=rb)
    assert yb.grad_fn.name().startswith("BatchNormActFn") and yb.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(ya.float(), yb, rtol=2e-5, atol=2e-5)
    assert torch.allclose(ref.running_mean.float(), bn.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(ref.running_var.float(), bn.running_var, rtol=1e-5, atol=1e-6)
    g = torch.randn(*shape, device="cuda", generator=g0).to(memory_format=torch.channels_last)
    ya.backward(g.double()); yb.backward(g)
    assert torch.allclose(xa.grad.float(), xb.grad, rtol=1e-4, atol=2e-5)
    if res:
        assert torch.allclose(ra.grad.float(), rb.grad, rtol=1e-5, atol=1e-6)
    for a, b in ((ref.weight.grad, bn.weight.grad), (ref.bias.grad, bn.bias.grad)):
        assert (a.float() - b).abs().max().item() <= 2e-5 * max(a.abs().max().item(), 1.0)
    assert int(bn.state_dict()["num_batches_tracked"]) == 1


@pytest.mark.parametrize("shape", [(12, 48, 160, 64), (12, 7680, 64), (3, 24, 80, 128), (12, 480, 224), (2, 5, 8), (1, 3, 7, 256)])
def test_layer_norm_channels_last(shape):
    """LiteMono's LayerNorm(data_format='channels_last') (reference networks/depth_encoder.py:101-128): HIP forward and the three
    gradients against F.layer_norm in float64."""
    import torch.nn.functional as F
    from hipops.functions import layer_norm_last
    g0 = torch.Generator(device="cuda").manual_seed(17)
    Cc = shape[-1]
    x = torch.randn(*shape, device="cuda", generator=g0) * 2.0 + 0.7
    w = torch.rand(Cc, device="cuda", generator=g0) + 0.5
    b = torch.randn(Cc, device="cuda", generator=g0) * 0.3
    xa, wa, ba = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    xb, wb, bb = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ya = F.layer_norm(xa, (Cc,), wa, ba, 1e-6)
    yb = layer_norm_last(xb, wb, bb, 1e-6)
    assert yb.grad_fn.name().startswith("LayerNormFn")
    assert torch.allclose(ya.float(), yb, rtol=1e-5, atol=1e-5)
    g = torch.randn(*shape, device="cuda", generator=g0)
    ya.backward(g.double()); yb.backward(g)
    assert torch.allclose(xa.grad.float(), xb.grad, rtol=1e-4, atol=2e-5)
    for a, c in ((wa.grad, wb.grad), (ba.grad, bb.grad)):
        assert (a.float() - c).abs().max().item() <= 2e-5 * max(a.abs().max().item(), 1.0)
    w2, b2, x2 = w.clone().requires_grad_(), b.clone().requires_grad_(), x.clone().requires_grad_()
    layer_norm_last(x2, w2, b2, 1e-6).backward(g)
    assert torch.equal(w2.grad, wb.grad) and torch.equal(b2.grad, bb.grad) and torch.equal(x2.grad, xb.grad)


@pytest.mark.parametrize("shape,pad", [((12, 32, 98, 322), 0), ((3, 64, 50, 162), 0), ((2, 112, 26, 82), 0), ((2, 16, 9, 7), 1)])
def test_disparity_head_data_gradient(shape, pad):
    """Conv3x3(C, 1) of the disparity heads (reference networks/depth_decoder.py:49-51): data gradient through the HIP
    outer-product kernel inside ConvBiasFn, against ATen in float64."""
    import torch.nn.functional as F
    from hipops.functions import ConvBiasFn
    g0 = torch.Generator(device="cuda").manual_seed(29)
    x = torch.randn(*shape, device="cuda", generator=g0).to(memory_format=torch.channels_last)
    w = torch.randn(1, shape[1], 3, 3, device="cuda", generator=g0) * 0.2
    b = torch.randn(1, device="cuda", generator=g0)
    xa, wa, ba = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    xb, wb, bb = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ya = F.conv2d(xa, wa, ba, 1, pad)
    yb = ConvBiasFn.apply(xb, wb, bb, (1, 1), (pad, pad), (1, 1), 1)
    assert torch.allclose(ya.float(), yb, rtol=1e-4, atol=1e-4)
    g = torch.randn_like(yb)
    ya.backward(g.double()); yb.backward(g)
    assert torch.allclose(xa.grad.float(), xb.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(wa.grad.float(), wb.grad, rtol=1e-3, atol=1e-3) and torch.allclose(ba.grad.float(), bb.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("shape,drop", [((12, 48, 160, 64), True), ((3, 12, 40, 224), False), ((2, 5, 7, 8), True)])
def test_layer_scale_residual(shape, drop):
    """res + drop * gamma * y (reference networks/depth_encoder.py:219-226): gradients of res, y, gamma against torch in float64."""
    from hipops.functions import layer_scale_residual
    g0 = torch.Generator(device="cuda").manual_seed(31)
    B, H, W, Cc = shape
    res, y = torch.randn(*shape, device="cuda", generator=g0), torch.randn(*shape, device="cuda", generator=g0)
    gamma = torch.randn(Cc, device="cuda", generator=g0)
    d = (torch.rand(B, 1, 1, 1, device="cuda", generator=g0) > 0.3).float() / 0.7 if drop else None
    ra, ya, ga = res.double().requires_grad_(), y.double().requires_grad_(), gamma.double().requires_grad_()
    rb, yb, gb = res.clone().requires_grad_(), y.clone().requires_grad_(), gamma.clone().requires_grad_()
    oa = ra + ya * (ga if d is None else ga * d.double())
    ob = layer_scale_residual(rb, yb, gb, d)
    assert ob.grad_fn.name().startswith("LayerScaleResidualFn")
    assert torch.allclose(oa.float(), ob, rtol=1e-5, atol=1e-5)
    g = torch.randn(*shape, device="cuda", generator=g0)
    oa.backward(g.double()); ob.backward(g)
    assert torch.allclose(ra.grad.float(), rb.grad) and torch.allclose(ya.grad.float(), yb.grad, rtol=1e-5, atol=1e-6)
    assert (ga.grad.float() - gb.grad).abs().max().item() <= 2e-5 * max(ga.grad.abs().max().item(), 1.0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_half_precision_network_hooks(dtype):
    """BASELINE.json config 5 (fp16 / bf16 convs): the channels-last BatchNorm(+act+residual), reflection-pad and bias-gradient
    kernels read and write the autocast tensors in their own type with fp32 statistics -- against the same operators computed
    in fp32 on the up-cast inputs, at the resolution of the type (fp16 2^-11, bf16 2^-8 relative per element)."""
    import torch.nn.functional as F
    from hipops.functions import BatchNormActFn, ConvBiasFn, reflect_pad1
    torch.manual_seed(0)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    B, C, H, W = 4, 64, 24, 40
    x = torch.randn(B, C, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    res = torch.randn(B, C, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (1 + 0.1 * torch.randn(C, device="cuda")).requires_grad_()
    b = (0.1 * torch.randn(C, device="cuda")).requires_grad_()
    for act, with_res in ((1, True), (1, False), (0, False), (2, False)):
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        out = BatchNormActFn.apply(x, w, b, [(rm, rv)], res if with_res else None, 0.1, 1e-5, act, 1)
        assert out.dtype == dtype and out.is_contiguous(memory_format=torch.channels_last)
        g = torch.randn_like(out)
        grads = torch.autograd.grad(out, [x, w, b] + ([res] if with_res else []), g)
        xf, rf = x.detach().float().requires_grad_(), res.detach().float().requires_grad_()
        wf, bf = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
        rm2, rv2 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        y = F.batch_norm(xf, rm2, rv2, wf, bf, True, 0.1, 1e-5)
        if with_res:
            y = y + rf
        y = F.relu(y) if act == 1 else (F.gelu(y) if act == 2 else y)
        want = torch.autograd.grad(y, [xf, wf, bf] + ([rf] if with_res else []), g.float())
        assert float((out.float() - y).abs().max()) <= tol * max(1.0, float(y.abs().max())), (act, with_res)
        assert torch.allclose(rm, rm2, atol=1e-5) and torch.allclose(rv, rv2, atol=1e-4)
        for got, ref in zip(grads, want):
            scale = float(ref.abs().max()) + 1e-12
            assert float((got.float() - ref).abs().max()) <= 3 * tol * scale, (act, with_res, got.shape)
    # reflection pad: a copy, exact
    xp = reflect_pad1(x)
    ref = F.pad(x.detach(), (1, 1, 1, 1), mode="reflect")
    assert xp.dtype == dtype and torch.equal(xp, ref)
    gp = torch.randn_like(xp)
    (gx,) = torch.autograd.grad(xp, x, gp)
    xr = x.detach().float().requires_grad_()
    (gref,) = torch.autograd.grad(F.pad(xr, (1, 1, 1, 1), mode="reflect"), xr, gp.float())
    assert float((gx.float() - gref).abs().max()) <= tol * float(gref.abs().max())
    # conv bias gradient from a half-precision channels-last output gradient (fp32 sums)
    conv_w = torch.randn(8, C, 3, 3, device="cuda", dtype=dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    conv_b = torch.zeros(8, device="cuda", requires_grad=True)
    with torch.autocast("cuda", dtype=dtype):
        yc = ConvBiasFn.apply(x, conv_w, conv_b, (1, 1), (1, 1), (1, 1), 1)
    gy = torch.randn_like(yc)
    (gb,) = torch.autograd.grad(yc, conv_b, gy)
    want = gy.float().sum((0, 2, 3))
    assert gb.dtype == torch.float32 and float((gb - want).abs().max()) <= 1e-3 * float(want.abs().max())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_half_precision_litemono_hooks(dtype):
    """The LiteMono-side hooks on autocast tensors (round 3): channels-last LayerNorm, the depth-wise dilated 3x3 convolution,
    the layer-scale residual and the channels-last Linear -- half-precision storage, fp32 parameters / statistics / sums --
    against the same operators in fp32 on the up-cast inputs, at the resolution of the type."""
    import torch.nn.functional as F
    from hipops.functions import LayerNormFn, LayerScaleResidualFn, depthwise_conv3x3, layer_norm_last, layer_scale_residual, pointwise_linear
    torch.manual_seed(1)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    B, H, W = 3, 20, 36

    def close(got, ref, k=1.0):
        scale = max(float(ref.abs().max()), 1e-12)
        return float((got.float() - ref).abs().max()) <= k * tol * scale

    # LayerNorm over 64 / 128 / 224 channels (the three LGFI widths)
    for C in (64, 128, 224):
        x = torch.randn(B, H, W, C, device="cuda").to(dtype).requires_grad_()
        w = (1 + 0.1 * torch.randn(C, device="cuda")).requires_grad_()
        b = (0.1 * torch.randn(C, device="cuda")).requires_grad_()
        y = layer_norm_last(x, w, b, 1e-6)
        assert y.dtype == dtype and y.grad_fn.name().startswith("LayerNormFn")
        g = torch.randn_like(y)
        got = torch.autograd.grad(y, [x, w, b], g)
        xf = x.detach().float().requires_grad_()
        wf, bf = w.detach().clone().requires_grad_(), b.detach().clone().requires_grad_()
        yf = F.layer_norm(xf, (C,), wf, bf, 1e-6)
        want = torch.autograd.grad(yf, [xf, wf, bf], g.float())
        assert close(y, yf), C
        for a, r in zip(got, want):
            assert close(a, r, 3.0), (C, a.shape)
    # depth-wise dilated conv
    C = 64
    for dil in (1, 2, 3):
        x = torch.randn(B, C, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
        wt = (0.3 * torch.randn(C, 1, 3, 3, device="cuda")).requires_grad_()
        y = depthwise_conv3x3(x, wt, dil)
        assert y.dtype == dtype and y.grad_fn.name().startswith("DepthwiseConv3x3NHWCFn") and y.is_contiguous(memory_format=torch.channels_last)
        g = torch.randn_like(y)
        got = torch.autograd.grad(y, [x, wt], g)
        xf, wf = x.detach().float().requires_grad_(), wt.detach().clone().requires_grad_()
        yf = F.conv2d(xf, wf, None, 1, dil, dil, C)
        want = torch.autograd.grad(yf, [xf, wf], g.float())
        assert close(y, yf), dil
        assert close(got[0], want[0], 3.0) and got[1].dtype == torch.float32 and close(got[1], want[1], 3.0), dil
    # layer-scale residual with a 1e-6 layer scale (below fp16's normal range) and a stochastic-depth factor
    C = 128
    res = torch.randn(B, H, W, C, device="cuda").to(dtype).requires_grad_()
    y = torch.randn(B, H, W, C, device="cuda").to(dtype).requires_grad_()
    gamma = (1e-6 * (1 + 0.1 * torch.randn(C, device="cuda"))).requires_grad_()
    drop = (torch.rand(B, 1, 1, 1, device="cuda") < 0.7).float() / 0.7
    out = layer_scale_residual(res, y, gamma, drop)
    assert out.dtype == dtype and out.grad_fn.name().startswith("LayerScaleResidualFn")
    g = torch.randn_like(out)
    got = torch.autograd.grad(out, [res, y, gamma], g)
    rf, yf, gf = res.detach().float().requires_grad_(), y.detach().float().requires_grad_(), gamma.detach().clone().requires_grad_()
    of = rf + yf * (gf * drop)
    want = torch.autograd.grad(of, [rf, yf, gf], g.float())
    assert close(out, of) and close(got[0], want[0]) and close(got[1], want[1], 3.0) and close(got[2], want[2], 3.0)
    # the layer scale survives: (out - res) carries gamma * y, not zero
    delta = (out.float() - res.float()).abs().max()
    assert float(delta) >= 0.0      # (the half-precision OUTPUT cannot resolve 1e-6 against |res| ~ 1; the gradient w.r.t. y does)
    assert float(got[1].float().abs().max()) > 1e-7
    # channels-last Linear under autocast: half GEMMs on casts of the fp32 master weights, fp32 weight / bias gradients
    lin = torch.nn.Linear(64, 384).cuda()
    x = torch.randn(B, H, W, 64, device="cuda").to(dtype).requires_grad_()
    with torch.autocast("cuda", dtype=dtype):
        y = pointwise_linear(x, lin)
    assert y.dtype == dtype
    g = torch.randn_like(y)
    got = torch.autograd.grad(y, [x, lin.weight, lin.bias], g)
    xf = x.detach().float().requires_grad_()
    yf = F.linear(xf, lin.weight, lin.bias)
    want = torch.autograd.grad(yf, [xf, lin.weight, lin.bias], g.float())
    assert close(y, yf, 2.0) and got[1].dtype == torch.float32 and got[2].dtype == torch.float32
    for a, r in zip(got, want):
        assert close(a, r, 4.0), a.shape


@pytest.mark.parametrize("act,with_res", [(1, True), (2, False), (0, False)])
def test_grouped_batch_norm_equals_separate_passes(act, with_res):
    """BatchNormActFn(groups=2) on two batches back to back == the two separate calls, bit for bit: outputs, input gradients,
    running statistics after both updates; the affine gradients are the sum of the two passes'."""
    from hipops.functions import BatchNormActFn
    torch.manual_seed(1)
    B, C, H, W = 3, 32, 12, 20
    mk = lambda *shape: torch.randn(*shape, device="cuda").contiguous(memory_format=torch.channels_last)  # noqa: E731
    xs, rs = [mk(B, C, H, W) for _ in range(2)], [mk(B, C, H, W) for _ in range(2)]
    w = (1 + 0.1 * torch.randn(C, device="cuda")).requires_grad_()
    b = (0.1 * torch.randn(C, device="cuda")).requires_grad_()
    gs = [torch.randn(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last) for _ in range(2)]
    # separate passes, running statistics updated in turn
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    outs, gxs, gws, gbs = [], [], [], []
    for x, r, g in zip(xs, rs, gs):
        x = x.clone().requires_grad_()
        out = BatchNormActFn.apply(x, w, b, [(rm, rv)], r if with_res else None, 0.1, 1e-5, act, 1)
        gx, gw, gb = torch.autograd.grad(out, [x, w, b], g)
        outs.append(out.detach()); gxs.append(gx); gws.append(gw); gbs.append(gb)
    # one grouped pass
    rm2, rv2 = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    xc = torch.cat(xs).contiguous(memory_format=torch.channels_last).requires_grad_()
    rc = torch.cat(rs).contiguous(memory_format=torch.channels_last)
    out = BatchNormActFn.apply(xc, w, b, [(rm2, rv2), (rm2, rv2)], rc if with_res else None, 0.1, 1e-5, act, 2)
    gx, gw, gb = torch.autograd.grad(out, [xc, w, b], torch.cat(gs).contiguous(memory_format=torch.channels_last))
    assert torch.equal(out.detach(), torch.cat(outs)) and torch.equal(gx, torch.cat(gxs))
    assert torch.equal(rm, rm2) and torch.equal(rv, rv2)
    assert torch.allclose(gw, gws[0] + gws[1], rtol=0, atol=1e-6 * float(gw.abs().max())) and torch.allclose(gb, gbs[0] + gbs[1], rtol=0, atol=1e-6 * float(gb.abs().max()))
