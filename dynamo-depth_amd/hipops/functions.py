"""torch.autograd bridges onto the C ABI (libdynamo_hip.so): one Function per tools.py operator.

torch is used here only for device memory, the current HIP stream and the autograd tape; every
computation is a HIP kernel behind include/dynamo_hip.h.  There is no CPU path: CPU tensors raise.
"""
import os

import torch

from . import lib as L


def _dev(t, name="tensor"):
    if not t.is_cuda:
        raise L.DynamoHipError("%s must live on the GPU: the Dynamo-Depth loss path has no CPU implementation "
                               "(libdynamo_hip.so is the only backend)" % name)
    if t.dtype != torch.float32:
        raise L.DynamoHipError("%s must be float32, got %s" % (name, t.dtype))
    return t.contiguous()


# element-type codes of the *_t entry points (include/dynamo_hip.h): the tensors of an autocast forward keep their own type
DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _p(t):
    """Device address for a `void*` parameter: ctypes takes the plain integer (None -> NULL); no c_void_p object per argument."""
    return None if t is None else t.data_ptr()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes) // 4, 1), dtype=torch.float32, device=device)


_WS_BYTES = {}


def _ws_bytes(name, *dims):
    """Memoised dd_*_workspace_bytes(dims...): the sizes depend on the shapes only; saves one foreign call per launch."""
    key = (name,) + dims
    n = _WS_BYTES.get(key)
    if n is None:
        n = _WS_BYTES[key] = int(getattr(L.load(), name)(*dims))
    return n


class BackprojectFn(torch.autograd.Function):
    """tools.BackprojectDepth.forward (reference tools.py:191-197)."""

    @staticmethod
    def forward(ctx, depth, inv_K):
        depth, inv_K = _dev(depth, "depth"), _dev(inv_K, "inv_K")
        B, _, h, w = depth.shape
        pts = torch.empty(B, 4, h * w, dtype=torch.float32, device=depth.device)
        L.check(L.load().dd_backproject(_p(depth), _p(inv_K), B, h, w, _p(pts), L.current_stream()), "dd_backproject")
        ctx.save_for_backward(inv_K)
        ctx.shape = (B, h, w)
        return pts

    @staticmethod
    def backward(ctx, g):
        (inv_K,) = ctx.saved_tensors
        B, h, w = ctx.shape
        g = _dev(g, "grad")
        gd = torch.empty(B, 1, h, w, dtype=torch.float32, device=g.device)
        L.check(L.load().dd_backproject_bwd(_p(g), _p(inv_K), B, h, w, _p(gd), L.current_stream()), "dd_backproject_bwd")
        return gd, None


class Project3DFn(torch.autograd.Function):
    """tools.Project3D.forward (reference tools.py:211-224)."""

    @staticmethod
    def forward(ctx, points, K, T, h, w, eps):
        points, K = _dev(points, "points"), _dev(K, "K")
        T = None if T is None else _dev(T, "T")
        B = points.shape[0]
        pix = torch.empty(B, h, w, 2, dtype=torch.float32, device=points.device)
        ego = torch.empty(B, 3, h * w, dtype=torch.float32, device=points.device)
        L.check(L.load().dd_project3d(_p(points), _p(K), _p(T), B, h, w, eps, _p(pix), _p(ego), L.current_stream()), "dd_project3d")
        ctx.save_for_backward(points, K, T if T is not None else torch.empty(0, device=points.device))
        ctx.has_T = T is not None
        ctx.dims = (B, h, w, eps)
        return pix, ego

    @staticmethod
    def backward(ctx, g_pix, g_ego):
        points, K, T = ctx.saved_tensors
        B, h, w, eps = ctx.dims
        T = T if ctx.has_T else None
        g_pix = None if g_pix is None else _dev(g_pix, "g_pix")
        g_ego = None if g_ego is None else _dev(g_ego, "g_ego")
        lib = L.load()
        g_points = torch.empty_like(points)
        g_T = torch.empty(B, 4, 4, dtype=torch.float32, device=points.device) if T is not None else None
        ws = _ws(lib.dd_project3d_workspace_bytes(B, h, w), points.device)
        L.check(lib.dd_project3d_bwd(_p(points), _p(K), _p(T), _p(g_pix), _p(g_ego), B, h, w, eps, _p(g_points), _p(g_T), _p(ws),
                                     L.current_stream()), "dd_project3d_bwd")
        return g_points, None, g_T, None, None, None


class SSIMFn(torch.autograd.Function):
    """tools.SSIM.forward (reference tools.py:243-257)."""

    @staticmethod
    def forward(ctx, x, y):
        x, y = _dev(x, "x"), _dev(y, "y")
        B, Cc, H, W = x.shape
        out = torch.empty_like(x)
        L.check(L.load().dd_ssim(_p(x), _p(y), B, Cc, H, W, _p(out), L.current_stream()), "dd_ssim")
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        B, Cc, H, W = x.shape
        g = _dev(g, "grad")
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        L.check(L.load().dd_ssim_bwd(_p(x), _p(y), _p(g), B, Cc, H, W, _p(gx), _p(gy), L.current_stream()), "dd_ssim_bwd")
        return gx, gy


class DispToDepthFn(torch.autograd.Function):
    """tools.disp_to_depth (reference tools.py:291-298)."""

    @staticmethod
    def forward(ctx, disp, min_depth, max_depth):
        disp = _dev(disp, "disp")
        scaled, depth = torch.empty_like(disp), torch.empty_like(disp)
        L.check(L.load().dd_disp_to_depth(_p(disp), disp.numel(), min_depth, max_depth, _p(scaled), _p(depth), L.current_stream()),
                "dd_disp_to_depth")
        ctx.save_for_backward(depth)
        ctx.span = 1.0 / min_depth - 1.0 / max_depth
        return scaled, depth

    @staticmethod
    def backward(ctx, g_scaled, g_depth):
        (depth,) = ctx.saved_tensors
        g = torch.zeros_like(depth)
        if g_scaled is not None:
            g = g + g_scaled * ctx.span
        if g_depth is not None:
            g = g - g_depth * depth * depth * ctx.span
        return g, None, None


class PoseMatrixFn(torch.autograd.Function):
    """networks.layers.transformation_from_parameters (reference networks/layers.py:7-82)."""

    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        B = axisangle.shape[0]
        aa = _dev(axisangle.reshape(B, 3), "axisangle")
        tr = _dev(translation.reshape(B, 3), "translation")
        T = torch.empty(B, 4, 4, dtype=torch.float32, device=aa.device)
        L.check(L.load().dd_pose_matrix(_p(aa), _p(tr), B, int(bool(invert)), _p(T), L.current_stream()), "dd_pose_matrix")
        ctx.save_for_backward(aa, tr)
        ctx.invert = int(bool(invert))
        ctx.shapes = (axisangle.shape, translation.shape)
        return T

    @staticmethod
    def backward(ctx, g):
        aa, tr = ctx.saved_tensors
        B = aa.shape[0]
        g = _dev(g, "grad")
        ga, gt = torch.empty_like(aa), torch.empty_like(tr)
        L.check(L.load().dd_pose_matrix_bwd(_p(aa), _p(tr), _p(g), B, ctx.invert, _p(ga), _p(gt), L.current_stream()), "dd_pose_matrix_bwd")
        return ga.reshape(ctx.shapes[0]), gt.reshape(ctx.shapes[1]), None


class SmoothLossFn(torch.autograd.Function):
    """tools.compute_smooth_loss (reference tools.py:311-326); normalise=True folds in Trainer.py:357-359."""

    @staticmethod
    def forward(ctx, inp, img, normalise):
        inp = _dev(inp, "inp")
        img = None if img is None else _dev(img, "img")
        B, Cc, h, w = inp.shape
        lib = L.load()
        g = torch.zeros_like(inp)
        sums = torch.empty(2, dtype=torch.float32, device=inp.device)
        ws = _ws(lib.dd_smooth_workspace_bytes(B, Cc, h, w), inp.device)
        L.check(lib.dd_smooth_loss(_p(inp), _p(img), B, Cc, h, w, int(bool(normalise)), 1.0, _p(g), _p(sums), _p(ws), L.current_stream()),
                "dd_smooth_loss")
        ctx.save_for_backward(g)
        return sums[0] / (B * Cc * h * (w - 1)) + sums[1] / (B * Cc * (h - 1) * w)

    @staticmethod
    def backward(ctx, go):
        (g,) = ctx.saved_tensors
        return g * go, None, None


class ConvBiasFn(torch.autograd.Function):
    """conv2d with bias whose BIAS gradient is computed by dd_channel_sum_nhwc when the output gradient is channels-last.
    ATen computes it with a generic reduction that takes 1.9 ms for one full-resolution 9-channel conv of the motion decoders
    (4 such convs per step); everything else (forward, input / weight gradients) stays MIOpen via the aten ops."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, groups):
        ctx.save_for_backward(x, weight)
        ctx.conf = (stride, padding, dilation, groups, bias.shape[0])
        return torch.nn.functional.conv2d(x, weight, bias, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, padding, dilation, groups, cout = ctx.conf
        if x.dtype != g.dtype:                       # autocast: the forward ran in reduced precision
            x = x.to(g.dtype)
        if weight.dtype != g.dtype:
            weight = weight.to(g.dtype)
        cin = x.shape[1]
        head = (ctx.needs_input_grad[0] and cout == 1 and groups == 1 and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1)
                and tuple(dilation) == (1, 1) and tuple(padding) in ((0, 0), (1, 1)) and g.is_cuda and g.dtype == torch.float32
                and x.dtype == torch.float32 and cin % 4 == 0 and cin <= 512 and x.is_contiguous(memory_format=torch.channels_last)
                and not x.is_contiguous())
        mask = (ctx.needs_input_grad[0] and not head, ctx.needs_input_grad[1], False)
        gx, gw, _ = torch.ops.aten.convolution_backward(g, x, weight, None, stride, padding, dilation, False, [0, 0], groups, mask)
        if head:
            # one output channel: MIOpen takes its naive kernel for this data gradient (165 us); it is an outer product
            B, _, Hi, Wi = x.shape
            gx = torch.empty_like(x)
            L.check(L.load().dd_conv3x3_cout1_bwd_data(_p(g.contiguous()), _p(weight.contiguous()), B, Hi, Wi, cin, padding[0], _p(gx),
                                                       L.current_stream()), "dd_conv3x3_cout1_bwd_data")
        gb = None
        if ctx.needs_input_grad[2]:
            fast = g.is_cuda and g.dtype in DTYPE_CODE and g.dim() == 4 and cout <= 256
            if fast and not g.is_contiguous():
                # channels-last or an arbitrary strided view (slice of a cat gradient): ATen's reduction is pathologically
                # slow on these (2 ms for (12,9,192,640)); a channels-last copy (if needed) + the HIP kernel is ~50 us
                gl = g if g.is_contiguous(memory_format=torch.channels_last) else g.contiguous(memory_format=torch.channels_last)
                lib = L.load()
                gb = torch.empty(cout, dtype=torch.float32, device=g.device)
                ws = _ws(_ws_bytes("dd_channel_sum_workspace_bytes", cout), g.device)
                B, _, H, W = gl.shape
                L.check(lib.dd_channel_sum_nhwc_t(_p(gl), B * H * W, cout, _p(gb), DTYPE_CODE[gl.dtype], _p(ws), L.current_stream()),
                        "dd_channel_sum_nhwc_t")
            else:
                gb = g.sum((0, 2, 3))
        if gw is not None and gw.dtype != ctx.saved_tensors[1].dtype:
            gw = gw.to(ctx.saved_tensors[1].dtype)
        return gx, gw, (gb.to(torch.float32) if gb is not None else None), None, None, None, None


def redu_ok(a, b, conv):
    """dd_redu covers conv(cat(a, b)): a 1x1 reduction of two equal channels-last fp32 tensors of 64 / 128 / 256 / 512 channels to 1 or 3."""
    if os.environ.get("DD_STOCK_REDU", "0") == "1" or torch.is_autocast_enabled():
        return False
    w = conv.weight
    if not (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and w.dtype == torch.float32 and a.dim() == 4 and a.shape == b.shape):
        return False
    B, Cc, H, W = a.shape
    if tuple(w.shape) != (w.shape[0], 2 * Cc, 1, 1) or conv.groups != 1 or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (0, 0):
        return False
    nhwc = (H * W * Cc, 1, W * Cc, Cc)
    if a.stride() != nhwc or b.stride() != nhwc:
        return False
    return bool(L.load().dd_redu_supported(Cc, w.shape[0]))


class ReduFn(torch.autograd.Function):
    """conv1x1(cat(a, b)) + bias through dd_redu (csrc/dd_redu.hip): the motion decoders' reductions to 3 / 1 channels (reference
    networks/motion_decoder.py:33,66) -- one launch forward, one for both data gradients, one pass + fold for weight and bias gradient."""

    @staticmethod
    def forward(ctx, a, b, weight, bias):
        lib = L.load()
        B, Cc, H, W = a.shape
        cout = weight.shape[0]
        y = torch.empty((B, cout, H, W), dtype=torch.float32, device=a.device).as_strided((B, cout, H, W), (H * W * cout, 1, W * cout, cout))
        wm = weight.reshape(cout, 2 * Cc)                  # (cout, 2C, 1, 1) in either layout is (cout, 2C) rows in memory
        if not wm.is_contiguous():
            wm = wm.contiguous()
        L.check(lib.dd_redu_fwd(_p(a), _p(b), _p(wm), _p(bias), B * H * W, Cc, cout, _p(y), L.current_stream()), "dd_redu_fwd")
        ctx.save_for_backward(a, b, wm)
        ctx.has_bias = bias is not None
        ctx.wshape = tuple(weight.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b, wm = ctx.saved_tensors
        lib = L.load()
        B, Cc, H, W = a.shape
        cout = wm.shape[0]
        P = B * H * W
        g = _dense_nhwc(g.to(torch.float32))
        ga = gb = gw = gbias = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
            gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
            L.check(lib.dd_redu_bwd_data(_p(g), _p(wm), P, Cc, cout, _p(ga), _p(gb), L.current_stream()), "dd_redu_bwd_data")
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            gw = torch.empty(cout, 2 * Cc, dtype=torch.float32, device=g.device)
            gbias = torch.empty(cout, dtype=torch.float32, device=g.device) if ctx.has_bias else None
            nbytes = _ws_bytes("dd_redu_workspace_bytes", P, Cc, cout)
            ws = _ws(nbytes, g.device)
            L.check(lib.dd_redu_bwd_weight(_p(a), _p(b), _p(g), P, Cc, cout, _p(gw), _p(gbias), _p(ws), nbytes, L.current_stream()), "dd_redu_bwd_weight")
            gw = gw.view(ctx.wshape)
        return ga, gb, gw, gbias


def head_conv_ok(x, weight, stride, padding, dilation, groups):
    """dd_conv_head covers this convolution: a disparity head -- 3x3, one output channel, no padding of its own (the input carries the
    reflection padding), 32 or 64 channels-last fp32 input channels."""
    if os.environ.get("DD_STOCK_HEAD_CONV", "0") == "1":
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)):
        return False
    if weight.shape[0] != 1 or groups != 1 or tuple(stride) != (1, 1) or tuple(dilation) != (1, 1) or tuple(padding) != (0, 0):
        return False
    B, Cc, Hp, Wp = x.shape
    if weight.shape[1] != Cc or Hp < 3 or Wp < 3 or torch.is_autocast_enabled() or x.stride() != (Hp * Wp * Cc, 1, Wp * Cc, Cc):
        return False
    return bool(L.load().dd_conv_head_supported(Cc))


class HeadConvFn(torch.autograd.Function):
    """A disparity head through dd_conv_head (csrc/dd_conv_head.hip; reference networks/depth_decoder.py:49-51,95-97): forward one
    launch, weight + bias gradient in one pass over the input, data gradient dd_conv3x3_cout1_bwd_data."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = L.load()
        B, Cc, Hp, Wp = x.shape
        out = torch.empty((B, 1, Hp - 2, Wp - 2), dtype=torch.float32, device=x.device)
        sw = weight.stride()
        L.check(lib.dd_conv_head_fwd(_p(x), _p(weight), sw[1], sw[2], sw[3], _p(bias), B, Hp, Wp, Cc, _p(out), L.current_stream()), "dd_conv_head_fwd")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        lib = L.load()
        B, Cc, Hp, Wp = x.shape
        g = g.to(torch.float32).contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            L.check(lib.dd_conv3x3_cout1_bwd_data(_p(g), _p(weight.contiguous()), B, Hp, Wp, Cc, 0, _p(gx), L.current_stream()), "dd_conv3x3_cout1_bwd_data")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            flat = torch.empty(9 * Cc, dtype=torch.float32, device=g.device)
            gb = torch.empty(1, dtype=torch.float32, device=g.device) if ctx.has_bias else None
            nbytes = int(lib.dd_conv_head_workspace_bytes(B, Hp, Wp, Cc))
            ws = _ws(nbytes, g.device)
            L.check(lib.dd_conv_head_bwd_weight(_p(x), _p(g), B, Hp, Wp, Cc, _p(flat), _p(gb), _p(ws), nbytes, L.current_stream()), "dd_conv_head_bwd_weight")
            gw = flat.view(1, 3, 3, Cc).permute(0, 3, 1, 2)
        return gx, gw, gb


def small_conv_ok(x, weight, stride, padding, dilation, groups):
    """dd_conv_small covers this convolution: a fp32 channels-last CUDA tensor with <= 16 channels in and out at a resolution where
    the library's implicit-GEMM tiles are mostly padding (the motion decoders' finest level), 1x1 or 3x3, stride 1, 'same' padding."""
    if os.environ.get("DD_STOCK_SMALL_CONV", "0") == "1":
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and weight.dim() == 4):
        return False
    cout, cin, kh, kw = weight.shape
    if kh != kw or kh not in (1, 3) or groups != 1 or tuple(stride) != (1, 1) or tuple(dilation) != (1, 1) or tuple(padding) != (kh // 2, kh // 2):
        return False
    if x.shape[1] != cin or x.shape[0] * x.shape[2] * x.shape[3] < (1 << 16):
        return False
    if torch.is_autocast_enabled():
        return False
    return bool(L.load().dd_conv_small_supported(kh, cin, cout))


def _dense_nhwc(t):
    """t (B,C,H,W) with dense channels-last memory (a copy unless it already is)."""
    B, Cc, H, W = t.shape
    if t.stride() == (H * W * Cc, 1, W * Cc, Cc):
        return t
    return t.contiguous(memory_format=torch.channels_last).as_strided((B, Cc, H, W), (H * W * Cc, 1, W * Cc, Cc)) if Cc > 1 else \
        t.contiguous().as_strided((B, Cc, H, W), (H * W, 1, W, 1))


_SMALL_CONV_CALLS = [0]


def small_conv_calls():
    """How many times SmallConvFn.forward has launched dd_conv_small_fwd in this process (eager steps and graph captures): what
    bench.py reports instead of inferring the path from flags, and what the hooks-vs-stock tests assert on."""
    return _SMALL_CONV_CALLS[0]


class SmallConvFn(torch.autograd.Function):
    """conv2d (+ bias) through dd_conv_small (csrc/dd_conv_small.hip): the motion decoders' full-resolution convolutions on 9-12
    channels (reference networks/motion_decoder.py:24-33,57-66).  Forward and data gradient are a direct convolution, the weight
    gradient runs on the matrix pipe and yields the bias gradient in the same pass; every result is bit-reproducible."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = L.load()
        cout, cin, ks, _ = weight.shape
        B, _, H, W = x.shape
        x = _dense_nhwc(x)
        y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device).as_strided((B, cout, H, W), (H * W * cout, 1, W * cout, cout))
        nbytes = _ws_bytes("dd_conv_small_workspace_bytes", ks, cin, cout)
        ws = _ws(nbytes, x.device)
        sw = weight.stride()
        L.check(lib.dd_conv_small_fwd(_p(x), _p(weight), sw[0], sw[1], sw[2], sw[3], _p(bias), B, H, W, cin, cout, ks, _p(y), _p(ws), nbytes,
                                      L.current_stream()), "dd_conv_small_fwd")
        _SMALL_CONV_CALLS[0] += 1
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        lib = L.load()
        cout, cin, ks, _ = weight.shape
        B, _, H, W = x.shape
        g = _dense_nhwc(g.to(torch.float32))
        nbytes = _ws_bytes("dd_conv_small_workspace_bytes", ks, cin, cout)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty((B, cin, H, W), dtype=torch.float32, device=g.device).as_strided((B, cin, H, W), (H * W * cin, 1, W * cin, cin))
            ws = _ws(nbytes, g.device)
            sw = weight.stride()
            L.check(lib.dd_conv_small_bwd_data(_p(g), _p(weight), sw[0], sw[1], sw[2], sw[3], B, H, W, cin, cout, ks, _p(gx), _p(ws), nbytes,
                                               L.current_stream()), "dd_conv_small_bwd_data")
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            flat = torch.empty(cout * ks * ks * cin, dtype=torch.float32, device=g.device)
            gb = torch.empty(cout, dtype=torch.float32, device=g.device) if ctx.has_bias else None
            ws = _ws(nbytes, g.device)
            L.check(lib.dd_conv_small_bwd_weight(_p(x), _p(g), B, H, W, cin, cout, ks, _p(flat), _p(gb), _p(ws), nbytes, L.current_stream()),
                    "dd_conv_small_bwd_weight")
            gw = flat.view(cout, ks, ks, cin).permute(0, 3, 1, 2)           # (cout,cin,ks,ks) on channels-last memory
        return gx, gw, gb


def small_conv(x, weight, bias=None):
    return SmallConvFn.apply(x, weight, bias)


def mfma_conv_ok(x, weight, stride, padding, dilation, groups):
    """dd_conv3x3_mfma covers this convolution: 3x3, stride 1, padding 0 or 1, fp32, 16+ channels in and out, on a channels-last CUDA
    tensor with enough pixels to fill the chip (csrc/dd_conv_mfma.hip: fp32 accuracy from three bf16 pieces per operand)."""
    if os.environ.get("DD_STOCK_MFMA_CONV", "0") == "1":
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)):
        return False
    cout, cin = weight.shape[:2]
    if groups != 1 or tuple(stride) != (1, 1) or tuple(dilation) != (1, 1) or tuple(padding) not in ((0, 0), (1, 1)) or x.shape[1] != cin:
        return False
    if torch.is_autocast_enabled():
        return False
    pad = padding[0]
    Ho, Wo = x.shape[2] + 2 * pad - 2, x.shape[3] + 2 * pad - 2
    # below ~20 k output pixels the 8 x 32-pixel tiles no longer fill the chip (12 x 24 x 80 is still ahead of the library on all three
    # passes, 12 x 12 x 40 is not: profiles/r05_conv_mfma.txt)
    if Ho < 8 or Wo < 32 or x.shape[0] * Ho * Wo < int(os.environ.get("DD_MFMA_CONV_MIN_PIXELS", "20000")):
        # small images (the encoders' and decoders' deep levels): flat pixel tiles + split contraction, forward and data gradient
        return pad == 1 and _flat_conv_ok(x.shape[0], x.shape[2], x.shape[3], cin, cout)
    return bool(L.load().dd_conv3x3_mfma_supported(cin, cout))


def _flat_conv_ok(B, H, W, cin, cout):
    """dd_conv3x3_mfma_flat serves this pad-1 convolution in BOTH directions (the data gradient swaps the channel counts)."""
    if os.environ.get("DD_FLAT_MFMA_CONV", "1") != "1" or min(cin, cout) < int(os.environ.get("DD_FLAT_MIN_CHANNELS", "256")):
        return False
    lib = L.load()
    return bool(lib.dd_conv3x3_mfma_flat_supported(B, H, W, cin, cout)) and bool(lib.dd_conv3x3_mfma_flat_supported(B, H, W, cout, cin))


def _flat_shape(B, H, W, pad, k_in=None, n_out=None):
    """Does MfmaConvFn take the flat-tile kernel for this input?  (the complement of the tile kernel's domain, see mfma_conv_ok;
    k_in / n_out: the channel counts of THIS pass -- the data gradient swaps them)"""
    Ho, Wo = H + 2 * pad - 2, W + 2 * pad - 2
    if not (pad == 1 and (Ho < 8 or Wo < 32 or B * Ho * Wo < int(os.environ.get("DD_MFMA_CONV_MIN_PIXELS", "20000")))):
        return False
    if os.environ.get("DD_FLAT_MFMA_CONV", "1") != "1":
        return False
    return k_in is None or bool(L.load().dd_conv3x3_mfma_flat_supported(B, H, W, k_in, n_out))


_MFMA_CONV_CALLS = [0]
_FLAT_CONV_CALLS = [0]      # ... of which dd_conv3x3_mfma_flat served (small images)


def mfma_conv_calls():
    """How many times MfmaConvFn.forward has launched dd_conv3x3_mfma in this process (bench.py reports it)."""
    return _MFMA_CONV_CALLS[0]


_PRODUCTS = {"highest": 6, "high": 3, "medium": 1}


def mfma_products():
    """Partial products per multiply-add of dd_conv3x3_mfma, from PyTorch's own switch for fp32 contractions:
    torch.set_float32_matmul_precision("highest") (the default) -> 6, all that reach fp32's last place: fp32 accuracy;
    "high" -> 3 ("bf16x3" in that function's documentation: two bf16 pieces per operand, products to 2^-16 -- cuDNN's default for the
    reference's fp32 convolutions on Ampere-class GPUs is TF32, 2^-11); "medium" -> 1 (operands rounded to bf16, fp32 accumulation).
    DD_MFMA_PRODUCTS overrides.  options.py: --matmul_precision."""
    env = os.environ.get("DD_MFMA_PRODUCTS")
    return int(env) if env else _PRODUCTS[torch.get_float32_matmul_precision()]


def _nhwc_empty(B, Cc, H, W, device):
    return torch.empty((B, H, W, Cc), dtype=torch.float32, device=device).permute(0, 3, 1, 2)


class _PackEntry:
    __slots__ = ("weight", "pack_f", "pack_b", "fresh", "sig")

    def __init__(self, weight, pack_f, pack_b):
        self.weight, self.pack_f, self.pack_b, self.fresh = weight, pack_f, pack_b, 0
        self.sig = (weight.data_ptr(), tuple(weight.stride()), tuple(weight.shape))


class PackSet:
    """The dd_conv3x3_mfma weight packs of ONE network module, made by one launch at the top of its forward pass (round 6: the step had 62
    pack launches of 4-8 us, each in front of its convolution on the network's stream; dd_conv3x3_mfma_pack_many makes a network's in one).
    A layer joins the set the first time MfmaConvFn.forward runs inside the module's forward (it packs alone that time, its buffers become
    the set's); from the next pass on the module's forward-pre hook packs every member and marks it fresh, the layer takes the fresh pack
    ONCE -- any other call (outside the module's forward, a second use in one pass, a weight that moved) packs alone as before, so a pack
    is never older than the forward pass that uses it.  Job tables and pack buffers are never freed: captured graphs hold their addresses."""

    def __init__(self):
        self.entries = {}          # id(weight) -> _PackEntry
        self.tables = {}           # (with data-gradient packs?, members' signatures) -> (jobs, block_job, n_blocks)
        self.retired = []          # buffers replaced by an upgrade (a member that later needed its data-gradient pack)

    def pack_all(self):
        if not self.entries:
            return
        with_b = torch.is_grad_enabled()
        members = list(self.entries.values())
        # a member whose weight moved (another device, another layout) leaves the set: its layer re-joins with the next call
        for e in members:
            if e.sig != (e.weight.data_ptr(), tuple(e.weight.stride()), tuple(e.weight.shape)):
                self.retired.append(self.entries.pop(id(e.weight)))
        members = list(self.entries.values())
        if not members:
            return
        sig = tuple((e.sig, e.pack_f.data_ptr(), 0 if e.pack_b is None else e.pack_b.data_ptr()) for e in members)
        table = self.tables.get((with_b, sig))
        lib = L.load()
        if table is None:
            if torch.cuda.is_current_stream_capturing():
                return             # no host-to-device copy inside a capture: this pass packs layer by layer
            for mode in (True, False):          # both tables at once: a tape-free pass of the same members may first come inside a capture
                self.tables[(mode, sig)] = self._table(lib, members, mode)
            table = self.tables[(with_b, sig)]
        L.check(lib.dd_conv3x3_mfma_pack_many(_p(table[0]), _p(table[1]), table[2], L.current_stream()), "dd_conv3x3_mfma_pack_many")
        _PACK_MANY_LAUNCHES[0] += 1
        for e in members:
            e.fresh = 2 if (with_b and e.pack_b is not None) else 1

    @staticmethod
    def _table(lib, members, with_b):
        words = lib.dd_conv3x3_mfma_pack_many_job_words()
        jobs, owner, first = [], [], 0
        for n, e in enumerate(members):
            cout, cin = e.weight.shape[:2]
            has_b = with_b and e.pack_b is not None
            sw = e.weight.stride()
            row = [e.weight.data_ptr(), sw[0], sw[1], sw[2], sw[3], cout, cin, e.pack_f.data_ptr(), e.pack_b.data_ptr() if has_b else 0, first]
            assert len(row) == words
            jobs.append(row)
            nb = lib.dd_conv3x3_mfma_pack_many_blocks(cout, cin, 1, 1 if has_b else 0)
            owner += [n] * nb
            first += nb
        dev = members[0].weight.device
        return torch.tensor(jobs, dtype=torch.int64).to(dev), torch.tensor(owner, dtype=torch.int32).to(dev), first

    def take(self, weight, need_b):
        """The fresh pack of this weight, once: (pack_fwd, pack_bwd_data) or None."""
        e = self.entries.get(id(weight))
        if e is None or e.weight is not weight or e.fresh < (2 if need_b else 1):
            return None
        if e.sig != (weight.data_ptr(), tuple(weight.stride()), tuple(weight.shape)):
            return None
        e.fresh = 0
        return e.pack_f, (e.pack_b if need_b else None)

    def join(self, weight, pack_f, pack_b):
        old = self.entries.get(id(weight))
        if old is not None:
            if old.weight is weight and old.sig == (weight.data_ptr(), tuple(weight.stride()), tuple(weight.shape)) and (pack_b is None or old.pack_b is not None):
                return             # a second use in one pass, or a tape-free pass of a member: nothing to learn
            self.retired.append(old)
        self.entries[id(weight)] = _PackEntry(weight, pack_f, pack_b)

    def end(self):
        for e in self.entries.values():
            e.fresh = 0


_PACK_SETS = {}                # id(module) -> (weak reference to the module, PackSet)
_ACTIVE_PACK_SETS = []         # the sets of the modules whose forward is running (innermost last)
_PACK_MANY_LAUNCHES = [0]


def pack_many_launches():
    """How many times a network's packs were made by one dd_conv3x3_mfma_pack_many launch in this process."""
    return _PACK_MANY_LAUNCHES[0]


def _pack_many_on():
    # OPT-IN: measured neutral in the headline step (341.7 / 341.3 img/s with it, 342.2 / 343.5 without, same box back to back) although it
    # takes 54 launches and 0.32 ms of kernel time out of the step -- the packs were never on the step's critical path, and one 60 us
    # launch at the head of an encoder delays its first convolution more than the 4-8 us packs interleaved with other streams' work did
    return os.environ.get("DD_PACK_MANY", "0") == "1"


def pack_weights_once_per_forward(module):
    """Make `module` (a network: an encoder, a decoder) pack the weights of its dd_conv3x3_mfma layers in ONE launch at the top of every
    forward pass (PackSet) when DD_PACK_MANY=1.  Returns the module.  Default (0): every layer packs in front of its own convolution."""
    import weakref
    if id(module) in _PACK_SETS and _PACK_SETS[id(module)][0]() is module:
        return module
    ps = PackSet()
    _PACK_SETS[id(module)] = (weakref.ref(module, lambda _r, k=id(module): _PACK_SETS.pop(k, None)), ps)

    def before(_m, _inputs):
        _ACTIVE_PACK_SETS.append(ps)
        if _pack_many_on():
            ps.pack_all()

    def after(_m, _inputs, _outputs):
        ps.end()
        if _ACTIVE_PACK_SETS and _ACTIVE_PACK_SETS[-1] is ps:
            _ACTIVE_PACK_SETS.pop()

    module.register_forward_pre_hook(before)
    module.register_forward_hook(after, always_call=True)
    return module


class MfmaConvFn(torch.autograd.Function):
    """conv2d 3x3 stride 1 (+ bias) through dd_conv3x3_mfma (csrc/dd_conv_mfma.hip): the motion decoders' refinement convolutions
    (reference networks/motion_decoder.py:24-33,57-66).  Forward and data gradient run on the bf16 matrix pipe with every fp32 operand split
    exactly into three bf16 pieces (fp32 accuracy), and so does the weight gradient (pixels as the contraction, per-workgroup partials folded
    in a fixed order); the bias gradient is dd_channel_sum_nhwc.  Every result is bit-reproducible."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad):
        lib = L.load()
        cout, cin = weight.shape[:2]
        B, _, Hi, Wi = x.shape
        x = _dense_nhwc(x)
        need_gx = ctx.needs_input_grad[0]
        stream = L.current_stream()
        products = mfma_products()
        pset = _ACTIVE_PACK_SETS[-1] if (_ACTIVE_PACK_SETS and _pack_many_on()) else None
        packs = pset.take(weight, need_gx) if pset is not None else None
        if packs is not None:
            pack_f, pack_b = packs              # made at the top of this forward pass of the network (PackSet)
        else:
            pack_f = torch.empty(_ws_bytes("dd_conv3x3_mfma_pack_bytes", cout, cin) // 4, dtype=torch.float32, device=x.device)
            pack_b = torch.empty(_ws_bytes("dd_conv3x3_mfma_pack_bytes", cin, cout) // 4, dtype=torch.float32, device=x.device) if need_gx else None
            sw = weight.stride()
            L.check(lib.dd_conv3x3_mfma_pack(_p(weight), sw[0], sw[1], sw[2], sw[3], cout, cin, _p(pack_f), _p(pack_b), stream), "dd_conv3x3_mfma_pack")
            if pset is not None and isinstance(weight, torch.nn.Parameter):
                pset.join(weight, pack_f, pack_b)
        Ho, Wo = Hi + 2 * pad - 2, Wi + 2 * pad - 2
        y = _nhwc_empty(B, cout, Ho, Wo, x.device)
        if _flat_shape(B, Hi, Wi, pad, cin, cout):
            nbytes = _ws_bytes("dd_conv3x3_mfma_flat_workspace_bytes", B, Hi, Wi, cin, cout)
            ws = _ws(nbytes, x.device)
            L.check(lib.dd_conv3x3_mfma_flat_n(_p(x), _p(pack_f), _p(bias), B, Hi, Wi, cin, cout, products, _p(y), _p(ws), nbytes, stream), "dd_conv3x3_mfma_flat")
            _FLAT_CONV_CALLS[0] += 1
        else:
            L.check(lib.dd_conv3x3_mfma_n(_p(x), _p(pack_f), _p(bias), B, Hi, Wi, cin, cout, pad, products, _p(y), stream), "dd_conv3x3_mfma")
        _MFMA_CONV_CALLS[0] += 1
        ctx.save_for_backward(x, weight, pack_b)
        ctx.conf = (pad, bias is not None, products)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, pack_b = ctx.saved_tensors
        pad, has_bias, products = ctx.conf
        lib = L.load()
        cout, cin = weight.shape[:2]
        B, _, Hi, Wi = x.shape
        Ho, Wo = g.shape[2:]
        g = _dense_nhwc(g.to(torch.float32))
        gx = gw = gb = None
        stream = L.current_stream()
        if ctx.needs_input_grad[0]:
            gx = _nhwc_empty(B, cin, Hi, Wi, g.device)
            if _flat_shape(B, Hi, Wi, pad, cout, cin):
                nbytes = _ws_bytes("dd_conv3x3_mfma_flat_workspace_bytes", B, Ho, Wo, cout, cin)
                ws = _ws(nbytes, g.device)
                L.check(lib.dd_conv3x3_mfma_flat_n(_p(g), _p(pack_b), None, B, Ho, Wo, cout, cin, products, _p(gx), _p(ws), nbytes, stream), "dd_conv3x3_mfma_flat (data gradient)")
            else:
                L.check(lib.dd_conv3x3_mfma_n(_p(g), _p(pack_b), None, B, Ho, Wo, cout, cin, 2 - pad, products, _p(gx), stream), "dd_conv3x3_mfma (data gradient)")
        if ctx.needs_input_grad[1]:
            # the kernel accumulates 64 x 64 (cout x cin) blocks: with fewer than 32 channels on either side most of a block is
            # padding and the library's kernel is faster (profiles/r05_conv_mfma_fold4.txt: 16 -> 16 at 192x640 913 against 550 us;
            # 32 -> 32 at 96x320 265 against 286 us + the library's zero-fill since the fold runs four waves per result)
            # (small images: the library's weight gradient is as fast as ours there -- 76 against 79 us at 12x256x256x12x40 -- and stays)
            # (with fewer partial products -- mfma_products() -- ours is ahead there too: 54 against 74 us with three)
            if (cout % 4 == 0 and min(cin, cout) >= 32 and os.environ.get("DD_STOCK_MFMA_WGRAD", "0") != "1"
                    and (products < 6 or not _flat_shape(B, Hi, Wi, pad, cin, cout))):
                flat = torch.empty(cout * 9 * cin, dtype=torch.float32, device=g.device)
                nbytes = _ws_bytes("dd_conv3x3_mfma_wgrad_workspace_bytes", B, Ho, Wo, cin, cout)
                ws = _ws(nbytes, g.device)
                L.check(lib.dd_conv3x3_mfma_bwd_weight_n(_p(x), _p(g), B, Hi, Wi, cin, cout, pad, products, _p(flat), _p(ws), nbytes, stream), "dd_conv3x3_mfma_bwd_weight")
                gw = flat.view(cout, 3, 3, cin).permute(0, 3, 1, 2)           # (cout,cin,3,3) on channels-last memory
            else:
                _, gw, _ = torch.ops.aten.convolution_backward(g, x, weight, None, (1, 1), (pad, pad), (1, 1), False, [0, 0], 1, (False, True, False))
        if has_bias and ctx.needs_input_grad[2]:
            if cout <= 256:
                gb = torch.empty(cout, dtype=torch.float32, device=g.device)
                ws = _ws(_ws_bytes("dd_channel_sum_workspace_bytes", cout), g.device)
                L.check(lib.dd_channel_sum_nhwc_t(_p(g), B * Ho * Wo, cout, _p(gb), DTYPE_CODE[g.dtype], _p(ws), stream), "dd_channel_sum_nhwc_t")
            else:
                gb = g.sum((0, 2, 3))
        return gx, gw, gb, None


def mfma_conv(x, weight, bias=None, pad=1):
    return MfmaConvFn.apply(x, weight, bias, int(pad))


_HALF_CONV_CALLS = [0]


def half_conv_calls():
    """How many times HalfConvFn.forward has launched dd_conv3x3_half in this process (bench.py reports it)."""
    return _HALF_CONV_CALLS[0]


def half_conv_ok(x, weight, stride, padding, dilation, groups):
    """dd_conv3x3_half covers this convolution: a 3x3, stride-1, padding-0/1 layer of a network running under autocast whose input already
    is a channels-last fp16 / bf16 CUDA tensor (the producing hook wrote it in the half type), 16+ channels in eights on both sides, and
    enough pixels to fill the chip (csrc/dd_conv_half.hip; the small images of the deep levels stay with the library).  DD_HALF_MFMA_CONV=0
    leaves every half-precision convolution with the library."""
    if os.environ.get("DD_HALF_MFMA_CONV", "1") != "1" or not x.is_cuda or x.dim() != 4 or x.dtype not in (torch.float16, torch.bfloat16):
        return False
    if not torch.is_autocast_enabled() or torch.get_autocast_dtype("cuda") != x.dtype or weight.dtype not in (torch.float32, x.dtype):
        return False
    cout, cin = weight.shape[:2]
    if tuple(weight.shape[2:]) != (3, 3) or groups != 1 or tuple(stride) != (1, 1) or tuple(dilation) != (1, 1) or tuple(padding) not in ((0, 0), (1, 1)):
        return False
    if x.shape[1] != cin or not (x.is_contiguous(memory_format=torch.channels_last) or x.stride(1) == 1):
        return False
    pad = padding[0]
    Ho, Wo = x.shape[2] + 2 * pad - 2, x.shape[3] + 2 * pad - 2
    if Ho < 8 or Wo < 32 or x.shape[0] * Ho * Wo < int(os.environ.get("DD_HALF_CONV_MIN_PIXELS", "20000")):
        return False
    return bool(L.load().dd_conv3x3_half_supported(cin, cout))


class HalfConvFn(torch.autograd.Function):
    """conv2d 3x3 stride 1 (+ bias) of a half-precision network through dd_conv3x3_half (csrc/dd_conv_half.hip): forward and data gradient on
    the half-precision matrix pipe with fp32 accumulation, the fp32 master weight converted while it is packed (no cast launch), the
    weight gradient likewise with an fp32 result (32+ channels on both sides; below that the library's half-precision kernel, its result
    promoted to the master weight's fp32 as autocast's own backward does), the bias gradient through dd_channel_sum_nhwc.  BASELINE.json config 5 ("fp16 (CDNA4 MFMA conv)")."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad):
        lib = L.load()
        cout, cin = weight.shape[:2]
        B, _, Hi, Wi = x.shape
        x = _dense_nhwc(x)
        code = DTYPE_CODE[x.dtype]
        w32 = weight if weight.dtype == torch.float32 else weight.float()
        need_gx = ctx.needs_input_grad[0]
        pack_f = torch.empty(_ws_bytes("dd_conv3x3_half_pack_bytes", cout, cin) // 4, dtype=torch.float32, device=x.device)
        pack_b = torch.empty(_ws_bytes("dd_conv3x3_half_pack_bytes", cin, cout) // 4, dtype=torch.float32, device=x.device) if need_gx else None
        sw = w32.stride()
        stream = L.current_stream()
        L.check(lib.dd_conv3x3_half_pack(_p(w32), sw[0], sw[1], sw[2], sw[3], cout, cin, code, _p(pack_f), _p(pack_b), stream), "dd_conv3x3_half_pack")
        Ho, Wo = Hi + 2 * pad - 2, Wi + 2 * pad - 2
        y = torch.empty((B, Ho, Wo, cout), dtype=x.dtype, device=x.device).permute(0, 3, 1, 2)
        b32 = None if bias is None else (bias if bias.dtype == torch.float32 else bias.float())
        L.check(lib.dd_conv3x3_half(_p(x), _p(pack_f), _p(b32), B, Hi, Wi, cin, cout, pad, code, _p(y), stream), "dd_conv3x3_half")
        _HALF_CONV_CALLS[0] += 1
        ctx.save_for_backward(x, weight, pack_b)
        ctx.conf = (pad, bias is not None, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, pack_b = ctx.saved_tensors
        pad, has_bias, bias_dtype = ctx.conf
        lib = L.load()
        cout, cin = weight.shape[:2]
        B, _, Hi, Wi = x.shape
        Ho, Wo = g.shape[2:]
        g = _dense_nhwc(g.to(x.dtype))
        code = DTYPE_CODE[x.dtype]
        gx = gw = gb = None
        stream = L.current_stream()
        if ctx.needs_input_grad[0]:
            gx = torch.empty((B, Hi, Wi, cin), dtype=x.dtype, device=g.device).permute(0, 3, 1, 2)
            L.check(lib.dd_conv3x3_half(_p(g), _p(pack_b), None, B, Ho, Wo, cout, cin, 2 - pad, code, _p(gx), stream), "dd_conv3x3_half (data gradient)")
        if ctx.needs_input_grad[1]:
            if (min(cin, cout) >= 32 and weight.dtype == torch.float32 and os.environ.get("DD_STOCK_HALF_WGRAD", "0") != "1"
                    and B * Ho * Wo >= int(os.environ.get("DD_HALF_WGRAD_MIN_PIXELS", "100000"))):
                # own kernel: half x half products, fp32 accumulation, an fp32 result (64 x 64 channel blocks: below 32 channels on either
                # side most of a block is padding, as in MfmaConvFn; below ~100 k pixels the split contraction's partials cost more than
                # the library's kernel: 38.9 against 34.4 us at 16x128x128x36x64, 38.8 against 43.6 + cast + zero-fill at 16x64x64x72x128,
                # 78 against 130 at 16x64x64x144x256 -- profiles/r06_conv_half.txt)
                flat = torch.empty(cout * 9 * cin, dtype=torch.float32, device=g.device)
                nbytes = _ws_bytes("dd_conv3x3_half_wgrad_workspace_bytes", B, Ho, Wo, cin, cout)
                ws = _ws(nbytes, g.device)
                L.check(lib.dd_conv3x3_half_bwd_weight(_p(x), _p(g), B, Hi, Wi, cin, cout, pad, code, _p(flat), _p(ws), nbytes, stream), "dd_conv3x3_half_bwd_weight")
                gw = flat.view(cout, 3, 3, cin).permute(0, 3, 1, 2)           # (cout,cin,3,3) on channels-last memory
            else:
                # the library's half-precision weight gradient (only the shape, type and layout of the weight argument are read)
                w_like = torch.empty_like(weight, dtype=x.dtype)
                _, gw, _ = torch.ops.aten.convolution_backward(g, x, w_like, None, (1, 1), (pad, pad), (1, 1), False, [0, 0], 1, (False, True, False))
                gw = gw.to(weight.dtype)
        if has_bias and ctx.needs_input_grad[2]:
            if cout <= 256:
                gb = torch.empty(cout, dtype=torch.float32, device=g.device)
                ws = _ws(_ws_bytes("dd_channel_sum_workspace_bytes", cout), g.device)
                L.check(lib.dd_channel_sum_nhwc_t(_p(g), B * Ho * Wo, cout, _p(gb), code, _p(ws), stream), "dd_channel_sum_nhwc_t")
            else:
                gb = g.float().sum((0, 2, 3))
            gb = gb.to(bias_dtype)
        return gx, gw, gb, None


def half_conv(x, weight, bias=None, pad=1):
    return HalfConvFn.apply(x, weight, bias, int(pad))


class ReflectPad1NHWCFn(torch.autograd.Function):
    """nn.ReflectionPad2d(1) that keeps channels-last tensors channels-last (ATen returns NCHW and forces a layout copy)."""

    @staticmethod
    def forward(ctx, x):
        B, Cc, H, W = x.shape
        out = torch.empty((B, Cc, H + 2, W + 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        L.check(L.load().dd_reflect_pad1_nhwc_t(_p(x), B, H, W, Cc, _p(out), DTYPE_CODE[x.dtype], L.current_stream()), "dd_reflect_pad1_nhwc_t")
        ctx.dims = (B, Cc, H, W, x.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Cc, H, W, dtype = ctx.dims
        g = g.to(dtype).contiguous(memory_format=torch.channels_last)
        gx = torch.empty((B, Cc, H, W), dtype=dtype, device=g.device, memory_format=torch.channels_last)
        L.check(L.load().dd_reflect_pad1_nhwc_bwd_t(_p(g), B, H, W, Cc, _p(gx), DTYPE_CODE[dtype], L.current_stream()), "dd_reflect_pad1_nhwc_bwd_t")
        return gx


def reflect_pad1(x):
    """ReflectionPad2d(1); HIP kernel for fp32 / fp16 / bf16 channels-last GPU tensors with more than one channel, ATen otherwise."""
    if (x.is_cuda and x.dtype in DTYPE_CODE and x.dim() == 4 and x.shape[1] > 1 and x.shape[2] >= 4 and x.shape[3] >= 4
            and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()):
        return ReflectPad1NHWCFn.apply(x)
    return torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect")


class UpCatPadFn(torch.autograd.Function):
    """ReflectionPad2d(1)(cat(up(act(x)), skip)) as one pass (csrc/dd_decoder.hip): the glue between two 3x3 convolutions of the
    disparity decoders.  `x` is the ConvBlock's convolution output BEFORE its ELU when elu is set."""

    @staticmethod
    def forward(ctx, x, skip, mode, elu):
        B, C1, h, w = x.shape
        C2 = 0 if skip is None else skip.shape[1]
        H, W = (h, w) if mode == 2 else (2 * h, 2 * w)
        out = torch.empty((B, C1 + C2, H + 2, W + 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        L.check(L.load().dd_up_cat_pad_t(_p(x), None if skip is None else _p(skip), B, h, w, C1, C2, mode, int(elu), _p(out), DTYPE_CODE[x.dtype],
                                         L.current_stream()), "dd_up_cat_pad_t")
        ctx.save_for_backward(x)
        ctx.cfg = (C2, mode, int(elu))
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        C2, mode, elu = ctx.cfg
        B, C1, h, w = x.shape
        H, W = (h, w) if mode == 2 else (2 * h, 2 * w)
        g = g.to(x.dtype).contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gs = (torch.empty((B, C2, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
              if C2 and ctx.needs_input_grad[1] else None)
        if gx is not None or gs is not None:
            L.check(L.load().dd_up_cat_pad_bwd_t(_p(g), _p(x), B, h, w, C1, C2, mode, elu, None if gx is None else _p(gx), None if gs is None else _p(gs),
                                                 DTYPE_CODE[x.dtype], L.current_stream()), "dd_up_cat_pad_bwd_t")
        return gx, gs, None, None


def _nhwc(t):
    return t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous()


def up_cat_pad_ok(x, skip, mode):
    """Whether up_cat_pad() has its HIP path for these tensors (channels-last fp32 / fp16 / bf16 GPU tensors, channels in fours)."""
    import os
    if os.environ.get("DD_STOCK_DECODER_GLUE", "0") == "1":
        return False
    if not (x.is_cuda and x.dtype in DTYPE_CODE and x.dim() == 4 and x.shape[1] % 4 == 0 and x.shape[1] >= 4 and _nhwc(x)
            and x.shape[2] >= 2 and x.shape[3] >= 2 and (mode != 2 or (x.shape[2] >= 4 and x.shape[3] >= 4))):
        return False
    if skip is not None:
        H, W = (x.shape[2], x.shape[3]) if mode == 2 else (2 * x.shape[2], 2 * x.shape[3])
        if not (skip.dtype == x.dtype and skip.is_cuda and skip.shape[0] == x.shape[0] and skip.shape[1] % 4 == 0 and tuple(skip.shape[2:]) == (H, W)
                and _nhwc(skip)):
            return False
    return True


def up_cat_pad(x, skip=None, mode="bilinear", elu=True):
    """ReflectionPad2d(1)(cat((upsample(ELU(x), 2, mode), skip), 1)); mode in ("nearest", "bilinear", None = no up-sampling).
    One HIP pass where up_cat_pad_ok(), the reference's operator sequence otherwise."""
    code = {"nearest": 0, "bilinear": 1, None: 2}[mode]
    if up_cat_pad_ok(x, skip, code):
        return UpCatPadFn.apply(x, skip, code, bool(elu))
    F = torch.nn.functional
    y = F.elu(x) if elu else x
    if mode is not None:
        y = F.interpolate(y, scale_factor=2, mode=mode)
    if skip is not None:
        y = torch.cat((y, skip), 1)
    return reflect_pad1(y)


class DepthwiseConv3x3NHWCFn(torch.autograd.Function):
    """Depth-wise dilated 3x3 convolution (stride 1, padding == dilation, no bias) on channels-last fp32 / fp16 / bf16 tensors
    (reference networks/depth_encoder.py:168-181 CDilated with groups == channels).  The weight is the fp32 master copy in
    every case (nine taps per channel: nothing to gain from a half-precision copy), accumulation is fp32."""

    @staticmethod
    def forward(ctx, x, weight, dilation):
        B, Cc, H, W = x.shape
        w = weight.contiguous()
        out = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        L.check(L.load().dd_dwconv3x3_nhwc_t(_p(x), _p(w), B, H, W, Cc, dilation, _p(out), DTYPE_CODE[x.dtype], L.current_stream()), "dd_dwconv3x3_nhwc_t")
        ctx.save_for_backward(x, w)
        ctx.dilation = dilation
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        B, Cc, H, W = x.shape
        g = g.to(x.dtype).contiguous(memory_format=torch.channels_last)
        lib, gx, gw = L.load(), None, None
        code = DTYPE_CODE[x.dtype]
        if ctx.needs_input_grad[0]:
            gx = torch.empty((B, Cc, H, W), dtype=x.dtype, device=g.device, memory_format=torch.channels_last)
            L.check(lib.dd_dwconv3x3_nhwc_bwd_data_t(_p(g), _p(w), B, H, W, Cc, ctx.dilation, _p(gx), code, L.current_stream()), "dd_dwconv3x3_nhwc_bwd_data_t")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            nbytes = _ws_bytes("dd_dwconv3x3_workspace_bytes", B, H, Cc)
            ws = _ws(nbytes, g.device)
            L.check(lib.dd_dwconv3x3_nhwc_bwd_weight_t(_p(g), _p(x), B, H, W, Cc, ctx.dilation, _p(gw), _p(ws), nbytes, code, L.current_stream()),
                    "dd_dwconv3x3_nhwc_bwd_weight_t")
        return gx, gw, None


def depthwise_conv3x3(x, weight, dilation):
    """CDilated's convolution when groups == channels; the HIP kernels for fp32 / fp16 / bf16 channels-last GPU tensors (fp32
    weight), ATen otherwise."""
    Cc = x.shape[1]
    if (x.is_cuda and x.dtype in DTYPE_CODE and weight.dtype == torch.float32 and x.dim() == 4 and Cc % 4 == 0 and Cc <= 512
            and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()):
        return DepthwiseConv3x3NHWCFn.apply(x, weight, dilation)
    return torch.nn.functional.conv2d(x, weight, None, 1, dilation, dilation, Cc)


class PointwiseLinearFn(torch.autograd.Function):
    """nn.Linear over the channel axis of a channels-last (B,H,W,Cin) tensor (LiteMono's pwconv1/pwconv2, reference
    networks/depth_encoder.py:200-203). Forward and data gradient are the plain GEMMs; the weight gradient -- a (Cout x Cin)
    result reduced over B*H*W = 92160 rows, which the BLAS back-end runs at 17 TFLOP/s without split-K -- goes through MIOpen's
    1x1 weight-gradient implicit GEMM (5x faster here), and the bias gradient through the fixed-order HIP column sum.
    x, weight and bias arrive in ONE dtype (under autocast the caller hands over the half-precision casts)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, H, W, cin = x.shape
        ctx.save_for_backward(x, weight)
        with torch.autocast("cuda", enabled=False):
            return torch.addmm(bias, x.reshape(-1, cin), weight.t()).view(B, H, W, weight.shape[0])

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        B, H, W, cin = x.shape
        cout = weight.shape[0]
        g = g.to(x.dtype).contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.mm(g.view(-1, cout), weight).view(B, H, W, cin)
        if ctx.needs_input_grad[1]:
            _, gw, _ = torch.ops.aten.convolution_backward(g.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), weight.view(cout, cin, 1, 1), None,
                                                           [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
            gw = gw.reshape(cout, cin)
        if ctx.needs_input_grad[2]:
            lib = L.load()
            gb = torch.empty(cout, dtype=torch.float32, device=g.device)
            ws = _ws(_ws_bytes("dd_channel_sum_workspace_bytes", cout), g.device)
            L.check(lib.dd_channel_sum_nhwc_t(_p(g), B * H * W, cout, _p(gb), DTYPE_CODE[g.dtype], _p(ws), L.current_stream()), "dd_channel_sum_nhwc_t")
            gb = gb.to(weight.dtype)
        return gx, gw, gb


def pointwise_linear(x, layer):
    """layer(x) for an nn.Linear over the last axis of a contiguous (B,H,W,C) tensor.  Under autocast the GEMMs run in the
    autocast type on casts of the fp32 master weights (differentiable: the weight gradient returns to fp32 through the cast)."""
    if (x.is_cuda and x.dtype in DTYPE_CODE and layer.weight.dtype == torch.float32 and layer.bias is not None and x.dim() == 4
            and x.is_contiguous() and torch.is_grad_enabled() and layer.out_features <= 4096):
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            return PointwiseLinearFn.apply(x.to(dt), layer.weight.to(dt), layer.bias.to(dt))
        if x.dtype == torch.float32:
            return PointwiseLinearFn.apply(x, layer.weight, layer.bias)
    return layer(x.reshape(-1, x.shape[-1])).view(*x.shape[:-1], layer.out_features)


_MLP_CALLS = [0]


def mlp_calls():
    """How many times MlpFn.forward has launched dd_pw_gemm in this process (bench.py reports it)."""
    return _MLP_CALLS[0]


def mlp_ok(y, block):
    """dd_pw_gemm covers pwconv2(GELU(pwconv1(y))) of this block: fp32, channels a multiple of 32, erf GELU, no autocast.
    OPT-IN (DD_MLP=1): measured on MI355X (profiles/r05_mlp.txt) the BLAS kernels already run these HBM-bound GEMMs at 2.2-3.2 TB/s --
    the fp32 matrix-pipe rate is not what bounds them, so the bf16 split buys nothing here -- and the fused GELU prologue (76 us against
    44 + 46 at stage 1) does not make up for the slower wide GEMM (105 against 65 us): the block is 12 % slower forward, 23 % with the
    backward.  The path stays for its tests and as the starting point of a fused block kernel."""
    if os.environ.get("DD_MLP", "0") != "1" or os.environ.get("DD_STOCK_MLP", "0") == "1" or torch.is_autocast_enabled():
        return False
    l1, l2 = block.pwconv1, block.pwconv2
    # below ~16 k rows (LiteMono's 1/16-resolution stage) there are too few workgroups to fill the chip
    if y.dim() != 4 or y.shape[0] * y.shape[1] * y.shape[2] < int(os.environ.get("DD_MLP_MIN_ROWS", "16384")):
        return False
    return bool(y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and y.is_contiguous() and l1.weight.dtype == torch.float32
                and l1.bias is not None and l2.bias is not None and l1.in_features % 32 == 0 and l1.out_features % 32 == 0
                and l2.in_features == l1.out_features and l2.out_features == l1.in_features and y.shape[-1] == l1.in_features
                and isinstance(block.act, torch.nn.GELU) and getattr(block.act, "approximate", "none") == "none")


class MlpFn(torch.autograd.Function):
    """pwconv2(GELU(pwconv1(y))) of a LiteMono block (reference networks/depth_encoder.py:200-203,216-224,262-272) on a contiguous
    channels-last (B,H,W,C) tensor through dd_pw_gemm (csrc/dd_pw_gemm.hip): both Linears and both data gradients on the bf16 matrix
    pipe from three bf16 pieces per fp32 operand (fp32 accuracy, bit-reproducible); the GELU is applied to the second Linear's operand
    while it is split, so the activated 6C-wide tensor is neither written nor read in the forward.  The backward rebuilds it together
    with the activation's gradient in ONE pass (dd_gelu_pair); the weight gradients go through MIOpen's 1x1 weight-gradient implicit
    GEMM and the bias gradients through the fixed-order HIP column sum, as in PointwiseLinearFn."""

    @staticmethod
    def forward(ctx, y, w1, b1, w2, b2):
        lib = L.load()
        B, H, W, Cc = y.shape
        hid, M = w1.shape[0], B * H * W
        need = any(ctx.needs_input_grad)
        nb1, nb2 = _ws_bytes("dd_pw_gemm_pack_bytes", hid, Cc), _ws_bytes("dd_pw_gemm_pack_bytes", Cc, hid)
        # one buffer: fwd1 | fwd2 | (bwd2 | bwd1): the four packs of a block come out of ONE launch
        packs = torch.empty(((nb1 + nb2) * (2 if need else 1)) // 4, dtype=torch.float32, device=y.device)
        p0 = packs.data_ptr()
        stream = L.current_stream()
        L.check(lib.dd_mlp_pack(_p(w1), w1.stride(0), w1.stride(1), _p(w2), w2.stride(0), w2.stride(1), Cc, hid, p0, p0 + nb1,
                                p0 + nb1 + nb2 if need else None, p0 + 2 * nb1 + nb2 if need else None, None, stream), "dd_mlp_pack")
        pre = torch.empty((M, hid), dtype=torch.float32, device=y.device)
        L.check(lib.dd_pw_gemm(_p(y), p0, _p(b1), M, Cc, hid, 0, _p(pre), stream), "dd_pw_gemm (pwconv1)")
        out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=y.device)
        L.check(lib.dd_pw_gemm(_p(pre), p0 + nb1, _p(b2), M, hid, Cc, 1, _p(out), stream), "dd_pw_gemm (GELU, pwconv2)")
        _MLP_CALLS[0] += 1
        if need:
            ctx.save_for_backward(y, pre, w1, w2, packs)
        return out

    @staticmethod
    def backward(ctx, g):
        y, pre, w1, w2, packs = ctx.saved_tensors
        lib = L.load()
        B, H, W, Cc = y.shape
        hid, M = w1.shape[0], B * H * W
        nb1, nb2 = _ws_bytes("dd_pw_gemm_pack_bytes", hid, Cc), _ws_bytes("dd_pw_gemm_pack_bytes", Cc, hid)
        p0 = packs.data_ptr()
        stream = L.current_stream()
        g = g.to(torch.float32).contiguous()
        # g_post = g . w2 (M, 6C), then in one pass post = GELU(pre) and g_pre = g_post * GELU'(pre) in place
        gpre = torch.empty((M, hid), dtype=torch.float32, device=g.device)
        L.check(lib.dd_pw_gemm(_p(g), p0 + nb1 + nb2, None, M, Cc, hid, 0, _p(gpre), stream), "dd_pw_gemm (g . w2)")
        post = torch.empty((M, hid), dtype=torch.float32, device=g.device)
        L.check(lib.dd_gelu_pair(_p(pre), _p(gpre), _p(post), M * hid, stream), "dd_gelu_pair")
        gy = gw1 = gb1 = gw2 = gb2 = None
        if ctx.needs_input_grad[0]:
            gy = torch.empty((B, H, W, Cc), dtype=torch.float32, device=g.device)
            L.check(lib.dd_pw_gemm(_p(gpre), p0 + 2 * nb1 + nb2, None, M, hid, Cc, 0, _p(gy), stream), "dd_pw_gemm (g_pre . w1)")

        def wgrad(go, x, w):                # (M,cout), (M,cin) -> (cout,cin): MIOpen's 1x1 weight gradient on channels-last views
            cout, cin = w.shape
            _, gw, _ = torch.ops.aten.convolution_backward(go.view(B, H, W, cout).permute(0, 3, 1, 2), x.view(B, H, W, cin).permute(0, 3, 1, 2),
                                                           w.view(cout, cin, 1, 1), None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
            return gw.reshape(cout, cin)

        def colsum(go, cout):
            gb = torch.empty(cout, dtype=torch.float32, device=go.device)
            ws = _ws(_ws_bytes("dd_channel_sum_workspace_bytes", cout), go.device)
            L.check(lib.dd_channel_sum_nhwc_t(_p(go), M, cout, _p(gb), DTYPE_CODE[torch.float32], _p(ws), stream), "dd_channel_sum_nhwc_t")
            return gb

        if ctx.needs_input_grad[1]:
            gw1 = wgrad(gpre, y, w1)
        if ctx.needs_input_grad[2]:
            gb1 = colsum(gpre, hid)
        if ctx.needs_input_grad[3]:
            gw2 = wgrad(g, post, w2)
        if ctx.needs_input_grad[4]:
            gb2 = colsum(g, Cc)
        return gy, gw1, gb1, gw2, gb2


def mlp(y, block):
    return MlpFn.apply(y, block.pwconv1.weight, block.pwconv1.bias, block.pwconv2.weight, block.pwconv2.bias)


_MLP_FUSED_CALLS = [0]


def mlp_fused_calls():
    """How many block forwards dd_mlp_fwd has run in this process (bench.py reports it)."""
    return _MLP_FUSED_CALLS[0]


def mlp_fused_ok(y, block):
    """dd_mlp_fwd covers this block's pwconv2(GELU(pwconv1(y))) in ONE kernel: a pass that keeps nothing for a backward (the
    statistics-only side batch, evaluation), fp32, C = 64 or 128, erf GELU, enough rows to fill the chip."""
    if torch.is_grad_enabled() or os.environ.get("DD_STOCK_MLP_FUSED", "0") == "1" or torch.is_autocast_enabled():
        return False
    l1, l2 = block.pwconv1, block.pwconv2
    if not (y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and y.is_contiguous() and l1.weight.dtype == torch.float32):
        return False
    if y.shape[0] * y.shape[1] * y.shape[2] < int(os.environ.get("DD_MLP_FUSED_MIN_ROWS", "16384")):
        return False
    return bool(l1.bias is not None and l2.bias is not None and l1.out_features == 6 * l1.in_features and l2.in_features == l1.out_features
                and l2.out_features == l1.in_features and y.shape[-1] == l1.in_features and isinstance(block.act, torch.nn.GELU)
                and getattr(block.act, "approximate", "none") == "none" and L.load().dd_mlp_fwd_supported(l1.in_features))


def mlp_fused(y, block):
    """pwconv2(GELU(pwconv1(y))) through dd_mlp_fwd (csrc/dd_pw_gemm.hip: mlp_fwd_kernel): no tape, the hidden tensor never exists."""
    lib = L.load()
    w1, b1, w2, b2 = block.pwconv1.weight, block.pwconv1.bias, block.pwconv2.weight, block.pwconv2.bias
    B, H, W, Cc = y.shape
    hid, M = w1.shape[0], B * H * W
    nb1, nb2 = _ws_bytes("dd_pw_gemm_pack_bytes", hid, Cc), _ws_bytes("dd_pw_gemm_pack_bytes", Cc, hid)
    packs = torch.empty((nb1 + nb2) // 4, dtype=torch.float32, device=y.device)
    p0, stream = packs.data_ptr(), L.current_stream()
    L.check(lib.dd_mlp_pack(_p(w1), w1.stride(0), w1.stride(1), _p(w2), w2.stride(0), w2.stride(1), Cc, hid, p0, None, None, None, p0 + nb1, stream), "dd_mlp_pack")
    out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=y.device)
    L.check(lib.dd_mlp_fwd(_p(y), p0, p0 + nb1, _p(b1), _p(b2), M, Cc, _p(out), stream), "dd_mlp_fwd")
    _MLP_FUSED_CALLS[0] += 1
    return out


_MLP_RECOMPUTE_CALLS = [0]


def mlp_recompute_calls():
    """How many TRAINING block forwards MlpRecomputeFn has run through dd_mlp_fwd in this process (bench.py reports it)."""
    return _MLP_RECOMPUTE_CALLS[0]


def mlp_recompute_ok(y, block):
    """The training pass of this block takes MlpRecomputeFn: what mlp_fused_ok asks (fp32, no autocast, C = 64 or 128, erf GELU, enough
    rows) with the tape ON.  OPT-IN (DD_MLP_RECOMPUTE=1): measured NEUTRAL in the step on MI355X (350.96 / 350.41 img/s against 350.90 /
    350.63 with the two library Linears and ATen's GELU between them, alternating on one box: the traffic it saves -- the hidden tensor
    written and re-read three times in the forward -- was already hidden beside the other streams' kernels, and the backward pays one
    recomputed GEMM); what it does buy is memory: the two 6C-wide tensors per block (2 x 141 MB at stage 1, B = 12) are no longer held
    between forward and backward."""
    if not torch.is_grad_enabled() or os.environ.get("DD_MLP_RECOMPUTE", "0") != "1" or torch.is_autocast_enabled():
        return False
    l1, l2 = block.pwconv1, block.pwconv2
    if not (y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and y.is_contiguous() and l1.weight.dtype == torch.float32):
        return False
    if y.shape[0] * y.shape[1] * y.shape[2] < int(os.environ.get("DD_MLP_FUSED_MIN_ROWS", "16384")):
        return False
    return bool(l1.bias is not None and l2.bias is not None and l1.out_features == 6 * l1.in_features and l2.in_features == l1.out_features
                and l2.out_features == l1.in_features and y.shape[-1] == l1.in_features and isinstance(block.act, torch.nn.GELU)
                and getattr(block.act, "approximate", "none") == "none" and L.load().dd_mlp_fwd_supported(l1.in_features))


class MlpRecomputeFn(torch.autograd.Function):
    """pwconv2(GELU(pwconv1(y))) of a LiteMono block in a TRAINING pass (reference networks/depth_encoder.py:200-203,216-224,262-272) with
    the forward in ONE kernel (dd_mlp_fwd: the 6C-wide hidden tile stays on chip) and nothing but the block's INPUT kept for the backward,
    which rebuilds the pre-activation with one library GEMM and then runs round 5's backward: g . w2, dd_gelu_pair (GELU(pre) and
    g * GELU'(pre) in one pass), the data gradient, MIOpen's 1x1 weight gradients, the fixed-order column sums.  Against the two Linears
    with ATen's GELU between them the forward no longer writes and re-reads the hidden tensor three times (565 MB per block at stage 1,
    B = 12) and the backward trades ATen's two activation passes for one (-140 MB) and one recomputed GEMM (+141 MB written); the
    6C-wide tensors are no longer held between forward and backward (2 x 141 MB per block)."""

    @staticmethod
    def forward(ctx, y, w1, b1, w2, b2):
        lib = L.load()
        B, H, W, Cc = y.shape
        hid, M = w1.shape[0], B * H * W
        nb1, nb2 = _ws_bytes("dd_pw_gemm_pack_bytes", hid, Cc), _ws_bytes("dd_pw_gemm_pack_bytes", Cc, hid)
        packs = torch.empty((nb1 + nb2) // 4, dtype=torch.float32, device=y.device)
        p0, stream = packs.data_ptr(), L.current_stream()
        L.check(lib.dd_mlp_pack(_p(w1), w1.stride(0), w1.stride(1), _p(w2), w2.stride(0), w2.stride(1), Cc, hid, p0, None, None, None, p0 + nb1, stream), "dd_mlp_pack")
        out = torch.empty((B, H, W, Cc), dtype=torch.float32, device=y.device)
        L.check(lib.dd_mlp_fwd(_p(y), p0, p0 + nb1, _p(b1), _p(b2), M, Cc, _p(out), stream), "dd_mlp_fwd")
        _MLP_RECOMPUTE_CALLS[0] += 1
        ctx.save_for_backward(y, w1, b1, w2)
        return out

    @staticmethod
    def backward(ctx, g):
        y, w1, b1, w2 = ctx.saved_tensors
        lib = L.load()
        B, H, W, Cc = y.shape
        hid, M = w1.shape[0], B * H * W
        stream = L.current_stream()
        g2 = g.to(torch.float32).contiguous().view(M, Cc)
        y2 = y.view(M, Cc)
        with torch.autocast("cuda", enabled=False):
            pre = torch.addmm(b1, y2, w1.t())                  # the pre-activation again: one library GEMM
            gpre = torch.mm(g2, w2)                             # g . w2 (M, 6C)
        post = torch.empty_like(pre)
        L.check(lib.dd_gelu_pair(_p(pre), _p(gpre), _p(post), M * hid, stream), "dd_gelu_pair")       # post = GELU(pre), gpre *= GELU'(pre)
        gy = gw1 = gb1 = gw2 = gb2 = None
        if ctx.needs_input_grad[0]:
            with torch.autocast("cuda", enabled=False):
                gy = torch.mm(gpre, w1).view(B, H, W, Cc)

        def wgrad(go, x, w):                # (M,cout), (M,cin) -> (cout,cin): MIOpen's 1x1 weight gradient on channels-last views
            cout, cin = w.shape
            _, gw, _ = torch.ops.aten.convolution_backward(go.view(B, H, W, cout).permute(0, 3, 1, 2), x.view(B, H, W, cin).permute(0, 3, 1, 2),
                                                           w.view(cout, cin, 1, 1), None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False])
            return gw.reshape(cout, cin)

        def colsum(go, cout):
            gb = torch.empty(cout, dtype=torch.float32, device=go.device)
            ws = _ws(_ws_bytes("dd_channel_sum_workspace_bytes", cout), go.device)
            L.check(lib.dd_channel_sum_nhwc_t(_p(go), M, cout, _p(gb), DTYPE_CODE[torch.float32], _p(ws), stream), "dd_channel_sum_nhwc_t")
            return gb

        if ctx.needs_input_grad[1]:
            gw1 = wgrad(gpre, y2, w1)
        if ctx.needs_input_grad[2]:
            gb1 = colsum(gpre, hid)
        if ctx.needs_input_grad[3]:
            gw2 = wgrad(g2, post, w2)
        if ctx.needs_input_grad[4]:
            gb2 = colsum(g2, Cc)
        return gy, gw1, gb1, gw2, gb2


def mlp_recompute(y, block):
    return MlpRecomputeFn.apply(y, block.pwconv1.weight, block.pwconv1.bias, block.pwconv2.weight, block.pwconv2.bias)


BN_ACTS = {None: 0, "relu": 1, "gelu": 2}


class BatchNormActFn(torch.autograd.Function):
    """act(batch_norm(x) [+ residual]) in training mode on channels-last fp32 / fp16 / bf16 tensors (statistics and affine
    parameters fp32, like autocast's own batch_norm): two launches forward, two backward
    (stock: 3 + 3 MIOpen kernels plus one element-wise kernel per activation / residual add in each direction).
    groups > 1: the batch holds that many independent passes back to back (layers.batch_groups); every group is normalised
    with its own batch statistics and updates the running statistics in turn -- the same kernels on the group's slice of the
    buffers (a batch slice of a channels-last tensor is a contiguous range of rows), one autograd node, no slicing copies."""

    @staticmethod
    def forward(ctx, x, weight, bias, running, residual, momentum, eps, act, groups):
        B, Cc, H, W = x.shape
        rows = (B // groups) * H * W
        step = rows * Cc * x.element_size()                 # bytes between the groups' slices
        lib = L.load()
        out = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        stats = torch.empty(groups, 2, Cc, dtype=torch.float32, device=x.device)       # [group][mean | invstd][C]
        nbytes = _ws_bytes("dd_bn_workspace_bytes", Cc)
        ws = _ws(nbytes, x.device)
        code, stream = DTYPE_CODE[x.dtype], L.current_stream()
        xp, op, sp = x.data_ptr(), out.data_ptr(), stats.data_ptr()
        rp = residual.data_ptr() if residual is not None else None
        for g in range(groups):
            rm, rv = running[g] if running is not None else (None, None)
            L.check(lib.dd_bn_act_fwd_t(xp + g * step, rp + g * step if rp is not None else None, rows, Cc, _p(weight), _p(bias), eps, momentum,
                                        _p(rm), _p(rv), sp + g * 8 * Cc, sp + g * 8 * Cc + 4 * Cc, act, op + g * step, code, _p(ws), nbytes, stream),
                    "dd_bn_act_fwd_t")
        # the ReLU mask: from `out` behind a residual add; without one the backward recomputes it from x (one read pass less)
        ctx.save_for_backward(x, weight, bias, stats, out if (act == 1 and residual is not None) else None)
        ctx.conf = (act, residual is not None, rows, Cc, groups)
        return out

    @staticmethod
    def backward(ctx, g):
        x, weight, bias, stats, out = ctx.saved_tensors
        act, has_res, rows, Cc, groups = ctx.conf
        lib = L.load()
        g = g.to(x.dtype).contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x)
        want_res = has_res and ctx.needs_input_grad[4]
        gres = torch.empty_like(x) if (want_res and act != 0) else None
        gwb = torch.empty(groups, 2, Cc, dtype=torch.float32, device=g.device)          # [group][d weight | d bias][C]
        nbytes = _ws_bytes("dd_bn_workspace_bytes", Cc)
        ws = _ws(nbytes, g.device)
        code, stream = DTYPE_CODE[x.dtype], L.current_stream()
        step = rows * Cc * x.element_size()
        xp, gp, gxp, sp, wp = x.data_ptr(), g.data_ptr(), gx.data_ptr(), stats.data_ptr(), gwb.data_ptr()
        op = out.data_ptr() if out is not None else None
        grp = gres.data_ptr() if gres is not None else None
        for k in range(groups):
            L.check(lib.dd_bn_act_bwd_t(xp + k * step, gp + k * step, op + k * step if op is not None else None, rows, Cc, _p(weight), _p(bias),
                                        sp + k * 8 * Cc, sp + k * 8 * Cc + 4 * Cc, act, gxp + k * step, grp + k * step if grp is not None else None,
                                        wp + k * 8 * Cc, wp + k * 8 * Cc + 4 * Cc, code, _p(ws), nbytes, stream), "dd_bn_act_bwd_t")
        if want_res and act == 0:
            gres = g                                   # the add passes the gradient through unchanged
        if groups > 1:
            gwb = gwb.sum(0)                            # the passes share the affine parameters: their gradients add (fixed order)
        else:
            gwb = gwb[0]
        return gx, gwb[0], gwb[1], None, gres, None, None, None, None


def batch_norm_act(x, bn, act=None, residual=None, running=None, groups=1):
    """act(bn(x) [+ residual]) for a training-mode nn.BatchNorm2d `bn` with affine parameters and a momentum.
    `running`: (mean, var) buffers to update instead of the module's own (deferred statistics of a concurrent pass); with
    groups > 1 a list of such pairs, one per group, updated in order."""
    if running is None:
        pair = (bn.running_mean, bn.running_var) if bn.track_running_stats else None
        running = [pair] * groups if pair is not None else None
    elif groups == 1:
        running = [running]
    return BatchNormActFn.apply(x, bn.weight, bn.bias, running, residual, float(bn.momentum), float(bn.eps), BN_ACTS[act], groups)


class LayerNormFn(torch.autograd.Function):
    """F.layer_norm over the last axis of a contiguous fp32 / fp16 / bf16 (..., C) tensor (LiteMono's channels-last LayerNorm);
    fp32 affine parameters and statistics, the output in the input's type."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        y = torch.empty_like(x)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.check(L.load().dd_layer_norm_fwd_t(_p(x), rows, Cc, _p(weight), _p(bias), eps, _p(y), _p(mean), _p(rstd), DTYPE_CODE[x.dtype], L.current_stream()),
                "dd_layer_norm_fwd_t")
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, rstd = ctx.saved_tensors
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        lib = L.load()
        g = g.to(x.dtype).contiguous()
        gx = torch.empty_like(x)
        gwb = torch.empty(2 * Cc, dtype=torch.float32, device=g.device)
        nbytes = _ws_bytes("dd_layer_norm_workspace_bytes", Cc)
        ws = _ws(nbytes, g.device)
        L.check(lib.dd_layer_norm_bwd_t(_p(x), _p(g), _p(weight), _p(mean), _p(rstd), rows, Cc, _p(gx), _p(gwb), _p(ws), nbytes, DTYPE_CODE[x.dtype],
                                        L.current_stream()), "dd_layer_norm_bwd_t")
        return gx, gwb[:Cc], gwb[Cc:], None


def layer_norm_last(x, weight, bias, eps):
    """F.layer_norm(x, (C,), weight, bias, eps); the HIP kernels for contiguous fp32 / fp16 / bf16 GPU tensors with C % 4 == 0,
    C <= 256 (under autocast the stock operator would promote to fp32 and hand an fp32 tensor to the next GEMM's cast)."""
    Cc = x.shape[-1]
    if x.is_cuda and x.dtype in DTYPE_CODE and weight.dtype == torch.float32 and Cc % 4 == 0 and Cc <= 256 and x.is_contiguous():
        return LayerNormFn.apply(x, weight, bias, float(eps))      # also without a tape (the statistics-only passes, evaluation)
    return torch.nn.functional.layer_norm(x, (Cc,), weight, bias, eps)


class SplitQKVFn(torch.autograd.Function):
    """(B,N,3C) -> the views q, k, v (B,N,C) and qk = [q | k] (B,N,2C) of LiteMono's cross-covariance attention (reference
    networks/depth_encoder.py:83-86 indexes a permuted copy; networks.depth_encoder.XCA works on the buffer itself).  Autograd's
    own slices cost a zero-filled full-size buffer, a copy and an accumulating add PER slice in the backward (eleven launches over
    (12, 7680, 192) tensors per attention block); here the backward is one concatenation and one in-place add.  Same values: the
    stock path adds exact zeros."""

    @staticmethod
    def forward(ctx, qkv, C):
        ctx.C = C
        ctx.set_materialize_grads(False)
        return qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], qkv[..., :2 * C]

    @staticmethod
    def backward(ctx, gq, gk, gv, gqk):
        C = ctx.C
        like = next(g for g in (gq, gk, gv, gqk) if g is not None)
        shape = tuple(like.shape[:-1]) + (C,)
        parts = [g if g is not None else like.new_zeros(shape) for g in (gq, gk, gv)]
        out = torch.cat(parts, -1)
        if gqk is not None:
            out[..., :2 * C] += gqk
        return out, None


class SplitChannelsFn(torch.autograd.Function):
    """w (O, 2C...) -> dense copies of w[:, :C] and w[:, C:] (the two halves of the motion decoders' 1x1 reduction weights,
    networks/motion_decoder.redu_split); the backward is ONE concatenation where autograd's slices need two zero-fills, two copies
    and an add -- on tensors of a few hundred floats, twelve times per decoder pass."""

    @staticmethod
    def forward(ctx, w, C):
        ctx.C = C
        ctx.set_materialize_grads(False)
        return w[:, :C].contiguous(), w[:, C:].contiguous()

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        like = ga if ga is not None else gb
        if ga is None:
            ga = like.new_zeros((like.shape[0], ctx.C) + tuple(like.shape[2:]))
        if gb is None:
            gb = like.new_zeros(like.shape)          # (the halves are equal: 2C channels)
        return torch.cat((ga, gb), 1), None


class LayerScaleResidualFn(torch.autograd.Function):
    """res + y * scale with scale (B,1,1,C) fp32: LiteMono's layer scale x stochastic depth x residual (reference
    networks/depth_encoder.py:219-226).  fp32: one addcmul forward; half types: one HIP pass that multiplies in fp32 (the layer
    scale starts at 1e-6, below fp16's normal range).  The backward is one HIP pass (+ a fold) in every type."""

    @staticmethod
    def forward(ctx, res, y, scale):
        ctx.save_for_backward(y, scale)
        if y.dtype == torch.float32:
            return torch.addcmul(res, y, scale)
        B, H, W, Cc = y.shape
        res = res.to(y.dtype).contiguous()
        out = torch.empty_like(y)
        sc = scale.reshape(B, Cc).float().contiguous()
        L.check(L.load().dd_layer_scale_fwd_t(_p(res), _p(y), _p(sc), B, H * W, Cc, _p(out), DTYPE_CODE[y.dtype], L.current_stream()), "dd_layer_scale_fwd_t")
        return out

    @staticmethod
    def backward(ctx, g):
        y, scale = ctx.saved_tensors
        B, H, W, Cc = y.shape
        lib = L.load()
        g = g.to(y.dtype).contiguous()
        sc = scale.reshape(B, Cc).float().contiguous()
        gy = torch.empty_like(y)
        gs = torch.empty((B, Cc), dtype=torch.float32, device=g.device)
        nbytes = _ws_bytes("dd_layer_scale_workspace_bytes", B, Cc)
        ws = _ws(nbytes, g.device)
        L.check(lib.dd_layer_scale_bwd_t(_p(g), _p(y), _p(sc), B, H * W, Cc, _p(gy), _p(gs), _p(ws), nbytes, DTYPE_CODE[y.dtype], L.current_stream()),
                "dd_layer_scale_bwd_t")
        return (g if ctx.needs_input_grad[0] else None), gy, gs.view(B, 1, 1, Cc)


def layer_scale_residual(res, y, gamma, drop):
    """res + drop * gamma * y  (gamma (C,), drop (B,1,1,1) or None), all channels-last (B,H,W,C)."""
    B, Cc = y.shape[0], y.shape[-1]
    if (y.is_cuda and y.dtype in DTYPE_CODE and y.dim() == 4 and Cc % 4 == 0 and Cc <= 1024 and y.is_contiguous() and torch.is_grad_enabled()
            and (y.dtype != torch.float32 or res.dtype == torch.float32)):
        scale = (gamma.view(1, 1, 1, Cc) * drop.float() if drop is not None else gamma.view(1, 1, 1, Cc).expand(B, 1, 1, Cc))
        return LayerScaleResidualFn.apply(res, y, scale)
    return torch.addcmul(res, y, gamma if drop is None else gamma * drop)
