"""Pose math and the small conv blocks shared by the decoders.

Surface kept from the reference (networks/layers.py:7-121): transformation_from_parameters,
rot_from_axisangle, get_translation_matrix, ConvBlock, Conv3x3, upsample.  On the GPU the pose vector ->
4x4 conversion (about 40 micro-kernels in the reference) is one HIP kernel (dd_pose_matrix) with an
explicit backward; CPU tensors (unit tests, gloo runs) take the plain torch formulation below.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


def rot_from_axisangle(vec):
    """(B,1,3) axis-angle -> (B,4,4) rotation, axis = v/(|v|+1e-7) (reference networks/layers.py:43-82)."""
    angle = vec.norm(dim=2, keepdim=True)
    axis = vec / (angle + 1e-7)
    c, s = torch.cos(angle), torch.sin(angle)
    t = 1 - c
    x, y, z = axis.unbind(-1)                       # each (B,1)
    c, s, t = c.squeeze(-1), s.squeeze(-1), t.squeeze(-1)
    row0 = torch.cat([x * x * t + c, x * y * t - z * s, z * x * t + y * s], 1)
    row1 = torch.cat([x * y * t + z * s, y * y * t + c, y * z * t - x * s], 1)
    row2 = torch.cat([z * x * t - y * s, y * z * t + x * s, z * z * t + c], 1)
    rot = torch.zeros(vec.shape[0], 4, 4, dtype=vec.dtype, device=vec.device)
    rot[:, 0, :3], rot[:, 1, :3], rot[:, 2, :3] = row0, row1, row2
    rot[:, 3, 3] = 1
    return rot


def get_translation_matrix(translation_vector):
    """(B,1,3) -> (B,4,4) homogeneous translation (reference networks/layers.py:27-40)."""
    B = translation_vector.shape[0]
    T = torch.eye(4, dtype=translation_vector.dtype, device=translation_vector.device).repeat(B, 1, 1)
    T[:, :3, 3] = translation_vector.reshape(B, 3)
    return T


def transformation_from_parameters(axisangle, translation, invert=False):
    """Network (axisangle, translation) -> 4x4; invert=True gives R^T @ Trans(-t) (reference networks/layers.py:7-24)."""
    if axisangle.is_cuda:
        from hipops.functions import PoseMatrixFn
        # under autocast the pose head may hand over half precision; the pose matrix and the loss behind it are fp32
        return PoseMatrixFn.apply(axisangle.float(), translation.float(), bool(invert))
    R = rot_from_axisangle(axisangle)
    if invert:
        return torch.matmul(R.transpose(1, 2), get_translation_matrix(-translation))
    return torch.matmul(get_translation_matrix(translation), R)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict keys) whose bias gradient goes through the HIP channel-sum kernel on the
    GPU -- see hipops.functions.ConvBiasFn.  CPU tensors and bias-free convs take the stock path."""

    def forward(self, x):
        if x.is_cuda and self.padding_mode == "zeros":
            from hipops.functions import HeadConvFn, half_conv, half_conv_ok, head_conv_ok, mfma_conv, mfma_conv_ok, small_conv, small_conv_ok
            if x.dtype != torch.float32 and half_conv_ok(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
                return half_conv(x, self.weight, self.bias, self.padding[0])      # a half-precision network's 3x3, stride 1: csrc/dd_conv_half.hip
            if small_conv_ok(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
                return small_conv(x, self.weight, self.bias)          # a handful of channels at full resolution: csrc/dd_conv_small.hip
            if mfma_conv_ok(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
                return mfma_conv(x, self.weight, self.bias, self.padding[0])      # 16+ channels, 3x3, stride 1: csrc/dd_conv_mfma.hip
            if head_conv_ok(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
                return HeadConvFn.apply(x, self.weight, self.bias)    # a disparity head (C -> 1): csrc/dd_conv_head.hip
        if (self.bias is not None and x.is_cuda and self.padding_mode == "zeros" and torch.is_grad_enabled()
                and os.environ.get("DD_STOCK_CONV_BIAS_GRAD", "0") != "1"):
            from hipops.functions import ConvBiasFn
            return ConvBiasFn.apply(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        return super().forward(x)


_ZERO_CHANNELS = {}


def stock_slices(which):
    """DD_STOCK_SLICES=1 | a comma list of qkv, redu, eye: autograd's own slices / pads at those places (A/B switch)."""
    v = os.environ.get("DD_STOCK_SLICES", "0")
    return v == "1" or which in v.split(",")


def _as_channels_last(t):
    """t with channels-last strides; a one-channel tensor is restrided in place (both layouts are the same memory)."""
    if t.shape[1] == 1:
        B, _, H, W = t.shape
        return t.contiguous().as_strided(t.shape, (H * W, 1, W, 1))
    return t.contiguous(memory_format=torch.channels_last)


def conv_cat_aligned(conv, parts, force=False):
    """conv(torch.cat(parts, 1)) for an nn.Conv2d `conv`.  On the GPU the concatenated channel count is padded to a multiple of
    eight with zero channels (and the weight with zero input planes -- same result): MIOpen's NHWC fp32 implicit-GEMM kernels
    are 20-45 % slower forward on the 67 / 131 / 259-channel tensors that `cat(3-channel image or flow, features)` produces
    (reference networks/motion_decoder.py:66, networks/depth_encoder.py:408-424) than on 72 / 136 / 264 (scripts/probe_odd_channels.py)."""
    total = sum(p.shape[1] for p in parts)
    pad = (-total) % 8
    x0 = parts[0]
    if (pad == 0 or not (x0.is_cuda or force) or conv.groups != 1 or conv.padding_mode != "zeros" or total < 32
            or os.environ.get("DD_STOCK_CAT_CONV", "0") == "1"):
        if x0.is_cuda and total <= 16 and isinstance(conv, Conv2d) and os.environ.get("DD_STOCK_SMALL_CONV", "0") != "1":
            # the finest level of the motion decoders (dd_conv_small reads channels-last): concatenate in that layout right away
            parts = [_as_channels_last(p) for p in parts]
        return conv(torch.cat(list(parts), 1))
    big = max(parts, key=lambda p: p.shape[1])
    nhwc = big.is_contiguous(memory_format=torch.channels_last) and not big.is_contiguous()
    key = (x0.shape[0], pad, x0.shape[2], x0.shape[3], x0.dtype, str(x0.device), nhwc)
    zeros = _ZERO_CHANNELS.get(key)
    if zeros is None:
        zeros = torch.zeros((x0.shape[0], pad, x0.shape[2], x0.shape[3]), dtype=x0.dtype, device=x0.device)
        if key[-1]:
            zeros = zeros.contiguous(memory_format=torch.channels_last)
        _ZERO_CHANNELS[key] = zeros
    if key[-1]:
        # torch.cat returns an NCHW tensor as soon as one input is not channels-last (the 1- or 3-channel up-sampled field is
        # not), and the conv would then copy all 72 channels; restride the small tensors instead
        parts = [_as_channels_last(p) for p in parts]
    x = torch.cat(list(parts) + [zeros], 1)
    w = F.pad(conv.weight, (0, 0, 0, 0, 0, pad))
    if x.is_cuda:
        from hipops.functions import half_conv, half_conv_ok, mfma_conv, mfma_conv_ok
        if x.dtype != torch.float32 and half_conv_ok(x, w, conv.stride, conv.padding, conv.dilation, conv.groups):
            return half_conv(x, w, conv.bias, conv.padding[0])
        if mfma_conv_ok(x, w, conv.stride, conv.padding, conv.dilation, conv.groups):
            return mfma_conv(x, w, conv.bias, conv.padding[0])
    if conv.bias is not None and x.is_cuda and torch.is_grad_enabled() and os.environ.get("DD_STOCK_CONV_BIAS_GRAD", "0") != "1":
        from hipops.functions import ConvBiasFn
        return ConvBiasFn.apply(x, w, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)
    return F.conv2d(x, w, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)


class DeferredStats:
    """Running-statistics updates of ONE forward pass, kept aside.  networks.Model runs the depth net on its three frames
    concurrently (--multi_stream); the three passes update the same BatchNorm buffers, and `running = (1-m)*running + m*stat`
    does not commute.  The two statistics-only passes therefore write their m*stat terms into zero-initialised scratch
    (same kernels, the scratch stands in for the buffers) and `apply()` folds them into the real buffers afterwards, in the
    reference's frame order, with two multi-tensor launches per pass."""

    def __init__(self, device, floats):
        self.buf = torch.zeros(floats, dtype=torch.float32, device=device)
        self.used = 0
        self.pairs = []

    def take(self, bn):
        C = bn.num_features
        if self.used + 2 * C > self.buf.numel():
            raise RuntimeError("DeferredStats scratch too small")
        mean, var = self.buf[self.used:self.used + C], self.buf[self.used + C:self.used + 2 * C]
        self.used += 2 * C
        self.pairs.append((bn, mean, var))
        return mean, var

    def apply(self):
        if not torch.cuda.is_current_stream_capturing():          # (a captured step keeps the scratch for good)
            self.buf.record_stream(torch.cuda.current_stream())      # filled on the pass's stream, consumed (and released) on this one
        groups = {}
        for bn, mean, var in self.pairs:
            g = groups.setdefault(float(bn.momentum), ([], []))
            g[0].extend((bn.running_mean, bn.running_var))
            g[1].extend((mean, var))
        for momentum, (running, terms) in groups.items():
            # through .data: the stock (NCHW) batch-norm node of the target-frame pass has saved these buffers for its backward
            # (it never reads them in training mode); an in-place update through the buffer itself would trip autograd's
            # version check there
            running = [r.data for r in running]
            torch._foreach_mul_(running, 1.0 - momentum)
            torch._foreach_add_(running, terms)


_DEFER = None      # the DeferredStats collector of the pass being issued (one Python thread issues all passes); a list of them
                   # (one per group) while a grouped pass is being issued
_GROUPS = 1        # > 1: the batch being pushed through holds that many independent passes back to back (see batch_groups)


class batch_groups:
    """The forward passes issued inside hold `groups` independent batches concatenated along the batch axis (networks.Model
    sends the two statistics-only depth passes through the net as one batch of 2B: half the launches and half the host work
    for everything that treats samples independently).  BatchNorm is the one layer that does not: under this context it takes
    its batch statistics per group and updates the running statistics group by group, i.e. exactly what the separate passes
    would have done.  `collectors`: one DeferredStats per group, or None to update the module's buffers directly (in order)."""

    def __init__(self, groups, collectors=None):
        self.groups, self.collectors = int(groups), collectors

    def __enter__(self):
        global _GROUPS, _DEFER
        self.prev = (_GROUPS, _DEFER)
        _GROUPS = self.groups
        if self.collectors is not None:
            _DEFER = list(self.collectors)
        return self

    def __exit__(self, *exc):
        global _GROUPS, _DEFER
        _GROUPS, _DEFER = self.prev


class defer_running_stats:
    def __init__(self, collector):
        self.collector = collector

    def __enter__(self):
        global _DEFER
        self.prev, _DEFER = _DEFER, self.collector
        return self.collector

    def __exit__(self, *exc):
        global _DEFER
        _DEFER = self.prev


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same keys, same arithmetic) whose `num_batches_tracked` counter is kept on the host between
    checkpoints. With a momentum set -- every BN of the reference -- the counter never enters the arithmetic, yet stock
    PyTorch increments the device scalar with one kernel per layer per forward (114 launches of ~4.5 us per training step)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._pending_batches = 0

    def forward(self, x, act=None, residual=None):
        """act(bn(x) [+ residual]); act in (None, 'relu', 'gelu').  The activation and the residual add are arguments so that
        the channels-last training path can run them inside the normalisation kernel (hipops.functions.BatchNormActFn)."""
        if self.training and self.track_running_stats and self.momentum is not None and _GROUPS > 1:
            # independent passes back to back along the batch axis (batch_groups): statistics and running-statistics updates
            # per group, in order -- what the separate passes would have done; tape-free passes only
            assert x.shape[0] % _GROUPS == 0
            self._check_input_dim(x)
            self.__dict__["_pending_batches"] += _GROUPS      # plain attribute: nn.Module.__setattr__ costs ~2 us per forward
            per = x.shape[0] // _GROUPS
            bufs = [(self.running_mean, self.running_var) if _DEFER is None else _DEFER[gi].take(self) for gi in range(_GROUPS)]
            if self._hip_path(x, act, residual):
                from hipops.functions import batch_norm_act
                return batch_norm_act(x, self, act, residual, running=bufs, groups=_GROUPS)
            y = torch.cat([F.batch_norm(x[gi * per:(gi + 1) * per], bufs[gi][0], bufs[gi][1], self.weight, self.bias, True, self.momentum, self.eps)
                           for gi in range(_GROUPS)])
        elif self.training and self.track_running_stats and self.momentum is not None:
            self._check_input_dim(x)
            self.__dict__["_pending_batches"] += 1
            rm, rv = (self.running_mean, self.running_var) if _DEFER is None else _DEFER.take(self)
            if self._hip_path(x, act, residual):
                from hipops.functions import batch_norm_act
                return batch_norm_act(x, self, act, residual, running=(rm, rv))
            y = F.batch_norm(x, rm, rv, self.weight, self.bias, True, self.momentum, self.eps)
        else:
            y = super().forward(x)
        if residual is not None:
            y = y + residual
        if act == "relu":
            return F.relu(y)
        if act == "gelu":
            return F.gelu(y)
        assert act is None, act
        return y

    def _hip_path(self, x, act, residual):
        Cc = x.shape[1]
        ok = (x.is_cuda and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and self.affine and self.weight.dtype == torch.float32
              and x.dim() == 4 and Cc % 4 == 0 and Cc <= 512
              and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
              and not (act == "gelu" and residual is not None) and os.environ.get("DD_STOCK_BATCHNORM", "0") != "1")
        if ok and residual is not None:
            ok = (residual.shape == x.shape and residual.dtype == x.dtype
                  and residual.is_contiguous(memory_format=torch.channels_last))
        return ok

    def flush_counter(self):
        if self._pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked += self._pending_batches
        self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self.flush_counter()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        super()._load_from_state_dict(*args, **kwargs)


class Conv3x3(nn.Module):
    """Reflection- (or zero-) padded 3x3 convolution; keys `conv.{weight,bias}`."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.use_refl = bool(use_refl)
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x):
        if self.use_refl and x.is_cuda and os.environ.get("DD_STOCK_REFLECT_PAD", "0") != "1":
            from hipops.functions import reflect_pad1
            return self.conv(reflect_pad1(x))
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    """Conv3x3 + ELU; keys `conv.conv.{weight,bias}`."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        return self.nonlin(self.conv(x))


def upsample(x, scale_factor=2, mode="nearest"):
    return F.interpolate(x, scale_factor=scale_factor, mode=mode)
