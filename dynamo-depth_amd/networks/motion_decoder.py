"""Coarse-to-fine residual motion field decoder (reference networks/motion_decoder.py:6-91).

A (B,out_dim,1,1) seed `Conv1x1(100*ego_motion)` is bilinearly up-sampled through the six encoder levels
(512,256,128,64,64 channels and the raw 9-channel input at full resolution); at each level two 3x3 convs
and a 1x1 reduction add a residual.  out_dim=3 -> complete_flow, out_dim=1 -> motion_prob / motion_mask.
"""
import os

import torch
import torch.nn as nn

from .layers import Conv2d, conv_cat_aligned, stock_slices
import torch.nn.functional as F


def redu_split(redu, a, b, C):
    from hipops.functions import ConvBiasFn, ReduFn, SplitChannelsFn, redu_ok
    if redu_ok(a, b, redu):
        return ReduFn.apply(a, b, redu.weight, redu.bias)          # the whole reduction as one operator: csrc/dd_redu.hip
    w = redu.weight
    if not stock_slices("redu"):
        wa, wb = SplitChannelsFn.apply(w, C)
    else:
        wa, wb = w[:, :C].contiguous(), w[:, C:].contiguous()
    from hipops.functions import small_conv, small_conv_ok
    if small_conv_ok(a, wa, redu.stride, redu.padding, redu.dilation, 1) and small_conv_ok(b, wb, redu.stride, redu.padding, redu.dilation, 1):
        return small_conv(a, wa, redu.bias) + small_conv(b, wb, None)
    ya = ConvBiasFn.apply(a, wa, redu.bias, redu.stride, redu.padding, redu.dilation, 1)
    return ya + F.conv2d(b, wb, None, redu.stride, redu.padding, redu.dilation, 1)


class MotionDecoder(nn.Module):
    def __init__(self, num_inp_feat, scales=4, num_input_images=2, inp_disp=True, out_dim=4):
        super().__init__()
        self.org_in_ch = num_input_images * (3 + int(inp_disp))
        self.num_inp_feat = [int(c) for c in num_inp_feat[::-1]] + [self.org_in_ch]
        self.out_dim = out_dim
        self.scales = scales
        assert max(self.scales) < len(self.num_inp_feat)
        self._residual_translation = Conv2d(6, out_dim, kernel_size=1)
        for level, ch in enumerate(self.num_inp_feat):
            setattr(self, "refine_motion_conv{}".format(level), nn.Sequential(
                Conv2d(ch + out_dim, ch, kernel_size=3, padding=1), Conv2d(ch, ch, kernel_size=3, padding=1)))
            setattr(self, "refine_motion_redu{}".format(level), Conv2d(2 * ch, out_dim, kernel_size=1))

    def forward(self, pose_feat, ego_motion):
        """pose_feat: [input (B,9,H,W), then encoder features fine -> coarse]; ego_motion (B,6,1,1)."""
        # Under autocast the FIELD stays fp32: it is a sum of residuals over six levels, and a half-precision accumulator rounds
        # the fine levels' small corrections away (bf16: 8 bits); the 3x3 convs that produce the residuals run in half precision.
        amp = ego_motion.is_cuda and torch.is_autocast_enabled()
        if amp:
            with torch.autocast("cuda", enabled=False):
                field = self._residual_translation(100 * ego_motion.float())
        else:
            field = self._residual_translation(100 * ego_motion)
        per_level = []
        for level in range(len(self.num_inp_feat)):
            feat = pose_feat[-1 - level]
            up = F.interpolate(field, size=feat.shape[-2:], mode="bilinear", align_corners=False)
            convs = getattr(self, "refine_motion_conv{}".format(level))
            a = conv_cat_aligned(convs[0], (up.to(feat.dtype) if amp else up, feat))
            b = convs[1](a)
            redu = getattr(self, "refine_motion_redu{}".format(level))
            if a.is_cuda and os.environ.get("DD_STOCK_REDU_CAT", "0") != "1":
                # redu(cat(a, b)) = W[:, :C] * a + W[:, C:] * b: two 1x1 convs on the tensors where they lie instead of a
                # 2C-channel concatenation (written, re-read, and sliced again -- with copies -- in the backward)
                C = a.shape[1]
                field = redu_split(redu, a, b, C)
            else:
                field = redu(torch.cat((a, b), 1))
            field = (field.float() if amp else field) + up
            per_level.append(field)
        outputs = {}
        for scale in self.scales:
            raw = 0.01 * per_level[len(self.num_inp_feat) - 1 - scale]
            if self.out_dim == 1:
                outputs[("motion_prob", scale)] = raw
                outputs[("motion_mask", scale)] = torch.sigmoid(raw)
            elif self.out_dim == 3:
                outputs[("complete_flow", scale)] = raw
            else:
                raise Exception("out_dim={} not excepted.".format(self.out_dim))
        return outputs
