"""Disparity decoders (reference networks/depth_decoder.py:10-115).

DepthDecoder      Monodepth2 U-Net: nearest x2 up-sampling, skip connections, sigmoid heads `dispconv_{s}`.
LiteDepthDecoder  Lite-Mono: bilinear x2, three levels, each head additionally x2 up-sampled.  Only the
                  `decoder` ModuleList is registered (keys `decoder.{0..8}...`), in the order
                  upconv(2,0),(2,1),(1,0),(1,1),(0,0),(0,1), dispconv per scale.
"""
import numpy as np
import torch
import torch.nn as nn

from .layers import ConvBlock, Conv3x3, upsample


def _glue(x, skip=None, mode=None, elu=True):
    """ReflectionPad2d(1)(cat((upsample(ELU(x)), skip), 1)): what sits between two 3x3 convolutions of a decoder (the ELU of one
    ConvBlock, the up-sampling, the skip concatenation, the padding inside the next Conv3x3).  One HIP pass on the GPU
    (hipops.functions.up_cat_pad), the reference's operator sequence elsewhere -- same values either way."""
    from hipops.functions import up_cat_pad
    return up_cat_pad(x, skip, mode, elu)


def _head(conv, padded):
    """A disparity head (C -> 1 channel).  Under autocast it runs in fp32 on the up-cast features: the sigmoid's argument rounded
    to bf16 (8 bits) quantises the disparity -- and with it every depth the view synthesis warps by -- into visible steps; the
    pose gradient, a heavily cancelling sum over all pixels, was 5.8x the fp32 one at 192x640 with half-precision heads."""
    if padded.is_cuda and torch.is_autocast_enabled():
        with torch.autocast("cuda", enabled=False):
            return conv(padded.float())
    return conv(padded)


def _conv(block):
    """The bare nn.Conv2d of a ConvBlock / Conv3x3 (its padding -- and a ConvBlock's ELU -- are applied by _glue)."""
    c3 = block.conv if isinstance(block, ConvBlock) else block
    assert c3.use_refl
    return c3.conv


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels, self.use_skips, self.scales = num_output_channels, use_skips, scales
        self.upsample_mode = "nearest"
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        for level in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if level == 4 else self.num_ch_dec[level + 1]
            setattr(self, "upconv_{}_0".format(level), ConvBlock(cin, self.num_ch_dec[level]))
            cin = self.num_ch_dec[level] + (self.num_ch_enc[level - 1] if (use_skips and level > 0) else 0)
            setattr(self, "upconv_{}_1".format(level), ConvBlock(cin, self.num_ch_dec[level]))
        for s in self.scales:
            setattr(self, "dispconv_{}".format(s), Conv3x3(self.num_ch_dec[s], num_output_channels))
        self.sigmoid = nn.Sigmoid()

    def forward(self, input_features):
        out = {}
        # `pre`: a ConvBlock's convolution output BEFORE its ELU; the ELU is applied by whoever reads it (_glue)
        pre = _conv(self.upconv_4_0)(_glue(input_features[-1], elu=False))
        for level in range(4, -1, -1):
            if level < 4:
                pre = _conv(getattr(self, "upconv_{}_0".format(level)))(_glue(pre))
            skip = input_features[level - 1] if self.use_skips and level > 0 else None
            pre = _conv(getattr(self, "upconv_{}_1".format(level)))(_glue(pre, skip, "nearest"))
            if level in self.scales:
                out[("disp", level)] = self.sigmoid(_head(_conv(getattr(self, "dispconv_{}".format(level))), _glue(pre)))
        return out           # (not parked on the module: that would keep the last forward's autograd graph alive)


class LiteDepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels, self.use_skips, self.scales = num_output_channels, use_skips, scales
        self.upsample_mode = "bilinear"
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = (self.num_ch_enc / 2).astype("int")
        self.convs = {}
        for level in range(2, -1, -1):
            cin = self.num_ch_enc[-1] if level == 2 else self.num_ch_dec[level + 1]
            self.convs[("upconv", level, 0)] = ConvBlock(cin, self.num_ch_dec[level])
            cin = self.num_ch_dec[level] + (self.num_ch_enc[level - 1] if (use_skips and level > 0) else 0)
            self.convs[("upconv", level, 1)] = ConvBlock(cin, self.num_ch_dec[level])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, input_features):
        out = {}
        pre = _conv(self.convs[("upconv", 2, 0)])(_glue(input_features[-1], elu=False))
        for level in range(2, -1, -1):
            if level < 2:
                pre = _conv(self.convs[("upconv", level, 0)])(_glue(pre))
            skip = input_features[level - 1] if self.use_skips and level > 0 else None
            pre = _conv(self.convs[("upconv", level, 1)])(_glue(pre, skip, "bilinear"))
            if level in self.scales:
                out[("disp", level)] = self.sigmoid(upsample(_head(_conv(self.convs[("dispconv", level)]), _glue(pre)), mode="bilinear"))
        return out           # (not parked on the module: that would keep the last forward's autograd graph alive)
