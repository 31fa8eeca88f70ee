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
        x = input_features[-1]
        for level in range(4, -1, -1):
            x = upsample(getattr(self, "upconv_{}_0".format(level))(x))
            if self.use_skips and level > 0:
                x = torch.cat((x, input_features[level - 1]), 1)
            x = getattr(self, "upconv_{}_1".format(level))(x)
            if level in self.scales:
                out[("disp", level)] = self.sigmoid(getattr(self, "dispconv_{}".format(level))(x))
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
        x = input_features[-1]
        for level in range(2, -1, -1):
            x = upsample(self.convs[("upconv", level, 0)](x), mode="bilinear")
            if self.use_skips and level > 0:
                x = torch.cat((x, input_features[level - 1]), 1)
            x = self.convs[("upconv", level, 1)](x)
            if level in self.scales:
                out[("disp", level)] = self.sigmoid(upsample(self.convs[("dispconv", level)](x), mode="bilinear"))
        return out           # (not parked on the module: that would keep the last forward's autograd graph alive)
