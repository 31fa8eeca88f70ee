"""Pose head: 1x1 squeeze, two 3x3 convs, 1x1 to 6*frames, global mean, x0.01
(reference networks/pose_decoder.py:5-44).  The four convs are also registered through the `net`
ModuleList, so every parameter appears twice in the state_dict (`squeeze.*` and `net.0.*`, ...)."""
import torch
import torch.nn as nn

from .layers import Conv2d


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_input_features = num_input_features
        self.num_frames_to_predict_for = (num_input_features - 1) if num_frames_to_predict_for is None else num_frames_to_predict_for
        self.squeeze = Conv2d(int(num_ch_enc[-1]), 256, 1)
        self.pose0 = Conv2d(num_input_features * 256, 256, 3, stride, 1)
        self.pose1 = Conv2d(256, 256, 3, stride, 1)
        self.pose2 = Conv2d(256, 6 * self.num_frames_to_predict_for, 1)
        self.net = nn.ModuleList([self.squeeze, self.pose0, self.pose1, self.pose2])
        self.relu = nn.ReLU()

    def forward(self, input_features):
        if torch.is_autocast_enabled() and input_features[0][-1].is_cuda:
            # reduced-precision networks (--amp): the pose head stays fp32.  Its input is the coarsest feature map (1/32
            # resolution, a few hundred KB), its output six numbers per frame of size ~1e-3 whose gradient is a sum over all
            # pixels with heavy cancellation -- the one place where 8-11 mantissa bits change the training signal, for no
            # measurable time
            with torch.autocast("cuda", enabled=False):
                return self._forward([[f[-1].float()] for f in input_features])
        return self._forward(input_features)

    def _forward(self, input_features):
        x = torch.cat([self.relu(self.squeeze(f[-1])) for f in input_features], 1)
        x = self.relu(self.pose0(x))
        x = self.relu(self.pose1(x))
        x = self.pose2(x).mean(3).mean(2)
        x = 0.01 * x.view(-1, self.num_frames_to_predict_for, 1, 6)
        return x[..., :3], x[..., 3:]
