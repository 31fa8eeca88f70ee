"""ResNet feature pyramids for the depth (MD2), pose and motion encoders.

Same module tree / state_dict keys as the reference's torchvision-based encoder
(networks/resnet_encoder.py:95-135): `encoder.conv1`, `encoder.bn1`,
`encoder.layer{1..4}...`, and the never-used `encoder.fc`.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .resnet import ResNet

_PRETRAINED_DIR = os.environ.get("DYNAMO_PRETRAINED_DIR", "./ckpt")


def _load_imagenet(net, depth, channels_per_image, num_images):
    """ImageNet initialisation needs a local torchvision-format file (there is no network here):
    $DYNAMO_PRETRAINED_DIR/resnet<depth>.pth.  The first conv is tiled over the stacked frames and
    divided by their number, as the reference does (networks/resnet_encoder.py:82-90)."""
    path = os.path.join(_PRETRAINED_DIR, "resnet{}.pth".format(depth))
    if not os.path.isfile(path):
        raise FileNotFoundError(
            "weights_init='pretrained' needs {} (torchvision resnet{} state_dict); use --weights_init scratch "
            "when it is unavailable".format(path, depth))
    loaded = torch.load(path, map_location="cpu")
    if num_images > 1 or channels_per_image != 3:
        first = torch.nn.init.kaiming_normal_(torch.ones(64, channels_per_image * num_images, 7, 7))
        for i in range(num_images):
            first[:, channels_per_image * i:channels_per_image * i + 3] = loaded["conv1.weight"] / num_images
        loaded["conv1.weight"] = first
    net.load_state_dict(loaded)


class ResnetEncoder(nn.Module):
    def __init__(self, num_layers, pretrained, num_input_images=1, inp_disp=False):
        super().__init__()
        if num_input_images == 1 and inp_disp:
            raise AssertionError("single input image cannot be RGBD")
        if num_input_images > 1 and num_layers not in (18, 50):
            raise AssertionError("Can only run with 18 or 50 layer resnet")
        per_image = 4 if inp_disp else 3
        self.encoder = ResNet(num_layers, in_channels=per_image * num_input_images)
        self.num_ch_enc = np.array(self.encoder.out_channels)
        if pretrained:
            _load_imagenet(self.encoder, num_layers, per_image, num_input_images)

    def forward(self, input_image):
        e = self.encoder
        x = (input_image - 0.45) / 0.225
        feats = [e.bn1(e.conv1(x), act="relu")]
        feats.append(e.layer1(e.maxpool(feats[-1])))
        for stage in (e.layer2, e.layer3, e.layer4):
            feats.append(stage(feats[-1]))
        # (the reference also parks the list on the module, resnet_encoder.py:124-135; nothing reads it, and a module attribute
        # holding tensors with a tape keeps the whole autograd graph of the last forward -- activations included -- alive)
        return feats
