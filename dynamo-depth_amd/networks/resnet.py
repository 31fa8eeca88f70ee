"""Residual-network trunk with the public torchvision layout.

torchvision is not a dependency of this tree (it is absent from the MI355X image), but the
reference encoders subclass ``torchvision.models.ResNet`` (reference networks/resnet_encoder.py:4,8,
103-107) and its checkpoints therefore carry torchvision's ``state_dict`` keys
(``conv1, bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2[,conv3,bn3],downsample.{0,1}}, fc``).
This file restates that topology so those checkpoints load unchanged.
The 3x3 stride-1 convolutions run through dd_conv3x3_mfma on the GPU (layers.Conv2d), the others through MIOpen via PyTorch-ROCm.
"""
import torch
import torch.nn as nn

try:
    from .layers import BatchNorm2d, Conv2d
except ImportError:          # loaded by file path as the torchvision stand-in of tests/golden/_refshim.py
    Conv2d = nn.Conv2d

    class BatchNorm2d(nn.BatchNorm2d):
        def forward(self, x, act=None, residual=None):
            y = super().forward(x)
            y = y if residual is None else y + residual
            return torch.relu(y) if act == "relu" else y

# depth -> (block kind, blocks per stage)
_SPECS = {
    18: ("basic", (2, 2, 2, 2)),
    34: ("basic", (3, 4, 6, 3)),
    50: ("bottleneck", (3, 4, 6, 3)),
    101: ("bottleneck", (3, 4, 23, 3)),
    152: ("bottleneck", (3, 8, 36, 3)),
}


def _conv(cin, cout, k, stride=1):
    # layers.Conv2d = nn.Conv2d (same keys) whose 3x3 stride-1 instances on 16+ channels run through dd_conv3x3_mfma on the GPU
    return Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, width, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(cin, width, 3, stride)
        self.bn1 = BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(width, width, 3)
        self.bn2 = BatchNorm2d(width)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.bn1(self.conv1(x), act="relu")
        return self.bn2(self.conv2(y), act="relu", residual=skip)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, width, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(cin, width, 1)
        self.bn1 = BatchNorm2d(width)
        self.conv2 = _conv(width, width, 3, stride)
        self.bn2 = BatchNorm2d(width)
        self.conv3 = _conv(width, width * 4, 1)
        self.bn3 = BatchNorm2d(width * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.bn1(self.conv1(x), act="relu")
        y = self.bn2(self.conv2(y), act="relu")
        return self.bn3(self.conv3(y), act="relu", residual=skip)


_BLOCKS = {"basic": BasicBlock, "bottleneck": Bottleneck}


class ResNet(nn.Module):
    """conv1/bn1/relu/maxpool/layer1..4/avgpool/fc, any number of input channels.

    ``fc``/``avgpool`` are never used by the encoders (reference networks/resnet_encoder.py:124-135)
    but are kept so the key set equals the reference checkpoints' (SURVEY.md Appendix D/E).
    """

    def __init__(self, depth=18, in_channels=3, num_classes=1000):
        super().__init__()
        if depth not in _SPECS:
            raise ValueError("{} is not a valid number of resnet layers".format(depth))
        kind, counts = _SPECS[depth]
        block = _BLOCKS[kind]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(block, 64, counts[0], 1)
        self.layer2 = self._stage(block, 128, counts[1], 2)
        self.layer3 = self._stage(block, 256, counts[2], 2)
        self.layer4 = self._stage(block, 512, counts[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        self.out_channels = [64] + [w * block.expansion for w in (64, 128, 256, 512)]
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _stage(self, block, width, count, stride):
        down = None
        if stride != 1 or self.inplanes != width * block.expansion:
            down = nn.Sequential(
                nn.Conv2d(self.inplanes, width * block.expansion, kernel_size=1, stride=stride, bias=False),
                BatchNorm2d(width * block.expansion),
            )
        blocks = [block(self.inplanes, width, stride, down)]
        self.inplanes = width * block.expansion
        blocks += [block(self.inplanes, width) for _ in range(count - 1)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.maxpool(self.bn1(self.conv1(x), act="relu"))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))
