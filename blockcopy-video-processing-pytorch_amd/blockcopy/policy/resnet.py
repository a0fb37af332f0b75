"""Trunk of the policy network: a three-stage residual CNN (CIFAR layout: no stem pooling, strides 1/2/2).

Parameter names are an interface: a ``state_dict`` of the reference's ``policy/resnet.py`` loads unchanged
(``conv1, bn1, layer{1,2,3}.<i>.{conv1,bn1,conv2,bn2,downsample.{0,1}}, fc``).  ``forward`` returns the stage-3 feature
map (stride 4 w.r.t. the policy input), which is what ``PolicyNet`` consumes; ``avgpool``/``fc`` exist only so that
classification checkpoints load.
"""
from __future__ import annotations

import math
from typing import Sequence

import torch.nn as nn

from blockcopy.policy.fused_bn import PolicyBatchNorm2d

BN_MOMENTUM = 0.02          # slow running statistics: the policy trains online on single frames
STAGE_WIDTHS = (16, 32, 64)  # x width_factor
STAGE_STRIDES = (1, 2, 2)


def _bn(channels: int) -> nn.BatchNorm2d:
    return PolicyBatchNorm2d(channels, momentum=BN_MOMENTUM)     # nn.BatchNorm2d with a two-launch training forward (fused_bn.py)


class BasicBlock(nn.Module):
    """y = relu(bn2(conv2(relu(bn1(conv1(x))))) + shortcut(x)), both convs 3x3."""

    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: nn.Module = None):
        super().__init__()
        self.stride = stride
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = _bn(planes)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample

    def forward(self, x):
        y = self.bn2(self.conv2(self.bn1(self.conv1(x), relu=True)))      # (bn1 + self.relu in one pass)
        y += x if self.downsample is None else self.downsample(x)
        return self.relu(y)


class ResNet_32x32(nn.Module):
    def __init__(self, layers: Sequence[int], num_classes: int = 10, in_channels: int = 3, width_factor: float = 1):
        super().__init__()
        assert len(layers) == len(STAGE_WIDTHS)
        widths = [int(w * width_factor) for w in STAGE_WIDTHS]
        self.in_channels = in_channels
        self.conv1 = nn.Conv2d(in_channels, widths[0], 3, 1, 1, bias=False)
        self.bn1 = _bn(widths[0])
        self.relu = nn.ReLU(inplace=False)
        cin = widths[0]
        for i, (depth, width, stride) in enumerate(zip(layers, widths, STAGE_STRIDES), start=1):
            setattr(self, f"layer{i}", self._stage(cin, width, depth, stride))
            cin = width * BasicBlock.expansion
        self.OUT_CHANNELS = cin
        self.avgpool = nn.AvgPool2d(8)
        self.fc = nn.Linear(STAGE_WIDTHS[-1] * BasicBlock.expansion, num_classes)
        self._init_weights()

    @staticmethod
    def _stage(cin: int, width: int, depth: int, stride: int) -> nn.Sequential:
        cout = width * BasicBlock.expansion
        shortcut = None
        if stride != 1 or cin != cout:   # projection shortcut: 1x1 conv + BN
            shortcut = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), _bn(cout))
        blocks = [BasicBlock(cin, width, stride, shortcut)] + [BasicBlock(cout, width) for _ in range(depth - 1)]
        return nn.Sequential(*blocks)

    def _init_weights(self) -> None:
        for m in self.modules():
            if isinstance(m, nn.Conv2d):      # He initialisation on fan-out
                fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1]
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.bn1(self.conv1(x), relu=True)
        return self.layer3(self.layer2(self.layer1(x)))


def resnet8(pretrained: bool = False, **kwargs) -> ResNet_32x32:
    """One block per stage (8 weight layers counting the classifier)."""
    if pretrained:
        raise ValueError("no pretrained policy trunks are shipped; the policy is trained online")
    return ResNet_32x32([1, 1, 1], **kwargs)
