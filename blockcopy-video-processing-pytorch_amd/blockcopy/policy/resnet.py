"""Small CIFAR-style ResNet used as the policy network's trunk (module names match the reference's
``policy/resnet.py`` so a state_dict of one loads into the other: conv1, bn1, layer{1,2,3}.N.{conv1,bn1,conv2,bn2,
downsample.{0,1}}, fc)."""
from __future__ import annotations

import math

import torch.nn as nn

BN_MOMENTUM = 0.02


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=False)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y += shortcut
        return self.relu(y)


class ResNet_32x32(nn.Module):
    """Three stages at strides 1/2/2 with widths (16,32,64)*width_factor; ``forward`` returns the stage-3 map."""

    def __init__(self, layers, num_classes=10, in_channels=3, width_factor=1):
        super().__init__()
        assert len(layers) == 3
        w = [int(16 * width_factor), int(32 * width_factor), int(64 * width_factor)]
        self.in_channels = in_channels
        self.inplanes = w[0]
        self.conv1 = _conv3x3(in_channels, w[0])
        self.bn1 = nn.BatchNorm2d(w[0], momentum=BN_MOMENTUM)
        self.relu = nn.ReLU(inplace=False)
        self.layer1 = self._make_layer(w[0], layers[0])
        self.layer2 = self._make_layer(w[1], layers[1], stride=2)
        self.layer3 = self._make_layer(w[2], layers[2], stride=2)
        self.avgpool = nn.AvgPool2d(8)
        self.fc = nn.Linear(64 * BasicBlock.expansion, num_classes)
        self.OUT_CHANNELS = w[2]
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * BasicBlock.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * BasicBlock.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * BasicBlock.expansion, momentum=BN_MOMENTUM))
        seq = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * BasicBlock.expansion
        seq += [BasicBlock(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = self.relu(self.bn1(self.conv1(x)))
        return self.layer3(self.layer2(self.layer1(x)))


def resnet8(pretrained=False, **kwargs):
    assert not pretrained, "no pretrained policy trunks are shipped; the policy is trained online"
    return ResNet_32x32([1, 1, 1], **kwargs)
