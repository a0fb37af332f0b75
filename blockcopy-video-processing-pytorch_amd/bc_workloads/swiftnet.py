"""SwiftNet (ResNet-18/34/50 encoder + spatial pyramid pooling + 3-stage ladder decoder) as the block-copy workload.

Own restatement of the architecture the reference evaluates (semantic_segmentation/lib/models/swiftnet/{swiftnet,util}.py,
backbones/resnet.py): identical parameter names and shapes (a reference checkpoint / state_dict loads with strict=True;
tests/golden/swiftnet_keys.json pins that) and identical op order per frame, which is what the
block engine's per-layer ring caches key on.  The three touch-points with the engine are the same as in the reference:
the SPP runs densely (``@blockcopy_noblocks``), everything else runs on packed tiles, and nothing else knows about tiles.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from blockcopy import blockcopy_noblocks
from blockcopy.core.tensorwrapper import TensorWrapper
from blockcopy.utils.profiler import timings


# ----------------------------------------------------------------------------------------------- encoder
def _c3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


def _c1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, 1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c3(cin, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _c3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y += x if self.downsample is None else self.downsample(x)
        return self.relu(y)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c1(cin, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _c3(planes, planes, stride)   # stride on the 3x3 (ResNet v1.5)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _c1(planes, planes * self.expansion)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y += x if self.downsample is None else self.downsample(x)
        return self.relu(y)


class ResNetEncoder(nn.Module):
    """ImageNet-style ResNet trunk without classifier; ``forward_down`` returns the four stage outputs."""

    def __init__(self, block, layers):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.block_features = []
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            setattr(self, f"layer{i + 1}", self._stage(block, planes, n, stride=1 if i == 0 else 2))
            self.block_features.append(self.inplanes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _stage(self, block, planes, n, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(_c1(self.inplanes, planes * block.expansion, stride), nn.BatchNorm2d(planes * block.expansion))
        blocks = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        blocks += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def forward_down(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        feats = []
        for i in range(1, 5):
            x = getattr(self, f"layer{i}")(x)
            feats.append(x)
        return feats


def resnet18():
    return ResNetEncoder(BasicBlock, [2, 2, 2, 2])


def resnet34():
    return ResNetEncoder(BasicBlock, [3, 4, 6, 3])


def resnet50():
    return ResNetEncoder(Bottleneck, [3, 4, 6, 3])


BACKBONES = {"resnet18": resnet18, "resnet34": resnet34, "resnet50": resnet50}


# ----------------------------------------------------------------------------------------------- decoder pieces
class PreActConv(nn.Sequential):
    """BN -> ReLU -> conv (children named norm / relu / conv: the BN comes first, so BN folding leaves it alone)."""

    def __init__(self, cin, cout, k=3, batch_norm=True, bn_momentum=0.9, bias=False):
        super().__init__()
        if batch_norm:
            self.add_module("norm", nn.BatchNorm2d(cin, momentum=bn_momentum))
        self.add_module("relu", nn.ReLU(inplace=False))
        self.add_module("conv", nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2, bias=bias))


class LadderUp(nn.Module):
    """x2 bilinear upsample of the coarse map, add the 1x1-projected skip, 3x3 blend."""

    def __init__(self, num_maps_in, skip_maps_in, num_maps_out, k=3):
        super().__init__()
        self.bottleneck = PreActConv(skip_maps_in, num_maps_in, k=1)
        self.blend_conv = PreActConv(num_maps_in, num_maps_out, k=k)

    def forward(self, x, skip):
        with timings.env("module/skip", 3):
            skip = self.bottleneck(skip)
        with timings.env("module/upsample", 3):
            x = F.interpolate(x, (x.shape[2] * 2, x.shape[3] * 2), mode="bilinear")
        with timings.env("module/blend", 3):
            x += skip
            return self.blend_conv(x)


class SpatialPyramidPooling(nn.Module):
    """Pyramid pooling over the dense coarsest map (needs the whole image => runs un-packed)."""

    def __init__(self, num_maps_in, num_levels, bt_size=512, level_size=128, out_size=128, grids=(6, 3, 2, 1), bn_momentum=0.1):
        super().__init__()
        self.grids = grids
        self.spp = nn.Sequential()
        self.spp.add_module("spp_bn", PreActConv(num_maps_in, bt_size, k=1, bn_momentum=bn_momentum))
        for i in range(num_levels):
            self.spp.add_module(f"spp{i}", PreActConv(bt_size, level_size, k=1, bn_momentum=bn_momentum))
        self.spp.add_module("spp_fuse", PreActConv(bt_size + num_levels * level_size, out_size, k=1, bn_momentum=bn_momentum))

    @blockcopy_noblocks
    def forward(self, x):
        with timings.env("module/spp_center", 3):
            size = x.size()[2:4]
            ar = size[1] / size[0]
            x = self.spp[0](x)
            levels = [x]
            # the stock adaptive pooling of a channels-last map is ~5x slower than of an NCHW one on ROCm; the map is tiny
            # (stride 32), so plain tensors pool from one NCHW copy (values are identical).  Inside the block engine the map
            # is a TensorWrapper and the library pools channels-last maps itself (bc_adaptive_avg_pool_nhwc): no copy.
            engine_pools = isinstance(x, TensorWrapper) and x.is_cuda
            x_pool = x if (engine_pools or x.is_contiguous()) else x.contiguous()
            for i in range(1, len(self.spp) - 1):
                g = self.grids[i - 1]
                pooled = F.adaptive_avg_pool2d(x_pool, (g, max(1, round(ar * g))))
                levels.append(F.interpolate(self.spp[i](pooled), size, mode="bilinear"))
            return self.spp[-1](torch.cat(levels, 1))


# ----------------------------------------------------------------------------------------------- the network
class SwiftNet(nn.Module):
    def __init__(self, backbone, num_classes, num_features=128, k_up=3, spp_grids=(8, 4, 2, 1)):
        super().__init__()
        assert num_classes > 0
        self.backbone = backbone
        self.num_classes = num_classes
        self.num_features = num_features
        feats = backbone.block_features
        num_levels = 3
        self.spp = SpatialPyramidPooling(feats[3], num_levels, bt_size=num_features, level_size=num_features // num_levels,
                                         out_size=num_features, grids=spp_grids, bn_momentum=0.01 / 2)
        # coarse-to-fine: upsample[0] takes the stride-16 skip, upsample[2] the stride-4 skip
        self.upsample = nn.ModuleList([LadderUp(num_features, feats[i], num_features, k=k_up) for i in (2, 1, 0)])
        self.logits = PreActConv(num_features, num_classes, k=1, bias=True)

    def forward_down(self, image):
        with timings.env("model/forward_down", 2):
            return self.backbone.forward_down(image)

    def forward_up(self, features):
        with timings.env("model/forward_up", 2):
            feats = features[::-1]
            with timings.env("model/spp", 3):
                x = self.spp(feats[0])
            for skip, up in zip(feats[1:], self.upsample):
                x = up(x, skip)
            return self.logits(x)

    def forward(self, image, additional=None):
        return self.forward_up(self.forward_down(image))


def build_swiftnet(backbone: str = "resnet18", num_classes: int = 19, num_features: int = 128) -> SwiftNet:
    return SwiftNet(BACKBONES[backbone](), num_classes=num_classes, num_features=num_features)
