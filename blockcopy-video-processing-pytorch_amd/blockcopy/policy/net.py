"""Policy network: one execution logit per tile from (down-scaled frame, frame state, previous output, previous grid).

Architecture and input recipe follow the reference (policy/net.py:17-125): nearest-downscale by 0.25*128/block_size,
channel concat, resnet8(width x2), then three stride-2 3x3 convs 128 -> 128 -> 1."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from blockcopy.policy.resnet import resnet8
from blockcopy.utils.profiler import timings


def build_policy_net_from_settings(settings: dict):
    return PolicyNet(block_size=settings["block_size"], task_num_classes=settings["block_num_classes"])


class PolicyNet(nn.Module):
    def __init__(self, block_size, task_num_classes) -> None:
        super().__init__()
        self.block_size = block_size
        self.scale_factor = 0.25 * 128 / self.block_size
        self.use_frame_state = True
        self.use_prev_output = True
        self.use_prev_grid = True
        self.task_num_classes = task_num_classes
        in_channels = 3 + (3 if self.use_frame_state else 0) + (task_num_classes if self.use_prev_output else 0) \
            + (1 if self.use_prev_grid else 0)
        self.backbone = resnet8(pretrained=False, in_channels=in_channels, width_factor=2)
        planes = 128
        self.layers = nn.Sequential(
            self._make_layer(self.backbone.OUT_CHANNELS, planes, stride=2, relu=True),
            self._make_layer(planes, planes, stride=2, relu=True),
            self._make_layer(planes, 1, stride=2, relu=False))

    @staticmethod
    def _make_layer(cin, cout, kernel_size=3, stride=1, relu=True):
        mods = [nn.Conv2d(cin, cout, kernel_size=kernel_size, padding=(kernel_size - 1) // 2, stride=stride, bias=not relu)]
        if relu:
            mods += [nn.BatchNorm2d(cout, momentum=0.02), nn.ReLU(inplace=False)]
        return nn.Sequential(*mods)

    def build_features(self, policy_meta: dict) -> torch.Tensor:
        frame = policy_meta["inputs"]
        assert frame.dim() == 4 and frame.size(1) == 3
        feats = [F.interpolate(frame, scale_factor=self.scale_factor, mode="nearest").float()]
        size = feats[0].shape[2:]
        if self.use_frame_state:
            feats.append(F.interpolate(policy_meta["frame_state"], size=size, mode="nearest").float())
        if self.use_prev_output:
            assert policy_meta.get("output_repr", None) is not None
            rep = policy_meta["output_repr"]
            assert rep.dim() == 4
            feats.append(F.interpolate(rep, size=size, mode="nearest").type(feats[0].dtype) - 0.5)
        if self.use_prev_grid:
            assert policy_meta.get("grid", None) is not None
            g = policy_meta["grid"].type(feats[0].dtype)
            assert g.dim() == 4
            feats.append(F.interpolate(g, size=size, mode="nearest") - 0.5)
        return torch.cat(feats, dim=1).detach()

    def forward(self, policy_meta: dict):
        N, C, H, W = policy_meta["inputs"].shape
        with timings.env("policy/net/build_features", 5):
            x = self.build_features(policy_meta)
        with timings.env("policy/net/layers", 5):
            logits = self.layers(self.backbone(x))
        expect = (N, 1, H // self.block_size, W // self.block_size)
        assert logits.shape == expect, f"logits shape: {logits.shape}, frame shape: {(N, C, H, W)}, block size: {self.block_size}"
        return logits
