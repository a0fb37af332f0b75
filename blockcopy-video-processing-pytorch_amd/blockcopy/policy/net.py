"""Policy network of the online-RL policies: one execution logit per tile.

What it sees (reference recipe, policy/net.py:17-125, kept so that trained policies transfer): the frame and everything
the engine knows about the previous step, all brought to ONE low resolution by nearest-neighbour resampling and stacked
along channels --

    frame (3) | frame state = last executed pixels (3) | previous task output as class scores (num_classes) | previous grid (1)

with the resolution chosen so that one tile covers 32 x 32 policy pixels (scale = 32 / block_size).  A resnet8 trunk
(width x2, stride 4) and three stride-2 3x3 convs (128, 128, 1 channels) then reduce every tile to one logit.
Parameter names (``backbone.*``, ``layers.<i>.<j>.*``) match the reference's, so its checkpoints load.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from blockcopy.policy.fused_bn import PolicyBatchNorm2d
from blockcopy.policy.resnet import resnet8
from blockcopy.utils.profiler import timings

import os

FUSED_FEATURES = os.environ.get("BLOCKCOPY_FUSED_FEATURES", "1") != "0"
POLICY_PIXELS_PER_TILE = 32
HEAD_WIDTH = 128


def _head_stage(cin: int, cout: int, last: bool) -> nn.Sequential:
    """3x3 stride-2 conv; every stage but the last is followed by BN (slow running stats) and ReLU and has no bias."""
    mods: List[nn.Module] = [nn.Conv2d(cin, cout, kernel_size=3, stride=2, padding=1, bias=last)]
    if not last:
        mods += [PolicyBatchNorm2d(cout, momentum=0.02), nn.ReLU(inplace=False)]
    return nn.Sequential(*mods)


class PolicyNet(nn.Module):
    def __init__(self, block_size: int, task_num_classes: int) -> None:
        super().__init__()
        self.block_size = block_size
        self.task_num_classes = task_num_classes
        self.scale_factor = POLICY_PIXELS_PER_TILE / block_size
        self.backbone = resnet8(pretrained=False, in_channels=3 + 3 + task_num_classes + 1, width_factor=2)
        self.layers = nn.Sequential(_head_stage(self.backbone.OUT_CHANNELS, HEAD_WIDTH, last=False),
                                    _head_stage(HEAD_WIDTH, HEAD_WIDTH, last=False),
                                    _head_stage(HEAD_WIDTH, 1, last=True))

    def build_features(self, policy_meta: Dict) -> torch.Tensor:
        """(N, 7 + num_classes, h, w) policy input; nothing here carries gradient."""
        frame = policy_meta["inputs"]
        if frame.dim() != 4 or frame.size(1) != 3:
            raise ValueError(f"policy expects (N,3,H,W) frames, got {tuple(frame.shape)}")
        if frame.is_cuda and FUSED_FEATURES:
            fused = self._build_features_fused(policy_meta)
            if fused is not None:
                return fused
        low = F.interpolate(frame, scale_factor=self.scale_factor, mode="nearest").float()
        hw = low.shape[2:]

        def at_policy_resolution(t: torch.Tensor, centre: bool) -> torch.Tensor:
            assert t is not None and t.dim() == 4
            t = F.interpolate(t.to(low.dtype) if t.dtype == torch.bool else t, size=hw, mode="nearest").to(low.dtype)
            return t - 0.5 if centre else t    # scores and masks live in [0, 1]: centre them

        parts = [low,
                 at_policy_resolution(policy_meta["frame_state"], centre=False),
                 at_policy_resolution(policy_meta.get("output_repr"), centre=True),
                 at_policy_resolution(policy_meta.get("grid"), centre=True)]
        return torch.cat(parts, dim=1).detach()

    def feature_sources(self, policy_meta: Dict):
        """([(tensor, scale_h, scale_w, offset)] * 4, h, w): the four sources of the policy input with the float32 source-index scales
        of F.interpolate(mode='nearest') (1/scale_factor where a factor is given, in/out where a size is given); None when a source
        is missing or not a GPU tensor the gather kernels read."""
        import numpy as np

        frame, state, rep, grid = (policy_meta["inputs"], policy_meta["frame_state"], policy_meta.get("output_repr"), policy_meta.get("grid"))
        ok = lambda t: t is not None and t.is_cuda and t.dim() == 4 and t.dtype in (torch.float32, torch.float16, torch.bfloat16, torch.bool, torch.uint8)
        if not all(ok(t) for t in (frame, state, rep, grid)):
            return None
        H, W = frame.shape[2:]
        h, w = int(np.floor(H * self.scale_factor)), int(np.floor(W * self.scale_factor))
        by_factor = float(np.float32(1.0 / self.scale_factor))
        by_size = lambda t: (float(np.float32(t.shape[2]) / np.float32(h)), float(np.float32(t.shape[3]) / np.float32(w)))
        srcs = [(frame, by_factor, by_factor, 0.0), (state,) + by_size(state) + (0.0,), (rep,) + by_size(rep) + (-0.5,), (grid,) + by_size(grid) + (-0.5,)]
        return srcs, h, w

    def _build_features_fused(self, policy_meta: Dict):
        """The same tensor from ONE gather kernel (bc_policy_features) instead of 4 resamplings + casts + 2 subtractions +
        concat; index arithmetic identical to F.interpolate(mode='nearest'), so the result is bit-identical."""
        from blockcopy.backend import get_backend

        be = get_backend()
        if not hasattr(be, "policy_features"):
            return None
        srcs = self.feature_sources(policy_meta)
        if srcs is None:
            return None
        srcs, h, w = srcs
        with torch.no_grad():
            return be.policy_features(srcs, h, w)

    def forward(self, policy_meta: Dict) -> torch.Tensor:
        n, _, height, width = policy_meta["inputs"].shape
        with timings.env("policy/net/build_features", 5):
            x = self.build_features(policy_meta)
        with timings.env("policy/net/layers", 5):
            logits = self.layers(self.backbone(x))
        want = (n, 1, height // self.block_size, width // self.block_size)
        if tuple(logits.shape) != want:
            raise AssertionError(f"policy logits {tuple(logits.shape)} do not tile a {height}x{width} frame with block {self.block_size}")
        return logits


def build_policy_net_from_settings(settings: dict) -> PolicyNet:
    return PolicyNet(block_size=settings["block_size"], task_num_classes=settings["block_num_classes"])
