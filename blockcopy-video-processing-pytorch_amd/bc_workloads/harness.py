"""Synthetic-clip harness: builds the workload the benchmark configs name and measures frames/s the way the
reference's driver does (semantic_segmentation/test_swiftnet.py:134-231): ``reset_temporal()`` per clip, one
``model(frame)`` per frame under ``no_grad``, device sync before the first and after the last frame,
fps = frames / wall."""
from __future__ import annotations

import time
from typing import List, Sequence

import torch

import blockcopy
from blockcopy.core.argparser import default_settings

from . import seeded
from .bn_fold import fold_batchnorm
from .swiftnet import build_swiftnet


def build_model(backbone="resnet18", block_policy="fixed", block_size=128, block_target=0.5, device="cuda",
                dtype=torch.float32, fold_bn=True, seed=0, channels_last=False, **settings_overrides):
    """SwiftNet with name-seeded weights, optionally wrapped in BlockCopyModel (``block_policy='static'`` = dense)."""
    net = build_swiftnet(backbone)
    net.load_state_dict(seeded.name_seeded_state_dict(net.state_dict()), strict=True)
    net.eval()
    model = net
    if block_policy != "static":
        settings = default_settings(block_policy=block_policy, block_size=block_size, block_target=block_target,
                                    block_seed=seed, **settings_overrides)
        model = blockcopy.BlockCopyModel(net, settings)
    model = model.to(device)
    if fold_bn:
        model = fold_batchnorm(model)
    if channels_last:
        # weights in channels-last: MIOpen then produces channels-last activations and the engine's packed tiles,
        # ring caches and dense maps follow (every halo access becomes an aligned vector)
        model = model.to(memory_format=torch.channels_last)
    if dtype != torch.float32:
        model = model.to(dtype)
        if block_policy != "static" and model.policy.net is not None:
            model.policy.net = model.policy.net.float()   # the policy trains in fp32 (reference test_swiftnet.py:118-123)
    return model


def synthetic_clip(n_frames: int, shape, seed: int = 0, device="cuda", dtype=torch.float32, static: bool = False) -> List[torch.Tensor]:
    """``n_frames`` seeded randn frames of ``shape`` already resident on ``device`` (static=True repeats frame 0)."""
    frames = []
    for t in range(n_frames):
        f = seeded.synthetic_frame(seed if static else seed + t, shape, dtype)
        frames.append(f.to(device))
    return frames


@torch.no_grad()
def run_clip(model, frames: Sequence[torch.Tensor]):
    """One clip through the model; returns the last frame's output."""
    if hasattr(model, "reset_temporal"):
        model.reset_temporal()
    out = None
    for f in frames:
        out = model(f)
    return out


def sync(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()


def measure_fps(model, clips: Sequence[Sequence[torch.Tensor]], n_clips: int, warmup_clips: int = 1, device="cuda"):
    """frames/s over ``n_clips`` clips (cycling through ``clips``), after ``warmup_clips`` untimed clips."""
    for i in range(warmup_clips):
        run_clip(model, clips[i % len(clips)])
    sync(device)
    t0 = time.perf_counter()
    n_frames = 0
    for i in range(n_clips):
        clip = clips[i % len(clips)]
        run_clip(model, clip)
        n_frames += len(clip) * clip[0].shape[0]
    sync(device)
    dt = time.perf_counter() - t0
    return n_frames / dt, dt, n_frames
