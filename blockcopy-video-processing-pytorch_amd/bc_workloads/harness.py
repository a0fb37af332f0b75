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


@torch.no_grad()
def measure_fps_with_upload(model, host_clips, n_clips: int, warmup_clips: int = 1, device="cuda", dtype=torch.float32, prefetch: bool = True, cross_clip: bool = True,
                            fused_tail: bool = None):
    """frames/s of the reference driver's FULL per-clip loop (semantic_segmentation/test_swiftnet.py:181-197): every frame
    starts in (pinned) host memory and is uploaded inside the timed region, and the last frame of each clip is upsampled to
    the input resolution, arg-maxed and copied back to the host (``preds = out.max(dim=1)[1].cpu()``).

    ``prefetch=False`` is the reference's own form: ``inputs.to(device, non_blocking=True)`` on the compute stream, i.e. the
    25 MB upload of frame t sits between the models of frames t-1 and t, and ``.cpu()`` of the predictions blocks the host at the
    end of every clip.  ``prefetch=True`` is the MI355X-first form of the same loop: frame t+1 travels on a copy stream into the
    next of three device buffers while frame t is computed (PCIe and compute overlap), and the predictions of a clip travel back
    into pinned host memory on a second copy stream while the next clip is already being enqueued -- every clip's predictions still end
    up on the host (checked one clip later), the host just does not stall for them.  ``fused_tail`` (default: with ``prefetch``): the
    prediction map of the last frame comes from ``blockcopy.utils.postprocess.upsample_argmax`` (one pass over the logits) instead of
    the two stock ops with their 160 MB intermediate; same predictions (tests/test_gpu_ops.py)."""
    if fused_tail is None:
        fused_tail = prefetch
    if fused_tail:
        from blockcopy.utils.postprocess import upsample_argmax
    dev = torch.device(device)
    host_clips = [[f.pin_memory() if not f.is_pinned() else f for f in clip] for clip in host_clips]
    compute = torch.cuda.current_stream(dev)
    copy = torch.cuda.Stream(dev) if prefetch else None
    shape = host_clips[0][0].shape
    # THREE device buffers: the upload of frame t + 1 reuses the buffer of frame t - 2, and the HOST makes sure that frame is through
    # (an event query that has long succeeded in the steady state) -- with two buffers the copy stream itself had to wait for frame t - 1
    # (a device-side cross-stream dependency per frame; measured, profiles/r04/41: the four event operations per frame cost 3.4 %)
    NB = 3 if prefetch else 2
    bufs = [torch.empty(shape, dtype=dtype, device=dev) for _ in range(NB)]
    staged = [torch.cuda.Event() for _ in range(NB)]      # upload into bufs[i] finished
    consumed = [torch.cuda.Event() for _ in range(NB)]    # model finished reading bufs[i]

    # double-buffered download of the per-clip predictions, on a stream of its own (behind the uploads it would delay the next clip's first frame)
    down = {"n": 0, "host": [None, None], "ev": [None, None], "stream": torch.cuda.Stream(dev) if prefetch else None}

    def upload(frame, i):
        consumed[i].synchronize()                          # (the frame that read bufs[i] three frames ago)
        with torch.cuda.stream(copy):
            bufs[i].copy_(frame, non_blocking=True)       # fp32 -> dtype conversion happens on the device side of the copy
            staged[i].record(copy)

    state = {"slot": 0, "primed": False}      # device buffer of the next frame; whether that frame is already on its way

    def run(clip, next_clip=None):
        if hasattr(model, "reset_temporal"):
            model.reset_temporal()
        preds = None
        if prefetch and not state["primed"]:
            upload(clip[0], state["slot"])
        for t, frame in enumerate(clip):
            if prefetch:
                k = state["slot"]
                # frame t + 1 -- or the first frame of the NEXT clip -- travels while frame t is computed
                nxt = clip[t + 1] if t + 1 < len(clip) else (next_clip[0] if next_clip is not None else None)
                if nxt is not None:
                    upload(nxt, (k + 1) % NB)
                state["primed"] = nxt is not None
                # the HOST waits for the upload (requested one frame ago: long finished), not the compute stream: the frame's graph is
                # then enqueued without any cross-stream dependency in front of it
                staged[k].synchronize()
                inputs = bufs[k]
                state["slot"] = (k + 1) % NB
            else:
                inputs = frame.to(dev, non_blocking=True, dtype=dtype)
            out = model(inputs)
            if prefetch:
                consumed[k].record(compute)
            if t == len(clip) - 1:
                if fused_tail:
                    preds = upsample_argmax(out.detach(), tuple(inputs.shape[2:]))
                else:
                    out = torch.nn.functional.interpolate(out, size=inputs.shape[2:], mode="bilinear")
                    preds = out.detach().max(dim=1)[1]
                if prefetch:
                    k = down["n"] % 2
                    down["n"] += 1
                    if down["ev"][k] is not None:
                        down["ev"][k].synchronize()              # the predictions of two clips ago have long arrived
                    if down["host"][k] is None:
                        down["host"][k] = torch.empty(preds.shape, dtype=preds.dtype, pin_memory=True)
                    ready = torch.cuda.Event()
                    ready.record(compute)
                    with torch.cuda.stream(down["stream"]):
                        down["stream"].wait_event(ready)
                        down["host"][k].copy_(preds, non_blocking=True)
                        preds.record_stream(down["stream"])
                        down["ev"][k] = torch.cuda.Event()
                        down["ev"][k].record(down["stream"])
                    preds = down["host"][k]
                else:
                    preds = preds.cpu()
        return preds

    for i in range(NB):
        consumed[i].record(compute)
    for i in range(warmup_clips):
        run(host_clips[i % len(host_clips)])
    sync(dev)
    t0 = time.perf_counter()
    n_frames = 0
    for i in range(n_clips):
        clip = host_clips[i % len(host_clips)]
        nxt = host_clips[(i + 1) % len(host_clips)] if (cross_clip and i + 1 < n_clips) else None
        preds = run(clip, nxt)
        n_frames += len(clip) * clip[0].shape[0]
    sync(dev)
    dt = time.perf_counter() - t0
    assert preds is not None and tuple(preds.shape) == (shape[0], shape[2], shape[3])
    measure_fps_with_upload.last_predictions = preds.clone()      # (host tensor: the last clip's prediction map, for the tests)
    return n_frames / dt, dt, n_frames
