#!/usr/bin/env python3
"""Which op of one eager frame costs what: every call into the HIP backend (conv3x3_ring, conv1x1, head1x1, pad_ring, ...) and every
library conv / transposed conv / norm that PyTorch still runs is bracketed with events and listed with its shapes, sorted by device
time (the per-layer view the kernel trace cannot give: the trace has kernel names, not layer shapes).
usage: python tools/frame_ops.py [--config C2|C4|C5] [--half] [--frames 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
import torch

import bench
import blockcopy.backend as bk

RECORDS = []
ACTIVE = [False]
DEPTH = [0]


def shapes(args, kwargs):
    out = []
    for a in list(args) + list(kwargs.values()):
        if isinstance(a, torch.Tensor) and a.dim() >= 3:
            out.append("x".join(map(str, a.shape)))
    return " ".join(out[:3])


def wrap(owner, name, label):
    fn = getattr(owner, name)

    def inner(*args, **kwargs):
        if not ACTIVE[0] or DEPTH[0]:
            return fn(*args, **kwargs)
        DEPTH[0] += 1
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        try:
            r = fn(*args, **kwargs)
        finally:
            DEPTH[0] -= 1
        b.record()
        extra = " ".join(f"{k}={v}" for k, v in kwargs.items() if k in ("stride", "dilation", "padding", "cfg", "groups") and not isinstance(v, torch.Tensor))
        RECORDS.append((label, shapes(args, kwargs), extra, a, b))
        return r

    setattr(owner, name, inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C5")
    ap.add_argument("--half", action="store_true")
    ap.add_argument("--frames", type=int, default=3)
    a = ap.parse_args()
    args = bench.parse_args(["--config", a.config] + (["--half"] if a.half else []))
    dtype = torch.float16 if a.half else torch.float32
    be = bk.get_backend()
    for name in dir(type(be)):
        if name.startswith("_") or name in ("tune", "tune_get", "time_routes", "prof_read", "prof_enable", "conv3x3_supported", "conv3x3_candidates",
                                            "pack_conv3x3_weights", "pack_head1x1_weights", "head1x1_supported", "conv1x1_supported", "grid_tables"):
            continue
        if callable(getattr(be, name, None)) and not name.endswith("_supported"):
            wrap(be, name, "hip:" + name)      # (on the instance: bound methods and static methods alike)
    for name in ("conv2d", "conv_transpose2d", "group_norm", "batch_norm", "interpolate", "max_pool2d"):
        wrap(torch.nn.functional, name, "torch:" + name)
    from bc_workloads import harness

    model = bench.build_workload(args, args.policy, dtype, "cuda", 0, graph=0)
    frames = harness.synthetic_clip(a.frames + 3, (1, 3, args.height, args.width), seed=0, device="cuda", dtype=dtype)
    with torch.no_grad():
        model.reset_temporal()
        for f in frames[:3]:
            model(f)
        torch.cuda.synchronize()
        ACTIVE[0] = True
        for f in frames[3:]:
            model(f)
        torch.cuda.synchronize()
        ACTIVE[0] = False
    agg = {}
    for label, shp, extra, e0, e1 in RECORDS:
        k = (label, shp, extra)
        t = e0.elapsed_time(e1) * 1e3
        n, s = agg.get(k, (0, 0.0))
        agg[k] = (n + 1, s + t)
    total = sum(s for _, s in agg.values()) / a.frames
    print(f"{a.config}{' fp16' if a.half else ''}: {a.frames} eager frames, bracketed ops {total:.0f} us/frame (event-to-event: includes launch gaps of multi-kernel ops)")
    by_label = {}
    for (label, shp, extra), (n, s) in agg.items():
        by_label[label] = by_label.get(label, 0.0) + s / a.frames
    for label, s in sorted(by_label.items(), key=lambda kv: -kv[1]):
        print(f"  {label:28s} {s:9.1f} us/frame")
    print()
    for (label, shp, extra), (n, s) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{s / a.frames:9.1f} us/frame  {n / a.frames:5.1f} calls  {s / n:8.1f} us  {label:24s} {shp}  {extra}")


if __name__ == "__main__":
    main()
