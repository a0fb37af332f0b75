#!/usr/bin/env python3
"""List the stock PyTorch elementwise / copy / pooling ops one eager C2 frame still launches (name, input shapes, device
time) -- the tool that found the BN + layout-copy + ReLU chains on the dense pyramid-pooling maps and the stem pool.
usage: python tools/list_dense_ops.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from bc_workloads import harness, seeded
torch.backends.cudnn.benchmark = True
model = harness.build_model("resnet18", block_policy="fixed", block_size=128, block_target=0.5, device="cuda", channels_last=True, block_graph=0)
frames = harness.synthetic_clip(4, (1, 3, 1024, 2048), seed=0)
with torch.no_grad():
    harness.run_clip(model, frames)
    harness.run_clip(model, frames)
    torch.cuda.synchronize()
    model.reset_temporal(); model(frames[0]); model(frames[1])
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        model(frames[2])
        torch.cuda.synchronize()
rows = []
for e in prof.events():
    if e.name in ("aten::copy_", "aten::contiguous", "aten::clone", "aten::_to_copy", "aten::cat", "aten::add", "aten::add_", "aten::clamp_min_", "aten::relu_", "aten::relu", "aten::batch_norm", "aten::native_batch_norm", "aten::adaptive_avg_pool2d", "aten::upsample_bilinear2d", "aten::max_pool2d_with_indices", "aten::fill_", "aten::zero_"):
        rows.append((e.name, str(e.input_shapes)[:100], round(e.device_time_total, 1) if hasattr(e, "device_time_total") else None))
for r in rows:
    print(r)
