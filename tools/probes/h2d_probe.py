#!/usr/bin/env python3
"""How fast does one 25 MB pinned frame reach the device?  One copy on one stream against the same bytes split over 2 / 4 streams
(each stream's copies are served by its own DMA engine), fp32 -> fp32 and fp32 -> fp16 (conversion on the device side).
usage: python tools/probes/h2d_probe.py   (HSA_ENABLE_SDMA=0 selects shader copies)"""
import os
import time
import torch

dev = torch.device("cuda")
shape = (1, 3, 1024, 2048)
hosts = [torch.randn(shape).pin_memory() for _ in range(4)]
dst = [torch.empty(shape, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(4)]
nbytes = hosts[0].numel() * 4


def run(parts, iters=60):
    flat_d = [d.view(-1) for d in dst]
    flat_h = [h.view(-1) for h in hosts]
    n = flat_h[0].numel()
    step = n // parts
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        h, d = flat_h[i % 4], flat_d[i % 2]
        for p in range(parts):
            with torch.cuda.stream(streams[p]):
                d[p * step:(p + 1) * step].copy_(h[p * step:(p + 1) * step], non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return dt


print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
for parts in (1, 2, 3, 4):
    run(parts, 5)
    dt = run(parts)
    print(f"{parts} stream(s): {dt * 1e3:.3f} ms per frame = {nbytes / dt / 1e9:.1f} GB/s -> upload-bound ceiling {1 / dt:.0f} fps")
