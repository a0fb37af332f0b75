#!/usr/bin/env python3
"""bc_maxpool3x3s2_ring_nhwc (fused halo + 3x3 / stride 2 max-pool of the ResNet stem) at the C2 / C4 shapes, hipGraph replay timing.
usage: python tools/kbench_pool.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import blockcopy.backend as bk
from kbench import timeit, grid_tables

be = bk.get_backend()
for name, (N, GH, GW, n, bs, C) in {"C2 (64 of 128 tiles, 64x64x64)": (1, 8, 16, 64, 64, 64), "C2 first frame": (1, 8, 16, 128, 64, 64),
                                     "C4 (512 of 2048 tiles, 32x32x64)": (1, 32, 64, 512, 32, 64)}.items():
    gi, m = grid_tables(N, GH, GW, n)
    for dt in (torch.float32, torch.float16):
        x = torch.randn((n, C, bs, bs), device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        ring = torch.randn((N * GH * GW, C, 4 * bs), device="cuda").to(dt)
        sh = torch.randn(C, device="cuda")
        for pro in (None, (None, sh, True)):
            us = timeit(lambda: be.maxpool3x3s2_ring(x, ring, gi, m, pro), 10)
            by = n * C * x.element_size() * ((bs + 1) ** 2 + (bs // 2) ** 2 + 4 * bs)
            print(f"{name} {str(dt)[6:]:8s} prologue {pro is not None!s:5s}: {us:6.1f} us, {by / us / 1e3:6.0f} GB/s", flush=True)
