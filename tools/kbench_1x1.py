#!/usr/bin/env python3
"""Pointwise (1x1) convs of the ResNet-50 bottlenecks at the C4 / C5 packed shapes: every decomposition of the one-tap form of the fused
conv kernel (bc_conv1x1_nhwc) against the library conv (+ the affine pass the library route needs for the folded BN / ReLU), hipGraph
replay timing.  usage: python tools/kbench_1x1.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import blockcopy.backend as bk
from kbench import timeit

be = bk.get_backend()
torch.backends.cudnn.benchmark = True
# (tiles, bs, cin, cout): C4 = 512 executed tiles of a 2048x4096 frame at block 64; C5 = 38 tiles at block 128
SHAPES = [(512, 16, 64, 256), (512, 16, 256, 64), (512, 16, 256, 128), (512, 8, 128, 512), (512, 8, 512, 128), (512, 8, 512, 256), (512, 4, 256, 1024),
          (512, 4, 1024, 256), (512, 4, 1024, 512), (512, 2, 512, 2048), (512, 2, 2048, 512), (38, 32, 64, 256), (38, 32, 256, 64), (38, 16, 128, 512),
          (38, 16, 512, 128), (38, 8, 256, 1024), (38, 8, 1024, 256), (38, 8, 512, 2048), (38, 8, 2048, 512)]
for dtype in (torch.float32, torch.float16):
    for n, bs, cin, cout in SHAPES:
        x = torch.randn((n, cin, bs, bs), device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((cout, cin, 1, 1), device="cuda") / cin ** 0.5).to(dtype)
        sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
        gf = 2.0 * n * bs * bs * cin * cout / 1e9
        lib = timeit(lambda: torch.nn.functional.conv2d(x, w), 10)
        y = torch.nn.functional.conv2d(x, w)
        lib_aff = lib + timeit(lambda: be.affine_act(y, sc, sh, None, True), 10)
        line = f"{str(dtype)[6:]:8s} {n:4d} tiles {bs:2d}x{bs:<2d} {cin:4d}->{cout:<4d} {gf:5.1f} GF | library {lib:6.1f} us ({gf / lib * 1e3:5.1f} TF), + affine pass {lib_aff:6.1f} |"
        if be.conv1x1_supported(x, w):
            wpk = be.pack_conv3x3_weights(w) if not hasattr(be, "pack_conv1x1_weights") else be.pack_conv1x1_weights(w)
            res = {}
            for c in be.conv1x1_candidates(x, cout):
                res[c] = timeit((lambda c_: lambda: be.conv1x1(x, wpk, cout, None, (sc, sh, None, True), cfg=c_))(c), 10)
            best = sorted((t, c) for c, t in res.items() if not c & 0x800)[:2]
            gemm = sorted((t, c) for c, t in res.items() if c & 0x800)[:2]
            line += " one-tap form (epilogue fused): " + ", ".join(f"{c}={t:.1f} ({gf / t * 1e3:.0f} TF)" for t, c in best)
            line += " | GEMM form: " + ", ".join(f"{c}={t:.1f} ({gf / t * 1e3:.0f} TF)" for t, c in gemm)
        else:
            line += " own: not covered"
        print(line, flush=True)
