#!/usr/bin/env python3
"""Direct (conv3x3_v2.inc) vs Winograd F(2x2,3x3) (conv3x3_wino.inc) form of the fused halo + 3x3 conv at the layer shapes of the
benchmark configs: every candidate decomposition of both forms is timed (hipGraph replay, prologue on, post-ReLU-like inputs) and
the best three / four of each are printed (Winograd codes 0x200 | 11..13: the shared-transform variant).  usage: python tools/kbench_wino.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import blockcopy.backend as bk
from kbench import grid_tables, timeit
be = bk.get_backend()
for name, n, Cin, Cout, bs in [("layer1", 64, 64, 64, 32), ("layer2", 64, 128, 128, 16), ("layer3", 64, 256, 256, 8), ("layer4", 64, 512, 512, 4), ("up1/8", 64, 128, 128, 16), ("up1/4", 64, 128, 128, 32), ("up1/16", 64, 128, 128, 8),
                               ("layer1 n128", 128, 64, 64, 32), ("layer2 n128", 128, 128, 128, 16), ("csp head n38", 38, 768, 256, 32)]:
    gi, m = grid_tables(1, 8, 16, n)
    feats = torch.relu(torch.randn((n, Cin, bs, bs), device="cuda")).contiguous(memory_format=torch.channels_last)
    ring = torch.randn((128, Cin, 4 * bs), device="cuda")
    w = (torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    wpk = be.pack_conv3x3_weights(w)
    sc = torch.rand(Cin, device="cuda") + 0.5
    res = {}
    for cfg in be.conv3x3_candidates(n, Cin, Cout, bs, 4, 1):
        f = lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, (sc, sc, True), None, cfg=cfg)
        res[cfg] = timeit(f, 10)
    v2 = sorted((t, c) for c, t in res.items() if not c & 0x600)[:3]
    wn = sorted((t, c) for c, t in res.items() if c & 0x200 and (c & 0xff) < 11)[:3]
    sh = sorted((t, c) for c, t in res.items() if c & 0x200 and (c & 0xff) >= 11)[:3]
    ww = sorted((t, c) for c, t in res.items() if c & 0x400)[:5]
    print(f"{name:14s} direct best: " + ", ".join(f"{c}={t:.1f}" for t, c in v2) + " | winograd best: " + ", ".join(f"{c}={t:.1f}" for t, c in wn)
          + " | shared transform: " + ", ".join(f"{c}={t:.1f}" for t, c in sh) + " | wide winograd: " + ", ".join(f"{c & 0xff}={t:.1f}" for t, c in ww), flush=True)
