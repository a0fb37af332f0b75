#!/usr/bin/env python3
"""layer3 / layer4 / up 1/16 shapes of C2 through the shared-transform F(2x2) forms (codes 0x20b..0x20d) and the best plain one: a quick
A/B of kernel variants (BLOCKCOPY_HIP_LIB selects the library).  usage: python tools/kbench_l34.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import blockcopy.backend as bk
from kbench import grid_tables, timeit

be = bk.get_backend()
for name, n, Cin, Cout, bs, cfgs in [("layer3", 64, 256, 256, 8, (0x20c, 0x207)), ("layer4", 64, 512, 512, 4, (0x20d, 0x208)), ("layer2", 64, 128, 128, 16, (0x20b,)),
                                     ("layer4 n128", 128, 512, 512, 4, (0x20c, 0x20d, 0x207))]:
    gi, m = grid_tables(1, 8, 16, n)
    feats = torch.relu(torch.randn((n, Cin, bs, bs), device="cuda")).contiguous(memory_format=torch.channels_last)
    ring = torch.randn((128, Cin, 4 * bs), device="cuda")
    w = (torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    wpk = be.pack_conv3x3_weights(w)
    sc = torch.rand(Cin, device="cuda") + 0.5
    out = []
    for cfg in cfgs:
        f = lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, (sc, sc, True), None, cfg=cfg)
        ts = sorted(timeit(f, 20) for _ in range(5))
        out.append(f"{cfg:#x}={ts[2]:.1f} (min {ts[0]:.1f})")
    print(f"{name:12s} " + ", ".join(out), flush=True)
