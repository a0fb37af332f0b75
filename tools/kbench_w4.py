#!/usr/bin/env python3
"""The four-wave decompositions of the split conv form (codes 0x2000 | 20..24: 4 x 2-block wave tiles, one wave per SIMD; csrc/conv3x3_v2.inc)
against the best eight-wave decomposition of the same form, on the large layers of the benchmark configs: 3x3 layers on packed tiles and
pointwise layers (incl. the detector neck's tap GEMMs).  usage: python tools/kbench_w4.py [--filter substr]"""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import blockcopy.backend as bk  # noqa: E402
from kbench import grid_tables, timeit  # noqa: E402

CONV3 = [("C5 head 768->256", 8, 16, 38, 768, 256, 32), ("C2 up 1/4 128->128", 8, 16, 64, 128, 128, 32), ("C2 layer1 64->64", 8, 16, 64, 64, 64, 32),
         ("C2 layer2 128->128", 8, 16, 64, 128, 128, 16), ("C4 layer1 rn50 64->64", 32, 64, 512, 64, 64, 16), ("C5 layer2 128->128", 8, 16, 38, 128, 128, 16)]
CONV1 = [("C5 neck p3 512->4096", 38, 512, 4096, 16), ("C5 neck p4 1024->4096", 38, 1024, 4096, 8), ("C5 neck p5 2048->4096", 38, 2048, 4096, 8),
         ("C5 layer1 64->256", 38, 64, 256, 32), ("C5 layer1 256->64", 38, 256, 64, 32), ("C5 layer2 512->128", 38, 512, 128, 16), ("C5 layer3 1024->256", 38, 1024, 256, 8),
         ("C4 layer1 64->256", 512, 64, 256, 16), ("C4 layer1 256->64", 512, 256, 64, 16), ("C4 layer2 128->512", 512, 128, 512, 8), ("C4 layer3 256->1024", 512, 256, 1024, 4),
         ("C2 up lateral 64->128", 64, 64, 128, 32)]


def report(name, flops, times):
    old = {c: t for c, t in times.items() if (c & 0xff) < 20}
    new = {c: t for c, t in times.items() if (c & 0xff) >= 20}
    bo = min(old, key=old.get) if old else None
    line = f"{name:26s} {flops / 1e9:7.2f} GFLOP | 8 waves: " + (f"{old[bo]:7.1f} us (0x{bo:x}, {flops / old[bo] / 1e6:5.0f} TFLOP/s)" if bo is not None else "-")
    line += " | 4 waves: " + "  ".join(f"0x{c:x} {t:7.1f}" for c, t in sorted(new.items()))
    if new and bo is not None:
        bn = min(new, key=new.get)
        line += f" | best x{old[bo] / new[bn]:.2f} ({flops / new[bn] / 1e6:5.0f} TFLOP/s)"
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--filter", default="")
    a = ap.parse_args()
    be = bk.get_backend()
    for name, GH, GW, n_exec, Cin, Cout, bs in CONV3:
        if a.filter not in name:
            continue
        gi, m = grid_tables(1, GH, GW, n_exec)
        feats = torch.randn((n_exec, Cin, bs, bs), device="cuda").contiguous(memory_format=torch.channels_last)
        ring = torch.randn((GH * GW, Cin, 4 * bs), device="cuda")
        w = (torch.randn((Cout, Cin, 3, 3), device="cuda") * (2.0 / (9 * Cin)) ** 0.5).contiguous(memory_format=torch.channels_last)
        wpk = be.pack_conv3x3_weights(w)
        sc = torch.rand(Cin, device="cuda") + 0.5
        pro = (sc, sc * 0.1, True)
        times = {}
        for c in be.conv3x3_candidates(n_exec, Cin, Cout, bs, 4, 1):
            if not (c & 0x2000) or (c & 0x100):
                continue
            be.tune("conv2_cfg", c)
            try:
                times[c] = timeit(lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, pro, None), a.iters)
            finally:
                be.tune("conv2_cfg", -1)
        report(name + " 3x3", 2.0 * n_exec * bs * bs * 9 * Cin * Cout, times)
    for name, n_tiles, Cin, Cout, bs in CONV1:
        if a.filter not in name:
            continue
        x = torch.randn((n_tiles, Cin, bs, bs), device="cuda").contiguous(memory_format=torch.channels_last)
        w = (torch.randn((Cout, Cin, 1, 1), device="cuda") * (2.0 / Cin) ** 0.5)
        wpk = be.pack_conv3x3_weights(w)
        times = {}
        for c in be.conv1x1_candidates(x, Cout, 1):
            if not (c & 0x2000) or (c & 0x100):
                continue
            times[c] = timeit(lambda: be.conv1x1(x, wpk, Cout, None, None, cfg=c, stride=1), a.iters)
        report(name + " 1x1", 2.0 * n_tiles * bs * bs * Cin * Cout, times)


if __name__ == "__main__":
    main()
