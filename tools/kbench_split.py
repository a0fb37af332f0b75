#!/usr/bin/env python3
"""The direct conv form on the 16-bit matrix pipe (codes | 0x2000: fp32 tensors, operands split hi + lo in fp16; csrc/conv3x3_v2.inc BC_F32S)
against every other form of the same layer (fp32 direct, Winograd F(2x2) / F(4x4)) at the packed shapes of the benchmark configs: best time per
family and the error of the split form against an fp64 conv of the same padded input.  usage: python tools/kbench_split.py [--filter substr]"""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import blockcopy.backend as bk  # noqa: E402
from kbench import grid_tables, timeit  # noqa: E402

CASES = [("C2 layer1", 8, 16, 64, 64, 64, 32), ("C2 layer2", 8, 16, 64, 128, 128, 16), ("C2 layer3", 8, 16, 64, 256, 256, 8),
         ("C2 layer4", 8, 16, 64, 512, 512, 4), ("C2 up 1/16", 8, 16, 64, 128, 128, 8), ("C2 up 1/8", 8, 16, 64, 128, 128, 16),
         ("C2 up 1/4", 8, 16, 64, 128, 128, 32), ("C2 layer1 all", 8, 16, 128, 64, 64, 32),
         ("C4 layer1 rn50", 32, 64, 512, 64, 64, 16), ("C4 layer2 rn50", 32, 64, 512, 128, 128, 8), ("C5 head 768", 8, 16, 38, 768, 256, 32)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--filter", default="")
    ap.add_argument("--all", action="store_true", help="print every candidate of the split family")
    a = ap.parse_args()
    be = bk.get_backend()
    for name, GH, GW, n_exec, Cin, Cout, bs in CASES:
        if a.filter not in name:
            continue
        gi, m = grid_tables(1, GH, GW, n_exec)
        feats = torch.randn((n_exec, Cin, bs, bs), device="cuda").contiguous(memory_format=torch.channels_last)
        ring = torch.randn((GH * GW, Cin, 4 * bs), device="cuda")
        w = (torch.randn((Cout, Cin, 3, 3), device="cuda") * (2.0 / (9 * Cin)) ** 0.5).contiguous(memory_format=torch.channels_last)
        wpk = be.pack_conv3x3_weights(w)
        sc = torch.rand(Cin, device="cuda") + 0.5
        pro = (sc, sc * 0.1, True)
        want = F.conv2d(be.pad_ring(feats, ring, gi, m, 1, pro).double(), w.double())
        fams = {"direct fp32": lambda c: c < 0x100, "winograd F(2x2)": lambda c: bool(c & 0x600), "winograd F(4x4)": lambda c: (c & 0x5000) == 0x1000, "F(4x4) split": lambda c: (c & 0x5000) == 0x5000,
                "split 16-bit": lambda c: bool(c & 0x2000)}
        best = {}
        for c in be.conv3x3_candidates(n_exec, Cin, Cout, bs, 4, 1):
            be.tune("conv2_cfg", c)
            try:
                fused = lambda: be.conv3x3_ring(feats, ring.clone(), wpk, Cout, gi, m, pro, None)
                got = fused()
                us = timeit(lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, pro, None), a.iters)
                err = float((got.double() - want).abs().max() / max(1.0, float(want.abs().max())))
            finally:
                be.tune("conv2_cfg", -1)
            if a.all and (c & 0x6000):
                print(f"    0x{c:x}: {us:7.1f} us  {2.0 * n_exec * bs * bs * 9 * Cin * Cout / us / 1e6:6.1f} TFLOP/s", flush=True)
            for fam, pred in fams.items():
                if pred(c) and (fam not in best or us < best[fam][0]):
                    best[fam] = (us, c, err)
        flops = 2.0 * n_exec * bs * bs * 9 * Cin * Cout
        print(f"{name:16s} ({n_exec},{Cin}->{Cout},{bs}x{bs}) {flops / 1e9:6.2f} GFLOP | " +
              " | ".join(f"{fam} {v[0]:6.1f} us (0x{v[1]:x}, err {v[2]:.1e})" for fam, v in best.items()), flush=True)


if __name__ == "__main__":
    main()
