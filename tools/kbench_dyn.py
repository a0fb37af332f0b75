#!/usr/bin/env python3
"""Ceiling-sized launches with a device-side executed-tile count (bc_dyn_set) against exact launches of the same decomposition:
what the surplus workgroups of the dynamic graph cost per kernel.  SwiftNet-RN18 layer shapes at C3 (128 tiles, k executed).
usage: python tools/kbench_dyn.py [--k 40] [--iters 20]"""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import blockcopy.backend as bk  # noqa: E402
from blockcopy.core import fusion  # noqa: E402
from kbench import grid_tables, timeit  # noqa: E402

CASES = [("layer1", 64, 64, 32, 1), ("layer2.0 s2", 64, 128, 32, 2), ("layer2", 128, 128, 16, 1), ("layer3.0 s2", 128, 256, 16, 2), ("layer3", 256, 256, 8, 1),
         ("layer4.0 s2", 256, 512, 8, 2), ("layer4", 512, 512, 4, 1), ("up 1/16", 128, 128, 8, 1), ("up 1/8", 128, 128, 16, 1), ("up 1/4", 128, 128, 32, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=40)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="float32")
    a = ap.parse_args()
    be = bk.get_backend()
    GH, GW, total, k = 8, 16, 128, a.k
    dt = getattr(torch, a.dtype)
    n_dev = torch.tensor([k, 0, 0, 0], dtype=torch.int32, device="cuda")
    gi, m = grid_tables(1, GH, GW, k)
    m_full = torch.cat([m, torch.arange(total, dtype=torch.int32, device="cuda")])[:total].contiguous()
    for name, cin, cout, bs, stride in CASES:
        x_full = torch.randn((total, cin, bs, bs), device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        x = x_full[:k].clone().contiguous(memory_format=torch.channels_last)
        ring = torch.randn((total, cin, 4 * bs), device="cuda").to(dt)
        w = (torch.randn((cout, cin, 3, 3), device="cuda") * 0.05).to(dt).contiguous(memory_format=torch.channels_last)
        wpk = be.pack_conv3x3_weights(w)
        key = lambda n: (n, bs, cin, cout, total, dt, stride, 3)
        plan_k, plan_all = fusion._conv_plans.get(key(k)), fusion._conv_plans.get(key(total))
        row = f"{name:12s} {cin:4d}->{cout:4d} {bs:2d}x{bs:<2d} s{stride}"
        for label, cfg in (("plan(k)", plan_k), ("plan(all)", plan_all)):
            if cfg is None:
                row += f" | {label}: library"
                continue
            exact = timeit(lambda: be.conv3x3_ring(x, ring, wpk, cout, gi, m, None, None, cfg=cfg, stride=stride), a.iters)
            dyn = timeit(lambda: be.conv3x3_ring(x_full, ring, wpk, cout, gi, m_full, None, None, cfg=cfg, stride=stride, dyn=(n_dev, total)), a.iters)
            row += f" | {label} c{cfg}: exact {exact:6.1f} us, ceiling+count {dyn:6.1f} us"
        print(row, flush=True)
    # the cheap ones: what a ceiling-sized elementwise / gather launch costs
    x_full = torch.randn((total, 128, 32, 32), device="cuda").contiguous(memory_format=torch.channels_last)
    x = x_full[:k].clone().contiguous(memory_format=torch.channels_last)
    sc = torch.rand(128, device="cuda")
    print(f"affine 128ch 32x32: exact {timeit(lambda: be.affine_act(x, sc, sc, None, True), a.iters):.1f} us, ceiling+count "
          f"{timeit(lambda: be.affine_act(x_full, sc, sc, None, True, dyn=(n_dev, total)), a.iters):.1f} us")


if __name__ == "__main__":
    main()
