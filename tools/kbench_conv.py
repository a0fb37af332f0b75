#!/usr/bin/env python3
"""Fused halo-gather + 3x3 conv (bc_conv3x3_ring_nhwc, fp32 MFMA) against the sequence it replaces
(bc_pad_ring_nhwc + the library conv on the padded batch) at the packed shapes of the benchmark configs.
usage: python tools/kbench_conv.py [--iters 20] [--filter substr]"""
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

# name, GH, GW, n_exec, Cin, Cout, bs   (C2: 8x16 grid, 67 of 128 tiles executed; C4: 32x64 grid, 589 of 2048)
CASES = [("C2 layer1", 8, 16, 67, 64, 64, 32), ("C2 layer2", 8, 16, 67, 128, 128, 16), ("C2 layer3", 8, 16, 67, 256, 256, 8),
         ("C2 layer4", 8, 16, 67, 512, 512, 4), ("C2 up 1/16", 8, 16, 67, 128, 128, 8), ("C2 up 1/8", 8, 16, 67, 128, 128, 16),
         ("C2 up 1/4", 8, 16, 67, 128, 128, 32), ("C2 batch2 layer1", 8, 16, 134, 64, 64, 32), ("C2 batch2 layer4", 8, 16, 134, 512, 512, 4),
         ("sweep layer1 n=64", 8, 16, 64, 64, 64, 32), ("sweep layer1 n=128", 8, 16, 128, 64, 64, 32), ("sweep layer1 n=32", 8, 16, 32, 64, 64, 32), ("sweep layer1 n=48", 8, 16, 48, 64, 64, 32), ("sweep layer1 n=96", 8, 16, 96, 64, 64, 32),
         ("sweep layer1 n=16", 8, 16, 16, 64, 64, 32),
         ("C4 layer1 rn50", 32, 64, 589, 64, 64, 16), ("C4 layer3 rn50", 32, 64, 589, 256, 256, 4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--filter", default="")
    ap.add_argument("--cfgs", default="", help="comma list of forced conv2 decompositions to time besides auto (e.g. 0,1,2,3,4,5,6,7)")
    ap.add_argument("--no-lib", action="store_true")
    ap.add_argument("--n", type=int, default=0, help="override the executed-tile count of every case")
    ap.add_argument("--dtype", default="float32", choices=["float32", "float16", "bfloat16"])
    ap.add_argument("--stride2", action="store_true", help="time the stride-2 form on the stage-entry layer shapes instead")
    a = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    be = bk.get_backend()
    cases = CASES if not a.stride2 else [("C2 layer2.0.conv1 s2", 8, 16, 64, 64, 128, 32), ("C2 layer3.0.conv1 s2", 8, 16, 64, 128, 256, 16),
                                         ("C2 layer4.0.conv1 s2", 8, 16, 64, 256, 512, 8)]
    stride = 2 if a.stride2 else 1
    for name, GH, GW, n_exec, Cin, Cout, bs in cases:
        if a.filter not in name:
            continue
        N = 2 if "batch2" in name else 1
        if a.n:
            n_exec = a.n
        gi, m = grid_tables(N, GH, GW // N if False else GW, n_exec) if N == 1 else grid_tables(2, GH, GW, n_exec)
        dt = getattr(torch, a.dtype)
        feats = torch.randn((n_exec, Cin, bs, bs), device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        ring = torch.randn((N * GH * GW, Cin, 4 * bs), device="cuda").to(dt)
        w = (torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05).to(dt).contiguous(memory_format=torch.channels_last)
        wpk = be.pack_conv3x3_weights(w)
        sc = torch.rand(Cin, device="cuda") + 0.5
        pro = (sc, sc, True)
        flops = 2.0 * n_exec * (bs // stride) ** 2 * 9 * Cin * Cout

        def lib_path():
            return F.conv2d(be.pad_ring(feats, ring, gi, m, 1, pro), w, stride=stride)

        fused = lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, pro, None, stride=stride)
        if a.no_lib:
            us_halo = us_lib = float("nan")
        else:
            lib_path()   # MIOpen find
            us_halo = timeit(lambda: be.pad_ring(feats, ring, gi, m, 1, pro), a.iters)
            us_lib = timeit(lib_path, a.iters)
        if dt == torch.float32 and stride == 1:
            be.tune("conv_impl", 1)
            us_v1 = timeit(fused, a.iters)
            be.tune("conv_impl", 2)
        else:
            us_v1 = float("nan")
        us_v2 = timeit(fused, a.iters)
        extra = f" auto=c{be.tune_get('conv_last_cfg')}"
        for c in [int(x) for x in a.cfgs.split(",") if x != ""]:
            if c not in be.conv3x3_candidates(n_exec, Cin, Cout, bs, feats.element_size(), stride):
                continue
            be.tune("conv2_cfg", c)
            try:
                extra += f" c{c}={timeit(fused, a.iters):.1f}"
            finally:
                be.tune("conv2_cfg", -1)
        tf = lambda us: flops / us / 1e6
        print(f"{name:18s} ({n_exec},{Cin}->{Cout},{bs}x{bs}) {flops / 1e9:6.2f} GFLOP | halo {us_halo:6.1f} + conv {us_lib - us_halo:6.1f} = {us_lib:6.1f} us"
              f" ({tf(us_lib):5.1f} TF) | v1 {us_v1:6.1f} us ({tf(us_v1) / 157.3:4.0%}) | v2 {us_v2:6.1f} us ({tf(us_v2):5.1f} TF = {tf(us_v2) / 157.3:4.0%} of fp32 MFMA peak; 16-bit peak is 16x)"
              f" | v2 vs lib x{us_lib / us_v2:4.2f}, vs v1 x{us_v1 / us_v2:4.2f}{extra}", flush=True)


if __name__ == "__main__":
    main()
