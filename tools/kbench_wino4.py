#!/usr/bin/env python3
"""Winograd F(4x4,3x3) (csrc/conv3x3_wino4.inc, codes 0x1000 | c) against the best existing form (direct / F(2x2,3x3) 16-channel /
shared-transform / wide) of the fused halo + 3x3 conv at the layer shapes of the benchmark configs: every candidate is timed
(hipGraph replay, prologue on, post-ReLU-like inputs); for the F(4x4) codes the maximum error against an fp64 conv of the same
padded input (relative to max(1, |out|max), the bar of tests/test_gpu_ops.py) is printed too.
usage: python tools/kbench_wino4.py [--check-only]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import torch.nn.functional as F
import blockcopy.backend as bk
from kbench import grid_tables, timeit

be = bk.get_backend()
check_only = "--check-only" in sys.argv
CASES = [("layer1", 64, 64, 64, 32), ("layer2", 64, 128, 128, 16), ("layer3", 64, 256, 256, 8), ("up1/8", 64, 128, 128, 16), ("up1/4", 64, 128, 128, 32),
         ("up1/16", 64, 128, 128, 8), ("layer1 n128", 128, 64, 64, 32), ("layer2 n128", 128, 128, 128, 16), ("up1/4 n128", 128, 128, 128, 32),
         ("csp head n38", 38, 768, 256, 32), ("rn50 l1 n512 bs16", 512, 64, 64, 16)]
for name, n, Cin, Cout, bs in CASES:
    n_total = 128 if n <= 128 else 2048
    gi, m = grid_tables(1, 8 if n_total == 128 else 32, 16 if n_total == 128 else 64, n)
    feats = torch.relu(torch.randn((n, Cin, bs, bs), device="cuda")).contiguous(memory_format=torch.channels_last)
    ring = torch.randn((n_total, Cin, 4 * bs), device="cuda")
    w = (torch.randn((Cout, Cin, 3, 3), device="cuda") * (2.0 / (9 * Cin)) ** 0.5).contiguous(memory_format=torch.channels_last)
    wpk = be.pack_conv3x3_weights(w)
    sc = torch.rand(Cin, device="cuda") + 0.5
    pro = (sc, sc * 0.1, True)
    cands = be.conv3x3_candidates(n, Cin, Cout, bs, 4, 1)
    w4 = [c for c in cands if c & 0x1000]
    errs = {}
    if w4:
        ring_a = ring.clone()
        want = F.conv2d(be.pad_ring(feats, ring_a, gi, m, 1, pro).double(), w.double())
        for cfg in w4:
            ring_b = ring.clone()
            got = be.conv3x3_ring(feats, ring_b, wpk, Cout, gi, m, pro, None, cfg=cfg)
            errs[cfg] = ((got.double() - want).abs().max().item() / max(1.0, want.abs().max().item()), torch.equal(ring_a, ring_b))
    if check_only:
        print(f"{name:18s} " + ", ".join(f"{c:#x}: err {e:.2e} ring {'ok' if r else 'DIFFERS'}" for c, (e, r) in errs.items()), flush=True)
        continue
    res = {}
    for cfg in cands:
        f = lambda: be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, pro, None, cfg=cfg)
        res[cfg] = timeit(f, 10)
    old = sorted((t, c) for c, t in res.items() if not c & 0x1000)[:3]
    new = sorted((t, c) for c, t in res.items() if c & 0x1000)
    print(f"{name:18s} best existing: " + ", ".join(f"{c:#x}={t:.1f}" for t, c in old) + " | F(4x4): "
          + ", ".join(f"{c:#x}={t:.1f} (err {errs[c][0]:.1e}{'' if errs[c][1] else ' RING DIFFERS'})" for t, c in new), flush=True)
