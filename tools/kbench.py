#!/usr/bin/env python3
"""Microbenchmark of the hand-written block-copy kernels on the shapes of the benchmark configs.

For every case: ITERS back-to-back launches between two events (same stream), reported as us/launch and
algorithmic GB/s (bytes per SURVEY.md section 8(d)), next to a plain ``dst.copy_(src)`` of the same byte count
(the practical HBM copy ceiling for that size, launch ramp included).
usage: python tools/kbench.py [--iters 50] [--filter substr] [--json out.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))

import torch  # noqa: E402

import blockcopy.backend as bk  # noqa: E402


def timeit(fn, iters, warm=3):
    """us per launch of `iters` back-to-back launches replayed from a hipGraph (no host launch cost in the number)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters   # us


def grid_tables(N, GH, GW, n_exec, seed=0):
    total = N * GH * GW
    g = torch.zeros(total, dtype=torch.bool)
    g[torch.randperm(total, generator=torch.Generator().manual_seed(seed))[:n_exec]] = True
    gi = torch.empty(total, dtype=torch.int32)
    e = g.cumsum(0).to(torch.int32) - 1
    k = (~g).cumsum(0).to(torch.int32) - 1
    gi = torch.where(g, e, k - total)
    m = torch.nonzero(g).squeeze(1).to(torch.int32)
    return gi.view(N, 1, GH, GW).cuda(), m.cuda()


def copy_baseline(nbytes, iters):
    n = max(1, nbytes // 2 // 4)
    src = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    return timeit(lambda: dst.copy_(src), iters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--filter", default="")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    be = bk.get_backend()
    rows = []

    def report(name, us, nbytes):
        base = copy_baseline(int(nbytes), args.iters)
        rows.append(dict(case=name, us=us, GBps=nbytes / us / 1e3, MB=nbytes / 1e6, copy_us=base, copy_GBps=nbytes / base / 1e3))
        print(f"{name:58s} {nbytes / 1e6:9.2f} MB {us:9.2f} us {nbytes / us / 1e3:8.1f} GB/s | copy_ {base:8.2f} us {nbytes / base / 1e3:8.1f} GB/s", flush=True)

    # ---------------- fused scatter+copy (the headline kernel)
    cc_cases = [("C2 logits", 1, 19, 256, 512, 32, 64), ("C4 logits", 1, 19, 512, 1024, 16, 512), ("C5 head", 1, 256, 256, 512, 32, 64),
                ("C4 input-level", 1, 64, 1024, 2048, 32, 512), ("big 1GB", 1, 128, 1024, 2048, 32, 1024)]
    for dt in (torch.float32, torch.float16):
        E = torch.empty(0, dtype=dt).element_size()
        for name, N, C, H, W, bs, n_exec in cc_cases:
            tag = f"combine_copy {name} ({N},{C},{H},{W}) bs{bs} {n_exec}t {str(dt)[6:]}"
            if args.filter not in tag:
                continue
            gi, m = grid_tables(N, H // bs, W // bs, n_exec)
            blocks = torch.randn((n_exec, C, bs, bs), device="cuda").to(dt)
            prev = torch.randn((N, C, H, W), device="cuda").to(dt)
            out = torch.empty_like(prev)
            us = timeit(lambda: be.combine_copy(blocks, prev, out, gi), args.iters)
            report(tag, us, 2.0 * N * C * H * W * E)
            del blocks, prev, out

    # ---------------- gather / in-place scatter
    t_cases = [("C2 input", 1, 3, 1024, 2048, 128, 64), ("C4 input", 1, 3, 2048, 4096, 64, 512), ("C2 spp-in", 1, 512, 32, 64, 4, 64),
               ("C4 spp-in", 1, 2048, 64, 128, 2, 512), ("C5 head", 1, 256, 256, 512, 32, 64)]
    for dt in (torch.float32, torch.float16):
        E = torch.empty(0, dtype=dt).element_size()
        for name, N, C, H, W, bs, n_exec in t_cases:
            for op in ("split", "combine"):
                tag = f"{op} {name} ({N},{C},{H},{W}) bs{bs} {n_exec}t {str(dt)[6:]}"
                if args.filter not in tag:
                    continue
                gi, m = grid_tables(N, H // bs, W // bs, n_exec)
                blocks = torch.randn((n_exec, C, bs, bs), device="cuda").to(dt)
                img = torch.randn((N, C, H, W), device="cuda").to(dt)
                fn = (lambda: be.split(blocks, img, m, gi)) if op == "split" else (lambda: be.combine(blocks, img, gi, m))
                report(tag, timeit(fn, args.iters), 2.0 * n_exec * C * bs * bs * E)
                del blocks, img

    # ---------------- halo gather over the ring cache (the 21 padded layers of SwiftNet-RN18 at C2 collapse to these shapes)
    h_cases = [("conv1 in", 8, 16, 64, 3, 128, 3), ("maxpool in", 8, 16, 64, 64, 64, 1), ("layer1", 8, 16, 64, 64, 32, 1),
               ("layer2", 8, 16, 64, 128, 16, 1), ("layer3", 8, 16, 64, 256, 8, 1), ("layer4", 8, 16, 64, 512, 4, 1),
               ("up2 blend", 8, 16, 64, 128, 32, 1), ("C4 layer1 rn50", 32, 64, 512, 64, 16, 1), ("C4 layer4 rn50", 32, 64, 512, 512, 2, 1),
               ("C5 head 768ch", 8, 16, 64, 768, 32, 1)]
    for dt in (torch.float32, torch.float16):
        E = torch.empty(0, dtype=dt).element_size()
        for name, GH, GW, n_exec, C, bs, p in h_cases:
            tag = f"pad_ring {name} ({n_exec},{C},{bs},{bs}) p{p} {str(dt)[6:]}"
            if not any(args.filter in t for t in (tag, tag.replace("pad_ring", "pad_ring_nhwc"), tag.replace("pad_ring", "pad(noring)"))):
                continue
            gi, m = grid_tables(1, GH, GW, n_exec)
            feats = torch.randn((n_exec, C, bs, bs), device="cuda").to(dt)
            ring = torch.randn((GH * GW, C, 4 * p * bs), device="cuda").to(dt)
            if args.filter in tag:
                us = timeit(lambda: be.pad_ring(feats, ring, gi, m, p), args.iters)
                report(tag, us, 2.0 * n_exec * C * (bs + 2 * p) ** 2 * E)
            tagn = tag.replace("pad_ring", "pad_ring_nhwc")
            if args.filter in tagn and (C * E) % 16 == 0:
                fcl = feats.contiguous(memory_format=torch.channels_last)
                us = timeit(lambda: be.pad_ring(fcl, ring, gi, m, p), args.iters)
                report(tagn, us, 2.0 * n_exec * C * (bs + 2 * p) ** 2 * E)
                sc = torch.rand(C, device="cuda") + 0.5
                us = timeit(lambda: be.pad_ring(fcl, ring, gi, m, p, (sc, sc, True)), args.iters)
                report(tagn + "+act", us, 2.0 * n_exec * C * (bs + 2 * p) ** 2 * E)
            tag2 = tag.replace("pad_ring", "pad(noring)")
            tr = torch.randn((GH * GW - n_exec, C, bs, bs), device="cuda").to(dt)
            if args.filter in tag2:
                us = timeit(lambda: be.pad(feats, tr, gi, m, p), args.iters)
                report(tag2, us, 2.0 * n_exec * C * (bs + 2 * p) ** 2 * E)
            del feats, ring, tr

    # ---------------- per-tile bilinear x2
    for dt in (torch.float32, torch.float16):
        E = torch.empty(0, dtype=dt).element_size()
        for name, B, C, h in [("up0", 64, 128, 4), ("up1", 64, 128, 8), ("up2", 64, 128, 16)]:
            tag = f"interp {name} ({B},{C},{h},{h})->x2 {str(dt)[6:]}"
            if args.filter not in tag:
                continue
            x = torch.randn((B, C, h, h), device="cuda").to(dt)
            us = timeit(lambda: be.interp_bilinear(x, 2 * h, 2 * h, False, 0.5, 0.5), args.iters)
            report(tag, us, B * C * (h * h + 4 * h * h) * E)

    if args.json:
        with open(args.json, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
