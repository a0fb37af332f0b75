#!/usr/bin/env python3
"""The network's output stage at C2 (64 of 128 tiles, 32x32 logits tiles, 128 -> 19 channels): k_head1x1 (one launch) against the
three launches it replaces (bc_affine_act + library 1x1 conv + fused scatter+copy), dispatch-attached events, cold caches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import blockcopy.backend as bk  # noqa: E402
from kbench import grid_tables  # noqa: E402


def main():
    be = bk.get_backend()
    flush = torch.zeros(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    for dtype in (torch.float32, torch.float16):
        for (n_exec, n_tot, GH, GW, bs, cin, cout, name) in [(64, 128, 8, 16, 32, 128, 19, "C2 logits 64/128"), (128, 128, 8, 16, 32, 128, 19, "C2 all-active"),
                                                              (24, 128, 8, 16, 32, 128, 19, "C3-like 24/128"), (512, 2048, 32, 64, 16, 128, 19, "C4 logits 512/2048")]:
            gi, m = grid_tables(1, GH, GW, n_exec)
            x = cl(torch.randn((n_exec, cin, bs, bs), device="cuda").to(dtype))
            w = (torch.randn((cout, cin, 1, 1), device="cuda") / cin ** 0.5).to(dtype)
            wpk = be.pack_head1x1_weights(w)
            scale, shift, bias = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda"), torch.randn(cout, device="cuda")
            prev = cl(torch.randn((1, cout, GH * bs, GW * bs), device="cuda").to(dtype))
            out = torch.empty_like(prev)
            for cold in (True, False):
                be.prof_reset()
                for _ in range(12):
                    if cold:
                        flush.add_(1.0)
                    be.prof_enable(["head1x1"])
                    be.head1x1_scatter(x, wpk, cout, (scale, shift, True), bias, gi, m, prev=prev, out=out)
                    be.prof_enable([])
                torch.cuda.synchronize()
                r = be.prof_read("head1x1")
                us = r["total_ms"] * 1e3 / r["launches"]
                mb = r["total_bytes"] / r["launches"] / 1e6
                print(f"{str(dtype)[6:]:8s} {name:20s} {'cold' if cold else 'warm'}: k_head1x1 {us:6.2f} us  {mb:6.1f} MB  {mb / us:5.2f} TB/s  {mb / us / 8:5.1%} of 8 TB/s", flush=True)
            if os.environ.get("KBENCH_HEAD_AB"):
                continue
            # the replaced sequence, timed with torch events (three launches incl. their gaps)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(12):
                flush.add_(1.0)
                a.record()
                y = be.affine_act(x, scale, shift, None, True)
                y = torch.nn.functional.conv2d(y, w, bias.to(dtype))
                be.combine_copy(y, prev, out, gi)
                b.record()
                b.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            ts.sort()
            print(f"{'':8s} {name:20s} cold: affine + library conv + scatter+copy (3 launches, torch events) median {ts[len(ts) // 2]:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
