#!/usr/bin/env python3
"""Does a software prefetch of the scatter+copy's cold operand into the Infinity Cache make the 19.9 MB launch of C2 warm?

Inside a frame `blocks` (5 MB) was written by the kernel before, but `prev` (last frame's output, 10 MB of which the skipped
half is read) is a whole frame of traffic old and comes from HBM: 5.9-6.2 us in-frame against 3.5 us back to back.  Cases, each
timed with the library's dispatch-attached events (what rocprofv3 reports), caches evicted (1 GiB read-modify-write) before every
repetition:
  cold                 flush -> scatter+copy
  producer             flush -> rewrite blocks (stand-in for the logits conv) -> scatter+copy
  producer+prefetch    flush -> rewrite blocks -> read prev (plain cached loads) -> scatter+copy
  ...+work             the same with ~0.1 ms / ~50 MB of unrelated matrix work between the prefetch and the launch
  warm                 back to back
"""
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
    N, C, H, W, bs, n_exec = 1, 19, 256, 512, 32, 64
    gi, m = grid_tables(N, H // bs, W // bs, n_exec)
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)
    blocks = cl(torch.randn((n_exec, C, bs, bs), device="cuda"))
    prev = cl(torch.randn((N, C, H, W), device="cuda"))
    outs = [torch.empty_like(prev) for _ in range(4)]
    h = torch.randn((2048, 2048), device="cuda")
    sink = torch.zeros(1, device="cuda")

    def case(name, steps, reps=15):
        be.prof_reset()
        for r in range(reps):
            for s in steps:
                s(r)
            be.prof_enable(["combine_copy"])
            be.combine_copy(blocks, prev, outs[r % 4], gi)
            be.prof_enable([])
        torch.cuda.synchronize()
        res = be.prof_read("combine_copy")
        us = res["total_ms"] * 1e3 / res["launches"]
        print(f"{name:28s} {us:6.2f} us  {res['total_bytes'] / res['launches'] / us / 1e3:6.0f} GB/s  {res['total_bytes'] / res['launches'] / us / 8e6:5.1%} of 8 TB/s", flush=True)

    fl = lambda r: flush.add_(1.0)
    prod = lambda r: blocks.mul_(1.0)
    pref = lambda r: sink.add_(prev.sum())
    work = lambda r: torch.mm(h, h)
    case("cold", [fl])
    case("producer", [fl, prod])
    case("producer+prefetch", [fl, prod, pref])
    case("prefetch+producer", [fl, pref, prod])
    case("producer+prefetch+work", [fl, prod, pref, work])
    case("prefetch+work+producer", [fl, pref, work, prod])
    case("warm (back to back)", [])


if __name__ == "__main__":
    main()
