#!/usr/bin/env python3
"""Launches each hand-written kernel a few times at the shapes bench.py runs them at (C2) plus the large scatter+copy
shape, with nothing else on the GPU -- the command to put behind `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
(one counter per pass; a PMC pass over the whole bench.py takes >15 min because every MIOpen kernel is serialised)."""
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
    reps = 5
    # fused scatter+copy: C2 logits and C5 head
    for (N, C, H, W, bs, n_exec) in [(1, 19, 256, 512, 32, 64), (1, 256, 256, 512, 32, 64)]:
        gi, m = grid_tables(N, H // bs, W // bs, n_exec)
        blocks = torch.randn((n_exec, C, bs, bs), device="cuda")
        prev = torch.randn((N, C, H, W), device="cuda")
        out = torch.empty_like(prev)
        for _ in range(reps):
            be.combine_copy(blocks, prev, out, gi)
        torch.cuda.synchronize()
    # gather / in-place scatter of the network input
    gi, m = grid_tables(1, 8, 16, 64)
    img = torch.randn((1, 3, 1024, 2048), device="cuda")
    blocks = torch.empty((64, 3, 128, 128), device="cuda")
    for _ in range(reps):
        be.split(blocks, img, m, gi)
        be.combine(blocks, img, gi, m)
    # halo gathers (layer1 and max-pool input shapes)
    for (C, bs, p) in [(64, 32, 1), (64, 64, 1)]:
        feats = torch.randn((64, C, bs, bs), device="cuda")
        ring = torch.randn((128, C, 4 * p * bs), device="cuda")
        for _ in range(reps):
            be.pad_ring(feats, ring, gi, m, p)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
