#!/usr/bin/env python3
"""Launch the fused conv kernel a few times at one C2 layer shape (the command to put behind rocprofv3 --pmc)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

import blockcopy.backend as bk  # noqa: E402
from kbench import grid_tables  # noqa: E402
from kbench_conv import CASES  # noqa: E402


def main():
    be = bk.get_backend()
    want = sys.argv[1] if len(sys.argv) > 1 else "C2 layer1"
    for name, GH, GW, n_exec, Cin, Cout, bs in CASES:
        if name != want:
            continue
        N = 2 if "batch2" in name else 1
        gi, m = grid_tables(N, GH, GW, n_exec)
        feats = torch.randn((n_exec, Cin, bs, bs), device="cuda").contiguous(memory_format=torch.channels_last)
        ring = torch.randn((N * GH * GW, Cin, 4 * bs), device="cuda")
        wpk = be.pack_conv3x3_weights(torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05)
        for _ in range(10):
            be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, None, None)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
