#!/usr/bin/env python3
"""Network-input stage: bc_stem7x7s2_nhwc (window gather + 7x7 stem conv, one launch) against the route it replaces
(bc_split + bc_pad_ring p=3 + layout copy + library conv) at the C2 shape.  usage: python tools/kbench_stem.py [--dtype float16]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--n", type=int, default=64)
    a = ap.parse_args()
    dt = getattr(torch, a.dtype)
    be = bk.get_backend()
    torch.backends.cudnn.benchmark = True
    gi, m = grid_tables(1, 8, 16, a.n)
    state = torch.randn((1, 3, 1024, 2048), device="cuda").to(dt)
    w = (torch.randn((64, 3, 7, 7), device="cuda") * 0.05).to(dt).contiguous(memory_format=torch.channels_last)
    wpk = be.pack_stem7x7_weights(w)
    blocks = torch.empty((a.n, 3, 128, 128), device="cuda", dtype=dt)
    ring = torch.zeros((128, 3, 4 * 3 * 128), device="cuda", dtype=dt)
    sh = torch.rand(64, device="cuda")

    def old():
        be.split(blocks, state, m, gi)
        return F.conv2d(be.pad_ring(blocks, ring, gi, m, 3), w, stride=2)

    old()
    us_old = timeit(old, 20)
    be.tune("stem_min_lds", 0)
    us_nat = timeit(lambda: be.stem7x7(state, wpk, m, 128, (None, sh, None, True)), 20)
    be.tune("stem_min_lds", 56 * 1024)
    us_2 = timeit(lambda: be.stem7x7(state, wpk, m, 128, (None, sh, None, True)), 20)
    be.tune("stem_min_lds", 84 * 1024)
    us_new = timeit(lambda: be.stem7x7(state, wpk, m, 128, (None, sh, None, True)), 20)
    print(f"   LDS request (fp32 only): natural occupancy {us_nat:.1f} us | >= 56 KB (2 WG/CU) {us_2:.1f} us | >= 84 KB (1 WG/CU, shipped for fp32) {us_new:.1f} us")
    flops = 2.0 * a.n * 64 * 64 * 147 * 64
    print(f"stem C2 n={a.n} {a.dtype}: split + halo(p=3) + library conv {us_old:.1f} us | bc_stem7x7s2_nhwc {us_new:.1f} us "
          f"({flops / us_new / 1e6:.1f} TFLOP/s) | x{us_old / us_new:.2f}")


if __name__ == "__main__":
    main()
