#!/usr/bin/env python3
"""Where does a launch of the balanced fused conv (csrc/conv3x3_v2.inc) spend its cycles?  In-kernel s_memtime stamps of
wave 0 of every workgroup: [start, tables done, first stage done, main loop done, reduction done, stores done].
usage: python tools/conv_stamps.py [--n 64] [--cfg -1]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import blockcopy.backend as bk  # noqa: E402
from kbench import grid_tables  # noqa: E402

CASES = [("layer1", 64, 64, 32), ("layer2", 128, 128, 16), ("layer3", 256, 256, 8), ("layer4", 512, 512, 4), ("up1/4", 128, 128, 32)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--cfg", type=int, default=-1)
    ap.add_argument("--half", action="store_true", help="fp16 tensors (16-bit matrix instructions)")
    a = ap.parse_args()
    dt = torch.float16 if a.half else torch.float32
    be = bk.get_backend()
    be.tune("conv2_cfg", a.cfg)
    for name, Cin, Cout, bs in CASES:
        gi, m = grid_tables(1, 8, 16, a.n)
        feats = torch.randn((a.n, Cin, bs, bs), device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
        ring = torch.randn((128, Cin, 4 * bs), device="cuda").to(dt)
        w = (torch.randn((Cout, Cin, 3, 3), device="cuda") * 0.05).to(dt).contiguous(memory_format=torch.channels_last)
        wpk = be.pack_conv3x3_weights(w)
        stamps = torch.zeros(8 * 4096, dtype=torch.int64, device="cuda")
        for _ in range(3):
            be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, None, None)
        torch.cuda.synchronize()
        be.tune_ptr("conv_stamps", stamps)
        be.prof_reset()
        be.prof_enable(["conv3x3"])
        be.conv3x3_ring(feats, ring, wpk, Cout, gi, m, None, None)
        torch.cuda.synchronize()
        be.prof_enable([])
        us = be.prof_read("conv3x3")["total_ms"] * 1e3      # execution time of this one dispatch (events attached to the packet)
        be.tune_ptr("conv_stamps", None)
        cfg = be.tune_get("conv_last_cfg")
        if a.cfg >= 0 and cfg != a.cfg:
            continue      # (the forced decomposition does not cover this layer)
        s = stamps.view(-1, 8).cpu()
        s = s[s[:, 0] != 0]
        t0 = s[:, 0].min()
        rel = (s[:, :6] - t0).double()
        seg = rel[:, 1:] - rel[:, :-1]
        names = ["tables", "first stage", "main loop", "reduction", "stores"]
        print(f"{name} n={a.n} cfg {cfg}: {s.shape[0]} workgroups; start skew max {rel[:, 0].max():.0f} cyc; end: mean {rel[:, 5].mean():.0f} max {rel[:, 5].max():.0f} cyc"
              f"")
        per_wg = float((s[:, 5] - s[:, 0]).double().mean())      # ticks one workgroup lives (one round: that is the whole launch)
        rounds = max(1.0, s.shape[0] / 256.0) if not (a.cfg >= 0 and a.cfg & 0x100) else max(1.0, s.shape[0] / 512.0)
        flops = 2.0 * a.n * bs * bs * 9 * Cin * Cout
        ghz = per_wg * rounds / us / 1e3
        print(f"   dispatch {us:.1f} us; a workgroup lives {per_wg:.0f} s_memtime ticks x {rounds:.2f} rounds => ~{ghz:.2f} GHz shader clock during the launch; "
              f"{flops / us / 1e6:.1f} TFLOP/s = {flops / us / 1e6 / 157.3:.0%} of the 2.4 GHz peak = {flops / us / 1e6 / (157.3 * ghz / 2.4):.0%} of the MFMA issue slots at that clock")
        print("   " + " | ".join(f"{n} {seg[:, i].mean():.0f} (max {seg[:, i].max():.0f})" for i, n in enumerate(names)))
        if (s[:, 6] != 0).all():      # s_memrealtime at entry / exit of every workgroup (100 MHz): the launch's timeline in microseconds
            r0, r1 = s[:, 6].double() / 100.0, s[:, 7].double() / 100.0
            o = r0.min()
            life = (r1 - r0)
            print(f"   timeline (us): first start 0, last start {float((r0 - o).max()):.2f}, first end {float((r1 - o).min()):.2f}, last end {float((r1 - o).max()):.2f}; "
                  f"a workgroup lives {float(life.mean()):.2f} us (min {float(life.min()):.2f}, max {float(life.max()):.2f}) => {per_wg / float(life.mean()) / 1e3:.2f} GHz while it runs")


if __name__ == "__main__":
    main()
