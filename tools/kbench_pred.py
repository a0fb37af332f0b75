#!/usr/bin/env python3
"""k_pred3x3 (dense 3x3 conv to 1..4 channels on a channels-last map: the CSP head's prediction convs, bc_pred3x3_nhwc) against the
library conv on the C5 map (1,256,256,512) and smaller ones: dispatch-attached events, caches flushed before every launch (in the
frame the map was just written by the out-of-place combine, 134 MB: it does not sit in L2)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))

import torch  # noqa: E402

import blockcopy.backend as bk  # noqa: E402


def main():
    be = bk.get_backend()
    flush = torch.zeros(256 * 1024 * 1024, dtype=torch.float32, device="cuda")
    torch.backends.cudnn.benchmark = True
    for dtype in (torch.float32, torch.float16):
        for (N, C, H, W, cout) in [(1, 256, 256, 512, 1), (1, 256, 256, 512, 2), (1, 256, 256, 512, 4), (1, 256, 128, 256, 1), (1, 64, 256, 512, 2)]:
            x = torch.randn((N, C, H, W), device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
            w = (torch.randn((cout, C, 3, 3), device="cuda") / (9 * C) ** 0.5).to(dtype)
            b = torch.randn(cout, device="cuda")
            wpk = be.pack_pred3x3_weights(w)
            line = f"{str(dtype)[6:]:8s} {N}x{C}x{H}x{W} -> {cout}: "
            for rows in (1,):
                for cold in (True, False):
                    be.prof_reset()
                    for _ in range(10):
                        if cold:
                            flush.add_(1.0)
                        be.prof_enable(["pred3x3"])
                        be.pred3x3(x, wpk, b, cout)
                        be.prof_enable([])
                    torch.cuda.synchronize()
                    r = be.prof_read("pred3x3")
                    us = r["total_ms"] * 1e3 / r["launches"]
                    mb = r["total_bytes"] / r["launches"] / 1e6
                    line += f" {'cold' if cold else 'warm'} {us:6.1f} us ({mb / us / 8:4.0%} of 8 TB/s) |"
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(8):
                flush.add_(1.0)
                a.record()
                torch.nn.functional.conv2d(x, w, b.to(dtype), padding=1)
                e.record()
                e.synchronize()
                ts.append(a.elapsed_time(e) * 1e3)
            ts.sort()
            print(line + f" library conv {ts[len(ts) // 2]:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
