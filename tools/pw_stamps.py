#!/usr/bin/env python3
"""Where does a launch of the one-tap (pointwise) form of the fused conv kernel spend its cycles at the decoder's lateral shapes?
In-kernel stamps of wave 0 of every workgroup.  usage: python tools/pw_stamps.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import blockcopy.backend as bk  # noqa: E402
from kbench import timeit  # noqa: E402

be = bk.get_backend()
for name, n, bs, Cin, Cout in [("lateral 1/4", 64, 32, 64, 128), ("lateral 1/8", 64, 16, 128, 128), ("lateral 1/16", 64, 8, 256, 128), ("spp in", 1, 64, 512, 128)]:
    x = torch.randn((n, Cin, bs, bs), device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn((Cout, Cin, 1, 1), device="cuda") * 0.05
    wpk = be.pack_conv3x3_weights(w)
    sh = torch.randn(Cout, device="cuda")
    res = {}
    for c in [None] + be.conv1x1_candidates(x, Cout, 1):
        res[c] = timeit((lambda c_: lambda: be.conv1x1(x, wpk, Cout, None, (None, sh, None, False), cfg=c_))(c), 10)
    print(f"{name}: {n} tiles {bs}x{bs} {Cin}->{Cout}: " + ", ".join(f"{c}={t:.1f}" for c, t in sorted(res.items(), key=lambda kv: kv[1])[:6]), flush=True)
    best = min((t, c) for c, t in res.items() if c is not None and not c & 0x800)[1]
    stamps = torch.zeros(8 * 8192, dtype=torch.int64, device="cuda")
    be.tune_ptr("conv_stamps", stamps)
    be.conv1x1(x, wpk, Cout, None, (None, sh, None, False), cfg=best)
    torch.cuda.synchronize()
    be.tune_ptr("conv_stamps", None)
    s = stamps.view(-1, 8).cpu()
    s = s[s[:, 0] != 0]
    seg = (s[:, 1:6] - s[:, 0:5]).double()
    print(f"   cfg {best}: {s.shape[0]} workgroups; " + " | ".join(f"{nm} {seg[:, i].mean():.0f}" for i, nm in enumerate(["tables", "first stage", "main loop", "reduction", "stores"]))
          + f" | a workgroup lives {float((s[:, 5] - s[:, 0]).double().mean()):.0f} cycles", flush=True)
be.tune("conv2_cfg", -1)
