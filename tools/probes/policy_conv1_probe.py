#!/usr/bin/env python3
"""Does the library run the policy net's first conv (26 -> 32, 3x3, on a 256x512 map) faster with the input zero-padded to 32 channels?
forward, and forward + backward (input gradient not needed), batch 1 and 2, channels-last fp32."""
import time
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = "cuda"
for N in (1, 2):
    for cin in (26, 32):
        x = torch.randn(N, cin, 256, 512, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.randn(32, cin, 3, 3, device=dev, requires_grad=True)
        wc = w.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        for mode in ("fwd", "fwd+bwd"):
            def step():
                if mode == "fwd":
                    with torch.no_grad():
                        return F.conv2d(x, wc, padding=1)
                y = F.conv2d(x, wc, padding=1)
                y.backward(y.detach())
                wc.grad = None
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            print(f"N={N} cin={cin} {mode}: {(time.perf_counter() - t0) / 50 * 1e6:.1f} us", flush=True)
