#!/usr/bin/env python3
"""bc_csp_topk_decode (+ bc_nms_sorted_dev) against the tensor expression it replaces (sigmoid + topk + gathers + exp) on a 256x512 score map
(C5's stride-4 map at 1024x2048), for spread logits, saturated scores and heavily tied logits.  usage: python tools/kbench_topk.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import blockcopy.backend as bk  # noqa: E402
from kbench import timeit  # noqa: E402


def main():
    be = bk.get_backend()
    h, w, k = 256, 512, 1000
    g = torch.Generator(device="cuda").manual_seed(1)
    reg = torch.randn((h, w), device="cuda", generator=g) * 0.3 + 2
    off = torch.randn((2, h, w), device="cuda", generator=g) * 0.3
    for name, cls in [("spread logits N(-4, 2)", torch.randn((h, w), device="cuda", generator=g) * 2 - 4),
                      ("saturated N(10, 8)", torch.randn((h, w), device="cuda", generator=g) * 8 + 10),
                      ("fp16-rounded logits", (torch.randn((h, w), device="cuda", generator=g) * 2 - 4).half().float()),
                      ("256 distinct values", torch.randint(0, 256, (h, w), device="cuda", generator=g).float() / 16 - 8),
                      ("constant map", torch.zeros((h, w), device="cuda"))]:
        dets = torch.empty((k, 5), device="cuda")
        cnt = torch.empty(2, dtype=torch.int32, device="cuda")

        def ours():
            be._check(be.lib.bc_csp_topk_decode(cls.data_ptr(), 0, reg.data_ptr(), off.data_ptr(), off.stride(0), off.stride(2), h * w, k, w, 4, 0.41,
                                                1024, 2048, 0.1, dets.data_ptr(), cnt.data_ptr(), None, be._stream()), "topk")

        def stock():
            s, top = cls.reshape(-1).sigmoid().topk(k)
            return s, reg.reshape(-1)[top].exp(), off.reshape(2, -1)[:, top]

        print(f"{name:24s}: bc_csp_topk_decode {timeit(ours, 50):7.1f} us | sigmoid + topk + gathers + exp (torch) {timeit(stock, 50):7.1f} us", flush=True)


if __name__ == "__main__":
    main()
