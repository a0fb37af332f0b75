#!/usr/bin/env python3
"""bc_nms_sorted (one launch: upper-triangular suppression tiles + sweep by the last workgroup) at the detector's nms_pre = 1000 and at the
ABI's maximum, kept sets checked against the oracle.  usage: python tools/kbench_nms.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import blockcopy.backend as bk
import oracle as O
from kbench import timeit

be = bk.get_backend()
rng = np.random.default_rng(0)
for n, spread in [(1000, 400.0), (1000, 1500.0), (1000, 100.0), (300, 200.0), (4096, 800.0)]:
    xy = rng.random((n, 2)) * spread
    wh = rng.random((n, 2)) * 60 + 5
    score = (rng.permutation(n)[:, None] + 1.0) / (n + 1.0)
    dets = np.concatenate([xy, xy + wh, score], 1).astype(np.float32)
    d = torch.from_numpy(dets).cuda()
    order = torch.argsort(d[:, 4], descending=True)
    srt = d[order].contiguous()
    ws = torch.empty(n * ((n + 63) // 64) + n, dtype=torch.int64, device="cuda")
    keep = torch.empty(n, dtype=torch.int32, device="cuda")
    count = torch.zeros(1, dtype=torch.int32, device="cuda")
    fn = lambda: be._check(be.lib.bc_nms_sorted(srt.data_ptr(), n, 0.5, ws.data_ptr(), keep.data_ptr(), count.data_ptr(), be._stream()), "nms")
    us = timeit(fn, 20)
    k = int(count.item())
    want = O.c_nms(dets, 0.5)
    got = order[keep[:k].long()].cpu().numpy()
    print(f"n {n:5d} spread {spread:6.0f}: {us:7.1f} us per call, {k} kept, {'== oracle' if np.array_equal(np.sort(got), np.sort(want)) else 'MISMATCH'}", flush=True)
    # timeline of one launch (100 MHz stamps)
    W = (n + 63) // 64
    tiles = W * (W + 1) // 2
    st = torch.zeros(2 * tiles + 8, dtype=torch.int64, device="cuda")
    be.tune_ptr("conv_stamps", st)
    fn()
    torch.cuda.synchronize()
    be.tune_ptr("conv_stamps", None)
    s = st.cpu().numpy().astype(np.int64)
    t0 = s[0:2 * tiles:2].min()
    rel = lambda v: (v - t0) / 100.0
    print(f"        tiles: first entry 0, last entry {rel(s[0:2 * tiles:2].max()):.2f}, tickets {rel(s[1:2 * tiles:2].min()):.2f} .. {rel(s[1:2 * tiles:2].max()):.2f} us; "
          f"sweeping workgroup: entry {rel(s[2 * tiles]):.2f}, ticket {rel(s[2 * tiles + 1]):.2f}, words in LDS {rel(s[2 * tiles + 2]):.2f}, done {rel(s[2 * tiles + 3]):.2f} us", flush=True)
