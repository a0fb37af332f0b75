#!/usr/bin/env python3
"""Cold-cache timing of the fused scatter+copy against a plain copy of the same bytes.

tools/kbench.py replays the same buffers back to back, so a 20 MB case runs out of the 256 MiB Infinity Cache; inside a
frame the previous output was written a whole frame (hundreds of MB of traffic) earlier and comes from HBM.  Here every
timed launch is preceded by a 1 GiB read-modify-write that evicts L2 and the Infinity Cache, which is the condition
bench.py's in-frame `roofline.achieved` is measured under.  Run under the profiler and read per-kernel durations:

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cold -o cold -- python3 tools/kbench_cold.py
    python tools/kbench_cold.py --summarise gpurun_out/cold

The launch log (gpurun_out/cold/launches.json) lists the timed launches in order; the summary zips it with the trace
(every timed launch directly follows a flush kernel, the warm ones follow another scatter+copy).
"""
from __future__ import annotations

import argparse
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

CASES = [("C2 logits", 1, 19, 256, 512, 32, 64), ("C4 logits", 1, 19, 512, 1024, 16, 512), ("C5 head", 1, 256, 256, 512, 32, 64)]


def run(reps, variants, logdir, busy=False, prof=False):
    import torch

    import blockcopy.backend as bk
    from kbench import grid_tables

    be = bk.get_backend()
    if prof:   # same launch path as bench.py's in-frame measurement (hipExtLaunchKernelGGL with start/stop events)
        be.prof_enable(["combine_copy"])
    flush = torch.zeros(256 * 1024 * 1024, dtype=torch.float32, device="cuda")   # 1 GiB
    log = []

    heavy = torch.randn((8192, 8192), device="cuda") if busy else None

    def cold(label, mb, fn):
        if busy:   # ~10 ms of fp32 MFMA work first: the chip is at its loaded clocks/power state, as inside a frame
            torch.mm(heavy, heavy)
        flush.add_(1.0)
        fn()
        log.append((label, mb))

    for name, N, C, H, W, bs, n_exec in CASES:
        for cl in (False, True):
            gi, m = grid_tables(N, H // bs, W // bs, n_exec)
            blocks = torch.randn((n_exec, C, bs, bs), device="cuda")
            prev = torch.randn((N, C, H, W), device="cuda")
            if cl:
                blocks = blocks.contiguous(memory_format=torch.channels_last)
                prev = prev.contiguous(memory_format=torch.channels_last)
            out = torch.empty_like(prev)
            src = torch.randn(prev.numel(), device="cuda")          # plain copy of the same 2 x |map| bytes
            dst = torch.empty_like(src)
            mb = 2 * prev.numel() * 4 / 1e6
            case = f"{name:10s} {'nhwc' if cl else 'nchw'}"
            for var in variants:
                os.environ["BC_CC_U"], os.environ["BC_CC_NT"] = str(var[0]), str(var[1])
                for _ in range(reps):
                    cold(f"{case} | scatter+copy U={var[0]} NT={var[1]}", mb, lambda: be.combine_copy(blocks, prev, out, gi))
            os.environ.pop("BC_CC_NT", None), os.environ.pop("BC_CC_U", None)
            for _ in range(reps):
                cold(f"{case} | scatter+copy (shipped)", mb, lambda: be.combine_copy(blocks, prev, out, gi))
                cold(f"{case} | dst.copy_(src)", mb, lambda: dst.copy_(src))
                cold(f"{case} | torch.add(src, 1, out=dst)", mb, lambda: torch.add(src, 1.0, out=dst))
            torch.cuda.synchronize()
            extra = ""
            if prof:
                r = be.prof_read("combine_copy")
                extra = f"  in-library event timing: {r['launches']} launches, avg {r['total_ms'] * 1e3 / max(1, r['launches']):.2f} us"
                be.prof_reset()
            print(f"done {case} {mb:.2f} MB{extra}", flush=True)
    os.makedirs(logdir, exist_ok=True)
    json.dump(log, open(os.path.join(logdir, "launches.json"), "w"))


def summarise(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert files, f"no kernel_trace.csv under {d}"
    log = json.load(open(os.path.join(d, "launches.json")))
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
    timed, prev_flush = [], False
    for name, us in seq:
        is_flush = "OnSelf_add" in name and us > 150.0
        if prev_flush and not is_flush:
            timed.append((name, us))
        prev_flush = is_flush
    assert len(timed) == len(log), (len(timed), len(log))
    agg = collections.OrderedDict()
    for (label, mb), (name, us) in zip(log, timed):
        agg.setdefault(label, (mb, []))[1].append(us)
    last_case = None
    for label, (mb, v) in agg.items():
        case, what = label.split(" | ")
        if case != last_case:
            print(f"{case}  {mb:.2f} MB")
            last_case = case
        v.sort()
        med = v[len(v) // 2]
        print(f"    {what:34s} median {med:7.2f} us  min {v[0]:7.2f} us  {mb / med * 1e3:6.0f} GB/s  {mb / med / 8:5.1%} of 8 TB/s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--variants", action="store_true", help="also time the BC_CC_U / BC_CC_NT experiment variants")
    ap.add_argument("--busy", action="store_true", help="run ~10 ms of MFMA work before every flush (loaded clocks)")
    ap.add_argument("--prof", action="store_true", help="launch through the library's event-timed path, as bench.py does")
    ap.add_argument("--logdir", default=os.path.join(ROOT, "gpurun_out", "cold"))
    ap.add_argument("--summarise", default="")
    a = ap.parse_args()
    if a.summarise:
        summarise(a.summarise)
    else:
        run(a.reps, [(u, nt) for u in (1, 2, 4, 8) for nt in (0, 1, 3)] + [(1, 2), (4, 2)] if a.variants else [], a.logdir, a.busy, a.prof)
