#!/usr/bin/env python3
"""Re-measure the plan-table entries the Winograd F(4x4,3x3) form (codes 0x1000 | c, csrc/conv3x3_wino4.inc) can serve: for every
fp32 / stride-1 / 3x3 key of blockcopy/plans/gfx950.json whose tile size it covers, the entry's current kernel form and every
F(4x4) candidate are timed stand-alone on tensors of the key's shape (hipGraph replay of 10 launches, median of 3), and the entry
is switched where F(4x4) is faster by more than --margin.  Entries that run the library route (null) are left alone.
usage: python tools/retune_w4.py [--margin 0.03] [--dry-run] [--only-n N ...]"""
import argparse
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import blockcopy.backend as bk  # noqa: E402
from kbench import grid_tables, timeit  # noqa: E402

PLAN = os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd", "blockcopy", "plans", "gfx950.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--margin", type=float, default=0.03)
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--max-mb", type=float, default=1500.0, help="skip shapes whose activations exceed this many MB")
    ap.add_argument("--all-forms", action="store_true", help="time EVERY candidate form of the fused kernel (direct, both F(2x2) forms, F(4x4)), not only F(4x4)")
    a = ap.parse_args()
    be = bk.get_backend()
    doc = json.load(open(PLAN))
    plans = doc["plans"]
    changed, kept, log = 0, 0, []
    for text in sorted(plans):
        f = text.split(",")
        n, bs, cin, cout, n_total, dt, stride, ks = int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]), f[5], int(f[6]), int(f[7])
        cur = plans[text]
        if dt != "f32" or stride != 1 or ks != 3 or cur is None or n_total < n or n_total <= 1:
            continue
        cands = [c for c in be.conv3x3_candidates(n, cin, cout, bs, 4, 1) if (c & 0x1000) or (a.all_forms and c != cur)]
        if not cands or n * bs * bs * max(cin, cout) * 4 / 1e6 > a.max_mb:
            continue
        gh = 1
        while gh * gh * 2 <= n_total and n_total % (gh * 2) == 0:
            gh *= 2
        gi, m = grid_tables(1, gh, n_total // gh, n)
        feats = torch.relu(torch.randn((n, cin, bs, bs), device="cuda")).contiguous(memory_format=torch.channels_last)
        ring = torch.randn((n_total, cin, 4 * bs), device="cuda")
        w = (torch.randn((cout, cin, 3, 3), device="cuda") * (2.0 / (9 * cin)) ** 0.5).contiguous(memory_format=torch.channels_last)
        wpk = be.pack_conv3x3_weights(w)
        sc = torch.rand(cin, device="cuda") + 0.5
        pro = (sc, sc * 0.1, True)
        times = {}
        for cfg in [cur] + cands:
            fn = lambda: be.conv3x3_ring(feats, ring, wpk, cout, gi, m, pro, None, cfg=cfg)
            times[cfg] = sorted(timeit(fn, 10) for _ in range(3))[1]
        best = min(cands, key=lambda c: times[c])
        if times[best] < (1.0 - a.margin) * times[cur]:
            plans[text] = best
            changed += 1
            tag = "->"
        else:
            kept += 1
            tag = "keep"
        log.append(f"{text:38s} {cur:#6x} {times[cur]:8.1f} us | best {'other' if a.all_forms else 'F(4x4)'} {best:#6x} {times[best]:8.1f} us {tag}")
        print(log[-1], flush=True)
        del feats, ring, w, wpk
    print(f"{changed} entries switched, {kept} kept")
    if not a.dry_run:
        tmp = PLAN + ".tmp"
        with open(tmp, "w") as fh:
            json.dump(doc, fh, indent=0, sort_keys=True)
            fh.write("\n")
        os.replace(tmp, PLAN)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        shutil.copy(PLAN, os.path.join(ROOT, "gpurun_out", "gfx950.json"))


if __name__ == "__main__":
    main()
