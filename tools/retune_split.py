#!/usr/bin/env python3
"""Re-measure the fp32 entries of blockcopy/plans/gfx950.json against the direct form on the 16-bit matrix pipe (codes | 0x2000: operands split
hi + lo in fp16, csrc/conv3x3_v2.inc BC_F32S): for every fp32 key with a fused-kernel entry -- 3x3 stride 1 / 2 and pointwise -- the entry's
current form and every split candidate are timed stand-alone on tensors of the key's shape (median of 3 timings of 10 launches) and the entry
is switched where a split candidate is faster by more than --margin.  Library-route entries (null) are left alone.
usage: python tools/retune_split.py [--margin 0.03] [--dry-run]"""
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
    ap.add_argument("--max-mb", type=float, default=1500.0)
    ap.add_argument("--family", default="0x2000", help="candidate family, as mask of code bits that must all be set: 0x2000 the split direct form (default), 0x5000 the split F(4x4) form")
    a = ap.parse_args()
    fam = int(a.family, 0)
    be = bk.get_backend()
    doc = json.load(open(PLAN))
    plans = doc["plans"]
    changed = kept = 0
    for text in sorted(plans):
        f = text.split(",")
        n, bs, cin, cout, n_total, dt, stride, ks = int(f[0]), int(f[1]), int(f[2]), int(f[3]), int(f[4]), f[5], int(f[6]), int(f[7])
        cur = plans[text]
        if dt != "f32" or cur is None or n <= 0 or ks not in (1, 3) or (fam == 0x5000 and (ks != 3 or stride != 1)) or cin % 32 or cout % 32 or n * bs * bs * max(cin, cout) * 4 / 1e6 > a.max_mb:
            continue
        feats = torch.relu(torch.randn((n, cin, bs, bs), device="cuda")).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((cout, cin, ks, ks), device="cuda") * (2.0 / (ks * ks * cin)) ** 0.5).contiguous(memory_format=torch.channels_last)
        wpk = be.pack_conv3x3_weights(w)
        sc = torch.rand(cin, device="cuda") + 0.5
        pro = (sc, sc * 0.1, True)
        try:
            if ks == 3:
                if n_total < n or n_total <= 1:
                    continue
                cands = [c for c in be.conv3x3_candidates(n, cin, cout, bs, 4, stride) if (c & fam) == fam]
                gh = 1
                while gh * gh * 2 <= n_total and n_total % (gh * 2) == 0:
                    gh *= 2
                gi, m = grid_tables(1, gh, n_total // gh, n)
                ring = torch.randn((n_total, cin, 4 * bs), device="cuda")
                run = lambda cfg: be.conv3x3_ring(feats, ring, wpk, cout, gi, m, pro, None, cfg=cfg, stride=stride)
            else:
                if not be.conv1x1_supported(feats, w, stride):
                    continue
                cands = [c for c in be.conv1x1_candidates(feats, cout, stride) if (c & fam) == fam]
                run = lambda cfg: be.conv1x1(feats, wpk, cout, pro, None, cfg=cfg, stride=stride)
            if not cands:
                continue
            times = {}
            for cfg in [cur] + cands:
                times[cfg] = sorted(timeit(lambda: run(cfg), 10) for _ in range(3))[1]
        except bk.BlockCopyBackendError as e:
            print(f"{text:38s} skipped: {e}", flush=True)
            continue
        best = min(cands, key=lambda c: times[c])
        if times[best] < (1.0 - a.margin) * times[cur]:
            plans[text] = best
            changed += 1
            tag = "->"
        else:
            kept += 1
            tag = "keep"
        print(f"{text:38s} {cur:#7x} {times[cur]:8.1f} us | best split {best:#7x} {times[best]:8.1f} us {tag}", flush=True)
    print(f"{changed} entries switched, {kept} kept")
    if not a.dry_run:
        doc["note"] = (doc.get("note") or "") + f" | fp32 entries re-measured against the candidates of family {fam:#x} (tools/retune_split.py)"
        tmp = PLAN + ".tmp"
        with open(tmp, "w") as fh:
            json.dump(doc, fh, indent=0, sort_keys=True)
            fh.write("\n")
        os.replace(tmp, PLAN)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        shutil.copy(PLAN, os.path.join(ROOT, "gpurun_out", "gfx950.json"))


if __name__ == "__main__":
    main()
