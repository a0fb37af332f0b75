#!/usr/bin/env python3
"""Measure the conv plan table of the BASELINE configs on this GPU and write it to blockcopy/plans/gfx950.json.

A plan decides which kernel FORM a padded 3x3 (or pointwise) conv layer runs in -- halo gather + library conv, the direct MFMA
form or the Winograd form of the fused kernel, and which decomposition -- per layer shape and executed-tile count
(blockcopy/core/fusion.py).  The forms differ by fp32 rounding, so a fixed table is what makes two runs (or two ranks) compute
the same logits; this tool is how the committed table was produced:

    python tools/tune_plans.py [--out FILE] [--configs C2 C2h C2b2 C3 C3h C4 C4h C5 C5h]

Every config runs as its own `bench.py` process (one warm-up clip, one timed clip, tuner live with 7 timing repetitions per
candidate instead of 3) that loads the table so far and writes it back with its additions (`--save-plan`).  A copy goes to
gpurun_out/ so that it travels back from a gpurun box."""
import argparse
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_OUT = os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd", "blockcopy", "plans", "gfx950.json")
CONFIGS = {
    "C2": ["--config", "C2"], "C2h": ["--config", "C2", "--half"], "C2b2": ["--config", "C2", "--batch", "2"],
    "C3": ["--config", "C3"], "C3h": ["--config", "C3h"],
    "C4": ["--config", "C4"], "C4h": ["--config", "C4", "--half"],
    "C5": ["--config", "C5"], "C5h": ["--config", "C5", "--half"],
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=DEFAULT_OUT)
    ap.add_argument("--configs", nargs="*", default=list(CONFIGS))
    ap.add_argument("--fresh", action="store_true", help="start from an empty table instead of extending --out")
    ap.add_argument("--reps", type=int, default=7)
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    if args.fresh or not os.path.exists(args.out):
        with open(args.out, "w") as f:
            json.dump({"format": 1, "note": "empty", "plans": {}}, f)
    env = dict(os.environ, BLOCKCOPY_CONV_PLAN=args.out, BLOCKCOPY_CONV_TUNE="1", BLOCKCOPY_CONV_TUNE_REPS=str(args.reps))
    for name in args.configs:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-dense", "--no-cpu-baseline",
               "--upload-variant", "0", "--save-plan", args.out, "--details", os.path.join(ROOT, "gpurun_out", f"tune_details_{name}.json")] + CONFIGS[name]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        with open(args.out) as f:
            n = len(json.load(f)["plans"])
        line = next((l for l in r.stdout.splitlines() if l.lstrip().startswith("{")), None)
        fps = json.loads(line)["value"] if line else None
        print(f"{name}: rc {r.returncode}, table now {n} entries, fps {fps}", flush=True)
        if r.returncode != 0:
            print(r.stderr[-3000:], file=sys.stderr)
    shutil.copy(args.out, os.path.join(ROOT, "gpurun_out", os.path.basename(args.out)))


if __name__ == "__main__":
    main()
