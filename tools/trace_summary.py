#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-frame kernel time inside the timed clips.

usage: trace_summary.py <kernel_trace.csv> <frames_in_window> [skip_last_frames] [gathers_per_frame] [out.json]
The window is delimited by the input-stage launches of the frames (k_tile_copy_ind, one per graph-replayed frame); traces without
them: by the gather (k_tiles<.., true>) launches, `gathers_per_frame` per frame.  out.json: the per-kernel averages of the window in machine-readable form (profiles/rocprof_latest.json
is what bench.py reports beside its own event-based duration of the roofline kernel)."""
import collections
import csv
import json
import re
import sys


def main():
    path, n_frames = sys.argv[1], int(sys.argv[2])
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = list(csv.DictReader(open(path)))
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    # frame markers: the input stage of a graph-replayed frame (k_tile_copy_ind, one per frame) where the trace has it, else the gathers
    marks = [s for s, e, n in ks if "k_tile_copy_ind<" in n]
    if marks and len(sys.argv) > 4:
        sys.argv[4] = "1"
    if not marks:
        marks = [s for s, e, n in ks if "k_tiles<" in n and ", true" in n]
    per_frame = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    fe = marks[-per_frame * skip] if skip else ks[-1][1] + 1
    fs = marks[-per_frame * (skip + n_frames)]
    sel = [(s, e, n) for s, e, n in ks if fs <= s < fe]
    busy = sum(e - s for s, e, n in sel)
    print(f"window: {n_frames} frames, wall {(fe - fs) / 1e6 / n_frames:.3f} ms/frame, GPU busy {busy / 1e6 / n_frames:.3f} ms/frame, "
          f"{len(sel) / n_frames:.1f} launches/frame")
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n in sel:
        a = agg[n[:100]]
        a[0] += e - s
        a[1] += 1
    print(f"{'kernel':100s} {'ms/frame':>9s} {'calls/frm':>9s} {'avg_us':>8s}")
    for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:40]:
        print(f"{n:100s} {t / 1e6 / n_frames:9.3f} {c / n_frames:9.1f} {t / c / 1e3:8.1f}")
    if len(sys.argv) > 5:
        full = collections.defaultdict(lambda: [0, 0])
        for s, e, n in sel:
            m = re.search(r"(k_\w+(?:<[^(]*>)?)\(", n)
            a = full[m.group(1) if m else n[:80]]
            a[0] += e - s
            a[1] += 1
        with open(sys.argv[5], "w") as f:
            json.dump({"source": "rocprofv3 --kernel-trace of `python bench.py` (hipGraph replays), summarised by tools/trace_summary.py",
                       "window_frames": n_frames, "wall_ms_per_frame": (fe - fs) / 1e6 / n_frames, "gpu_busy_ms_per_frame": busy / 1e6 / n_frames,
                       "launches_per_frame": len(sel) / n_frames,
                       "kernels": {n: {"avg_us": t / c / 1e3, "calls_per_frame": c / n_frames} for n, (t, c) in sorted(full.items(), key=lambda x: -x[1][0])}},
                      f, indent=1)


if __name__ == "__main__":
    main()
