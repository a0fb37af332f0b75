#!/usr/bin/env python3
"""One frame of a rocprofv3 --kernel-trace CSV of bench.py, launch by launch: start offset, duration, gap to the previous kernel.

usage: trace_frame.py <kernel_trace.csv> [frame_from_end=3] [n_frames=1]
Frames are delimited by the input-stage launches (k_tile_copy_ind, one per graph-replayed frame)."""
import csv
import re
import sys


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = list(csv.DictReader(open(path)))
    ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    marks = [s for s, e, n in ks if "k_tile_copy_ind<" in n]
    for f in range(nfr):
        fs, fe = marks[-back - f], marks[-back - f + 1]
        sel = [(s, e, n) for s, e, n in ks if fs <= s < fe]
        print(f"frame {-back - f}: {len(sel)} launches, wall {(fe - fs) / 1e3:.1f} us, busy {sum(e - s for s, e, n in sel) / 1e3:.1f} us")
        prev = None
        for s, e, n in sel:
            m = re.search(r"(k_\w+(?:<[^(]*>)?)\(", n)
            name = (m.group(1) if m else n)[:70]
            print(f"  {(s - fs) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  gap {((s - prev) / 1e3 if prev else 0):5.1f}  {name}")
            prev = e


if __name__ == "__main__":
    main()
