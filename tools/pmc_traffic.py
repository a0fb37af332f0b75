#!/usr/bin/env python3
"""Per-launch HBM traffic of the hand-written kernels from two rocprofv3 --pmc runs of tools/pmc_driver.py (FETCH_SIZE and
WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md 'rocprofv3 PMC slots').

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <pmc_manifest.json> [out.json]
Corrections for gfx950 as the guide prescribes: counter unit = KiB; FETCH_SIZE reports exactly half of the bytes of a
wide coalesced read stream, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  The i-th run of consecutive
launches of one (kernel, grid) pair in dispatch order belongs to the i-th manifest entry (the driver launches each case
REPS times back to back)."""
import csv
import json
import re
import sys


def runs(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name") == counter and "(anonymous namespace)::k_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = []
    for r in rows:
        key = (r["Kernel_Name"], int(r["Grid_Size"]))
        if out and out[-1][0] == key:
            out[-1][1].append(float(r["Counter_Value"]))
        else:
            out.append([key, [float(r["Counter_Value"])]])
    return out


def main():
    fetch, write = runs(sys.argv[1], "FETCH_SIZE"), runs(sys.argv[2], "WRITE_SIZE")
    manifest = json.load(open(sys.argv[3]))
    assert len(fetch) == len(write) == len(manifest), (len(fetch), len(write), len(manifest))
    out = {}
    for (key, fv), (key2, wv), man in zip(fetch, write, manifest):
        assert key == key2, (key, key2)
        m = re.search(r"\(anonymous namespace\)::(k_\w+(?:<[^(]*>)?)\(", key[0])
        rd, wr = 2.0 * 1024 * sum(fv) / len(fv), 1024 * sum(wv) / len(wv)
        alg = man["algorithmic_bytes"]
        out[man["label"]] = dict(kernel=m.group(1) if m else key[0][:60], grid=key[1], launches=len(fv), read_bytes_per_launch=rd,
                                 write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr, algorithmic_bytes_per_launch=alg,
                                 traffic_over_algorithmic=(rd + wr) / alg)
        print(f"{man['label']:62s} {out[man['label']]['kernel'][:44]:44s} read {rd / 1e6:8.2f} MB write {wr / 1e6:8.2f} MB | algorithmic {alg / 1e6:8.2f} MB | x{(rd + wr) / alg:5.2f}")
    if len(sys.argv) > 4:
        cc = out.get("combine_copy C2 logits (1,19,256,512)")
        with open(sys.argv[4], "w") as fjson:
            hd = out.get("head1x1_scatter C2 logits (64,128,32,32) -> (1,19,256,512)")
            json.dump({"k_combine_copy_bytes_per_launch": cc["hbm_bytes_per_launch"] if cc else None,
                       "k_head1x1_bytes_per_launch": hd["hbm_bytes_per_launch"] if hd else None,
                       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of tools/pmc_driver.py (same kernels and shapes as bench.py's default path); FETCH_SIZE x2 (gfx950), KiB units",
                       "kernels": out}, fjson, indent=1)


if __name__ == "__main__":
    main()
