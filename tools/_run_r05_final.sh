#!/bin/bash
# final records of the round from the current tree: benches of every BASELINE config -> gpurun_out/r05f/
mkdir -p gpurun_out/r05f
python bench.py > gpurun_out/r05f/bench_C2.json 2> gpurun_out/r05f/bench_C2.err
cp gpurun_out/bench_details_C2.json gpurun_out/r05f/bench_details_C2.json
for c in C3 C3h C4 C5; do
  python bench.py --config $c --steps 10 --warmup 2 > gpurun_out/r05f/bench_$c.json 2>gpurun_out/r05f/bench_$c.err
done
python - <<P
import json
for c in ("C2", "C3", "C3h", "C4", "C5"):
    try:
        d = json.loads(open(f"gpurun_out/r05f/bench_{c}.json").read().strip().splitlines()[-1]); k = d["kernels"]
        print(c, round(d["value"], 1), "x dense", k.get("speedup_vs_dense_gpu"), "dense", k.get("dense_gpu_fps"), "fp16", (k.get("fp16") or {}).get("fps"), "b2", (k.get("batch2") or {}).get("fps"),
              "ref loop", d.get("value_reference_loop"), "roofline", d["roofline"]["frac"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(c, "failed", e)
P
