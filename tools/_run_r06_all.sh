#!/bin/bash
# full GPU suite + benches of every config
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r06/10_gpu_test_suite.log
for c in C2 C3 C3h C4 C5; do
  python bench.py --config $c --steps 6 --warmup 2 $( [ "$c" = "C2" ] || echo "--no-cpu-baseline" ) > gpurun_out/r06/11_bench_$c.json 2> gpurun_out/r06/11_bench_$c.err
  cp gpurun_out/bench_details_$c*.json gpurun_out/r06/ 2>/dev/null
done
tail -4 gpurun_out/r06/10_gpu_test_suite.log
python - <<'P'
import json
for c in ("C2", "C3", "C3h", "C4", "C5"):
    try:
        d = json.loads(open(f"gpurun_out/r06/11_bench_{c}.json").read().strip().splitlines()[-1])
        k = d.get("kernels", {})
        print(c, "fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "dense", k.get("dense_gpu_fps"), "x", k.get("speedup_vs_dense_gpu"), "roofline", d.get("roofline", {}).get("frac"), "fp16", (k.get("fp16") or {}).get("fps"))
    except Exception as e:
        print(c, "parse failed", e)
P
