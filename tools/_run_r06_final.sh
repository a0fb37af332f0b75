#!/bin/bash
# end-of-round records from the current tree -> gpurun_out/r06/: GPU suite, the driver's default bench line, every config, kernel traces, PMC traffic
set -uo pipefail
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r06/30_gpu_test_suite_final.log; tail -3 gpurun_out/r06/30_gpu_test_suite_final.log
python bench.py > gpurun_out/r06/31_bench_default.json 2> gpurun_out/r06/31_bench_default.err
for c in C2 C3 C3h C4 C5; do
  python bench.py --config $c --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r06/32_bench_$c.json 2> gpurun_out/r06/32_bench_$c.err
  cp gpurun_out/bench_details_$c*.json gpurun_out/r06/ 2>/dev/null
done
python - <<'P'
import json
for c in ("default", "C2", "C3", "C3h", "C4", "C5"):
    try:
        d = json.loads(open(f"gpurun_out/r06/{'31_bench_default' if c == 'default' else '32_bench_' + c}.json").read().strip().splitlines()[-1])
        k = d.get("kernels", {})
        print(c, "fps", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 3), "dense", k.get("dense_gpu_fps"), "x", k.get("speedup_vs_dense_gpu"), "roofline", d.get("roofline", {}).get("frac"),
              "fp16", (k.get("fp16") or {}).get("fps"), "b2", (k.get("batch2") or {}).get("fps"), "refloop", d.get("value_reference_loop"), "tuned_live", (k.get("conv_plan") or {}).get("decisions", {}).get("tuned_live"))
    except Exception as e:
        print(c, "parse failed", e)
P
for c in C2 C3 C3h C4 C5; do bash tools/_run_r06_trace.sh f$c --config $c > /dev/null 2>&1; head -1 gpurun_out/r06/f${c}_frame_breakdown.txt | cut -c1-120; done
bash tools/_run_r06_trace.sh fC2h --config C2 --half > /dev/null 2>&1; head -1 gpurun_out/r06/fC2h_frame_breakdown.txt | cut -c1-120
bash tools/_run_r06_pmc.sh
python tools/kbench_split.py > gpurun_out/r06/06_kbench_split.txt 2>&1
python tools/kbench_topk.py > gpurun_out/r06/07_kbench_topk.txt 2>&1
