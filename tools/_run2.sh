set -x
mkdir -p gpurun_out
python tools/kbench_prefetch.py > gpurun_out/kbench_prefetch.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "graph_node or c2_full_size or graph_replay" > gpurun_out/e2e2.log 2>&1; echo "rc $?" >> gpurun_out/e2e2.log
python tools/tune_plans.py --configs C3 C3h C4 C4h C5 C5h > gpurun_out/tune2.log 2>&1
python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -- python3 $R/bench.py --no-dense --no-cpu-baseline --upload-variant 0 --steps 2 --warmup 2 > $R/gpurun_out/prof_c2_bench.json 2> $R/gpurun_out/prof_c2.err
cd $R
f=$(find gpurun_out/prof_c2 -name "*kernel_trace.csv" | head -1)
python - $f <<'PY' > gpurun_out/prof_c2_combine_ind.txt
import csv, sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "k_combine_copy_ind" in r["Kernel_Name"]]
d=sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows)
print("k_combine_copy_ind launches", len(d), "median us", d[len(d)//2], "min", d[0], "max", d[-1], "mean", sum(d)/len(d))
PY
python tools/trace_summary.py $f 38 1 > gpurun_out/prof_c2_frame_breakdown.txt 2>&1
find gpurun_out/prof_c2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_c2_kernel_stats.csv \;
rm -rf gpurun_out/prof_c2
cat gpurun_out/kbench_prefetch.txt; tail -3 gpurun_out/e2e2.log; cat gpurun_out/tune2.log | tail -8; cat gpurun_out/prof_c2_combine_ind.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_c2.json"))
print(d["value"], json.dumps(d["roofline"])[:900])
print(json.dumps(d["kernels"].get("fp16")), d["kernels"].get("speedup_vs_dense_gpu"))
d=json.load(open("gpurun_out/prof_c2_bench.json"))
print(d["value"], json.dumps(d["roofline"])[:600])
PY
