set -x
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_all.log 2>&1; echo "rc $?" >> gpurun_out/gpu_all.log
python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -- python3 $R/bench.py --no-dense --no-cpu-baseline --upload-variant 0 --steps 2 --warmup 2 > $R/gpurun_out/prof_c2_bench.json 2> $R/gpurun_out/prof_c2.err
cd $R
f=$(find gpurun_out/prof_c2 -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $f 38 1 > gpurun_out/prof_c2_frame_breakdown.txt 2>&1
find gpurun_out/prof_c2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_c2_kernel_stats.csv \;
rm -rf gpurun_out/prof_c2
tail -4 gpurun_out/gpu_all.log; head -40 gpurun_out/prof_c2_frame_breakdown.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_c2.json"))
print(len(json.dumps(d)), d["value"], json.dumps(d["roofline"])[:400])
print(json.dumps(d["kernels"].get("fp16")), d["kernels"].get("speedup_vs_dense_gpu"), d["kernels"].get("batch2"), d["kernels"].get("host_enqueue_ms_per_frame"))
PY
