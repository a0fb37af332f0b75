set -x
mkdir -p gpurun_out
python tools/tune_plans.py --fresh > gpurun_out/tune_all.log 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_all.log 2>&1; echo "rc $?" >> gpurun_out/gpu_all.log
python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
tail -12 gpurun_out/tune_all.log; tail -4 gpurun_out/gpu_all.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_c2.json"))
print(len(json.dumps(d)), d["value"], json.dumps(d["roofline"])[:300])
print(json.dumps(d["kernels"].get("fp16")), d["kernels"].get("speedup_vs_dense_gpu"), d["kernels"].get("batch2"), d["kernels"].get("host_enqueue_ms_per_frame"))
det=json.load(open("gpurun_out/bench_details_C2.json"))
import collections
c=collections.Counter()
for k,v in det["conv_plan_table"].items():
    f=k.split(",")
    if f[0] in ("64","128") and f[4]=="128" and f[5]=="f32" and f[7]=="3":
        c["wide" if (v or 0)>0 and v&0x400 else "wino" if (v or 0)>0 and v&0x200 else "direct" if v is not None else "library"]+=1
print(c)
PY
