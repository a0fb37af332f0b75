set -x
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "rl_policy or device_policy or c3_rl" > gpurun_out/rl.log 2>&1; echo "rc $?" >> gpurun_out/rl.log
tail -5 gpurun_out/rl.log
python tools/tune_plans.py --configs C3 C3h > gpurun_out/tune_c3.log 2>&1
tail -4 gpurun_out/tune_c3.log
python bench.py --config C3 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err
BLOCKCOPY_GRAPH_TRAIN=0 python bench.py --config C3 --no-dense --no-cpu-baseline > gpurun_out/bench_c3_eager_train.json 2> gpurun_out/bench_c3b.err
python - <<'PY'
import json
for f in ("gpurun_out/bench_c3.json", "gpurun_out/bench_c3_eager_train.json"):
    try:
        d=json.load(open(f))
        print(f, d["value"], d["config"]["exec_fraction"], d["kernels"].get("host_enqueue_ms_per_frame"), d["kernels"].get("dense_gpu_fps"), json.dumps(d["kernels"].get("fp16")))
    except Exception as e:
        print(f, "failed", e)
PY
tail -5 gpurun_out/bench_c3.err
