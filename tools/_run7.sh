set -x
mkdir -p gpurun_out
cd tools/probes && hipcc --offload-arch=gfx950 -O3 -o mfma_probe4 mfma_probe4.hip 2>/dev/null; ./mfma_probe4 > ../../gpurun_out/mfma_probe4.txt 2>&1; cd ../..
python tools/kbench_head.py 2>&1 | grep "C2 logits" > gpurun_out/kbench_head.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "head1x1" > gpurun_out/ops3.log 2>&1; echo "rc $?" >> gpurun_out/ops3.log
python bench.py --no-cpu-baseline --upload-variant 0 --also-half 0 > gpurun_out/bench_c2_quick.json 2> gpurun_out/bench_c2.err
cat gpurun_out/mfma_probe4.txt gpurun_out/kbench_head.txt; tail -3 gpurun_out/ops3.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_c2_quick.json"))
print(len(json.dumps(d)), d["value"], json.dumps(d["roofline"])[:900])
PY
