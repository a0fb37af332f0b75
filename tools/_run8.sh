set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "every_decomposition" > gpurun_out/ops8.log 2>&1; echo "rc $?" >> gpurun_out/ops8.log
tail -5 gpurun_out/ops8.log
timeout 600 python tools/kbench_wino.py > gpurun_out/kbench_wino.txt 2>&1
cat gpurun_out/kbench_wino.txt
