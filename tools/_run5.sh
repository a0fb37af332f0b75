set -x
mkdir -p gpurun_out
python tools/kbench_head.py > gpurun_out/kbench_head.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "head1x1" > gpurun_out/ops3.log 2>&1; echo "rc $?" >> gpurun_out/ops3.log
cat gpurun_out/kbench_head.txt; tail -4 gpurun_out/ops3.log
