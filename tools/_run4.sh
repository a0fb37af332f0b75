set -x
mkdir -p gpurun_out
python tools/kbench_head.py > gpurun_out/kbench_head.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "head1x1 or split_combine" > gpurun_out/ops3.log 2>&1; echo "rc $?" >> gpurun_out/ops3.log
timeout 900 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu -k "graph_node or benched or full_size" > gpurun_out/e2e3.log 2>&1; echo "rc $?" >> gpurun_out/e2e3.log
cat gpurun_out/kbench_head.txt; tail -4 gpurun_out/ops3.log; tail -4 gpurun_out/e2e3.log
