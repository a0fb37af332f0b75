set -x
mkdir -p gpurun_out
python tools/tune_plans.py --fresh --configs C2 C2h C2b2 > gpurun_out/tune.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_e2e.py -x -q -m gpu > gpurun_out/e2e.log 2>&1; echo "e2e rc $?" >> gpurun_out/e2e.log
python bench.py > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -- python3 $R/bench.py --no-dense --no-cpu-baseline --upload-variant 0 --steps 2 --warmup 2 > $R/gpurun_out/prof_c2_bench.json 2> $R/gpurun_out/prof_c2.err
cd $R
f=$(find gpurun_out/prof_c2 -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $f 38 1 > gpurun_out/prof_c2_frame_breakdown.txt 2>&1
find gpurun_out/prof_c2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_c2_kernel_stats.csv \;
rm -rf gpurun_out/prof_c2
tail -3 gpurun_out/tune.log; tail -5 gpurun_out/e2e.log; head -c 1500 gpurun_out/bench_c2.json; head -30 gpurun_out/prof_c2_frame_breakdown.txt
