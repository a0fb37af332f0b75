mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/bench.py --config C3 --no-dense --no-cpu-baseline --steps 2 --warmup 2 > $R/gpurun_out/prof_c3_bench.json 2> $R/gpurun_out/prof_c3.err
cd $R
f=$(find gpurun_out/prof_c3 -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $f 38 1 > gpurun_out/prof_c3_frame_breakdown.txt 2>&1
rm -rf gpurun_out/prof_c3
head -60 gpurun_out/prof_c3_frame_breakdown.txt | cut -c1-150
