mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_h -- python3 $R/bench.py --half --no-dense --no-cpu-baseline --upload-variant 0 --steps 2 --warmup 2 > $R/gpurun_out/prof_h_bench.json 2> $R/gpurun_out/prof_h.err
cd $R
f=$(find gpurun_out/prof_h -name "*kernel_trace.csv" | head -1)
python tools/trace_summary.py $f 38 1 > gpurun_out/prof_h_frame_breakdown.txt 2>&1
rm -rf gpurun_out/prof_h
head -48 gpurun_out/prof_h_frame_breakdown.txt | cut -c1-150
