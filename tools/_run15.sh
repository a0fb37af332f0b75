mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in C5 C4; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$cfg -- python3 $R/bench.py --config $cfg --no-dense --no-cpu-baseline --steps 2 --warmup 2 > $R/gpurun_out/prof_${cfg}_bench.json 2> $R/gpurun_out/prof_$cfg.err
f=$(find $R/gpurun_out/prof_$cfg -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_summary.py $f 38 1 $([ $cfg = C5 ] && echo 1 || echo 2) > $R/gpurun_out/prof_${cfg}_frame_breakdown.txt 2>&1
rm -rf $R/gpurun_out/prof_$cfg
head -34 $R/gpurun_out/prof_${cfg}_frame_breakdown.txt | cut -c1-150
done
