#!/bin/bash
# usage: tools/_run_r06_trace.sh <tag> [bench args...]   -> gpurun_out/r06/<tag>_frame_breakdown.txt, _frame_launches.txt, _kernel_stats.csv, _rocprof.json
# (rocprofv3 --kernel-trace --stats of `bench.py --steps 4 --warmup 1 ...`: the hipGraph replays of the timed clips)
set -uo pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
tag=$1; shift
rm -rf "/tmp/tr_$tag"
rm -f "gpurun_out/r06/${tag}_frame_breakdown.txt" "gpurun_out/r06/${tag}_frame_launches.txt" "gpurun_out/r06/${tag}_kernel_stats.csv" "gpurun_out/r06/${tag}_rocprof.json"
(cd /tmp && rocprofv3 --kernel-trace --stats -d "/tmp/tr_$tag" --output-format csv -- python3 "$R/bench.py" --steps 4 --warmup 1 --no-dense --no-cpu-baseline --upload-variant 0 --also-half 0 "$@") > "/tmp/tr_$tag.log" 2>&1
t=$(find "/tmp/tr_$tag" -name '*kernel_trace.csv' | head -1); st=$(find "/tmp/tr_$tag" -name '*kernel_stats.csv' | head -1)
if [ -z "$t" ] || [ -z "$st" ]; then echo "trace $tag: rocprofv3 left no kernel trace" >&2; tail -20 "/tmp/tr_$tag.log" >&2; exit 1; fi
python3 tools/trace_summary.py "$t" 38 1 1 "gpurun_out/r06/${tag}_rocprof.json" > "gpurun_out/r06/${tag}_frame_breakdown.txt" 2>&1 || { echo "trace $tag: summary failed" >&2; exit 1; }
cp "$st" "gpurun_out/r06/${tag}_kernel_stats.csv"
cut -c1-60,100-140 "gpurun_out/r06/${tag}_frame_breakdown.txt" | head -45
python3 tools/trace_frame.py "$t" 3 2 > "gpurun_out/r06/${tag}_frame_launches.txt" 2>&1
