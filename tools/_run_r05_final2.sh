#!/bin/bash
# final records of the round from the current tree: benches + rocprofv3 frame breakdowns of every BASELINE config -> gpurun_out/r05f/
bash tools/_run_r05_final.sh
mkdir -p gpurun_out/r05
for c in C2 C3 C4 C5; do
  if [ $c = C2 ]; then tools/_run_r05_trace.sh f$c > /dev/null 2>&1; else tools/_run_r05_trace.sh f$c --config $c > /dev/null 2>&1; fi
  head -1 gpurun_out/r05/f${c}_frame_breakdown.txt | cut -c1-120
done
tools/_run_r05_trace.sh fC2h --half > /dev/null 2>&1; head -1 gpurun_out/r05/fC2h_frame_breakdown.txt | cut -c1-120
cp gpurun_out/r05/f*_frame_breakdown.txt gpurun_out/r05/f*_frame_launches.txt gpurun_out/r05/fC2_rocprof.json gpurun_out/r05f/ 2>/dev/null
