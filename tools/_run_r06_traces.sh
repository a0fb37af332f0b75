#!/bin/bash
# kernel traces of the benchmark configs (steady-state frames): gpurun_out/r06/<tag>_frame_breakdown.txt / _frame_launches.txt
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/r06
bash tools/_run_r06_trace.sh C4 --config C4 | head -40
bash tools/_run_r06_trace.sh C2 --config C2 | head -30
bash tools/_run_r06_trace.sh C2h --config C2 --half | head -30
bash tools/_run_r06_trace.sh C3h --config C3 --half | head -45
