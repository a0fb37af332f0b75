#!/bin/bash
# PMC traffic of the hand-written kernels from the current tree (separate FETCH_SIZE / WRITE_SIZE passes, --kernel-trace only) -> gpurun_out/r06/
set -uo pipefail
cd "$(dirname "$0")/.." || exit 1
R=$PWD
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
rm -f gpurun_out/r06/traffic_latest.json gpurun_out/r06/40_pmc_traffic.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "/tmp/pmc_$c"
  (cd /tmp && rocprofv3 --kernel-trace --pmc "$c" -d "/tmp/pmc_$c" --output-format csv -- python3 "$R/tools/pmc_driver.py") > "/tmp/pmc_$c.log" 2>&1
done
f=$(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); w=$(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
if [ -z "$f" ] || [ -z "$w" ]; then echo "pmc: a counter pass left no CSV" >&2; tail -5 /tmp/pmc_FETCH_SIZE.log /tmp/pmc_WRITE_SIZE.log >&2; exit 1; fi
python3 tools/pmc_traffic.py "$f" "$w" gpurun_out/pmc_manifest.json gpurun_out/r06/traffic_latest.json > gpurun_out/r06/40_pmc_traffic.txt 2>&1 || { echo "pmc: join failed" >&2; exit 1; }
grep -i "head1x1\|stem7x7\|maxpool\|dilation" gpurun_out/r06/40_pmc_traffic.txt | cut -c1-200
