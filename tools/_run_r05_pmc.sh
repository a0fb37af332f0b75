#!/bin/bash
# PMC traffic of the hand-written kernels from the current tree (separate FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/r05f/
mkdir -p gpurun_out/r05f
R=$PWD
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  (cd /tmp && rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c --output-format csv -- python3 $R/tools/pmc_driver.py) > /tmp/pmc_$c.log 2>&1
done
f=$(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1); w=$(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py $f $w gpurun_out/pmc_manifest.json gpurun_out/r05f/traffic_latest.json > gpurun_out/r05f/pmc_traffic_end_of_round5.txt 2>&1
grep -i "F(4x4)\|upsample\|maxpool\|head1x1" gpurun_out/r05f/pmc_traffic_end_of_round5.txt | cut -c1-220
tail -3 /tmp/pmc_FETCH_SIZE.log | cut -c1-200
