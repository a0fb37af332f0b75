#!/bin/bash
# builds measurement variants of the F(4x4) conv kernel (W4_DBG knock-outs in csrc/conv3x3_wino4.inc) next to the library:
# lib/dbg/libbc_w4dbg<N>.so = the shipped objects with part 10 recompiled with -DW4_DBG=N.  usage: tools/w4_knockouts.sh 1 2 4 8 ...
set -euo pipefail
cd "$(dirname "$0")/../blockcopy-video-processing-pytorch_amd"
mkdir -p lib/dbg
# every object of the shipped library except part 10 (the list follows build.py: whatever it compiled is linked)
objs=$(ls lib/obj/*.o | grep -v '/part10\.o$' | tr '\n' ' ')
[ -n "$objs" ] || { echo "no objects under lib/obj: run python build.py first" >&2; exit 1; }
for n in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DBC_PART=10 -DW4_DBG="$n" -c -o "lib/dbg/part10_dbg$n.o" csrc/blockcopy_hip.hip
  hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -o "lib/dbg/libbc_w4dbg$n.so" $objs "lib/dbg/part10_dbg$n.o"
  rm "lib/dbg/part10_dbg$n.o"
done
ls -la lib/dbg
