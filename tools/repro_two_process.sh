#!/bin/bash
# Builds and runs tools/repro_two_process.hip: one copy alone, then two copies concurrently, for every mode.  usage: tools/repro_two_process.sh [seconds] -> stdout
set -u
SEC=${1:-4}
R=$(cd "$(dirname "$0")/.." && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/repro_two_process.hip -I$R/include -L$R/jnerf_amd/csrc -lngp_hip -Wl,-rpath,$R/jnerf_amd/csrc -o /tmp/repro2p || exit 2
for mode in 0 1 2 3; do
  /tmp/repro2p $SEC $mode solo
  /tmp/repro2p $SEC $mode pair-a & p1=$!
  /tmp/repro2p $SEC $mode pair-b & p2=$!
  wait $p1; wait $p2
done
