#!/bin/bash
set -u
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf && mkdir -p /tmp/pf/kt
timeout 600 rocprofv3 --kernel-trace -d /tmp/pf/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-psnr --no-fox --steps 128 --warmup 16 --no-kernel-events > /tmp/pf/kt.log 2>&1
grep "^{\"metric" /tmp/pf/kt.log | tail -1 | cut -c1-200
KT=$(find /tmp/pf/kt -name "*.db" | head -1)
cd $R
python tools/rocprof_gaps.py "$KT" 64
