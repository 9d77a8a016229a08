#!/bin/bash
# kernel timeline of one steady-state step (all streams) of the lego bench under rocprofv3 --kernel-trace
set -u
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && mkdir -p /tmp/tl
timeout 600 rocprofv3 --kernel-trace -d /tmp/tl -o kt -- python $R/bench.py --no-cpu-baseline --no-psnr --no-fox --no-neus --steps 64 --warmup 16 "$@" > /tmp/tl/log 2>&1
DB=$(find /tmp/tl -name "*.db" | head -1)
cd $R && python tools/rocprof_timeline.py "$DB" > gpurun_out/timeline_step.txt; tail -60 gpurun_out/timeline_step.txt
python tools/rocprof_gaps.py "$DB" 48 | head -30
