#!/bin/bash
# round 3: kernel traces + timeline again (the post-processing anchored on the old forward kernel's name)
set -u
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_lego && mkdir -p /tmp/pf_lego
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_lego -o kt -- python $R/bench.py --no-cpu-baseline --no-psnr --no-fox --no-neus --config lego > /tmp/pf_lego/log 2>&1
grep "^{\"metric" /tmp/pf_lego/log | tail -1 > $R/gpurun_out/r03_lego_bench_under_rocprof.json
KT=$(find /tmp/pf_lego -name "*.db" | head -1)
cd $R && python tools/rocprof_summary.py "$KT" gpurun_out/r03_lego_kernel_trace.md "bench.py --config lego (N=1, 1024 burn-in + 64 warm-up + 200 timed steps), rocprofv3 --kernel-trace --stats" 200 && python tools/rocprof_gaps.py "$KT" 128 > gpurun_out/r03_lego_timeline.txt
python tools/rocprof_timeline.py "$KT" > gpurun_out/r03_lego_timeline_step.txt 2>&1
head -12 gpurun_out/r03_lego_kernel_trace.md | cut -c1-150; head -5 gpurun_out/r03_lego_timeline.txt
