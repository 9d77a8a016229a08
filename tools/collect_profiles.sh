#!/bin/bash
# Runs on the GPU box (through gpurun): the three rocprofv3 passes of the bench command, post-processed into gpurun_out/ (copy the results into profiles/).
# usage: tools/collect_profiles.sh <tag>      e.g. r01_e
set -u
TAG=${1:-r01}
R=$PWD
CMD="python $R/bench.py --no-cpu-baseline --no-psnr"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf && mkdir -p /tmp/pf/kt /tmp/pf/fs /tmp/pf/ws
rocprofv3 --kernel-trace --stats -d /tmp/pf/kt -o kt -- $CMD > /tmp/pf/kt.log 2>&1
grep "^{\"metric" /tmp/pf/kt.log | tail -1 > $R/gpurun_out/${TAG}_bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf/fs -o fs -- $CMD --steps 40 --warmup 80 > /tmp/pf/fs.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pf/ws -o ws -- $CMD --steps 40 --warmup 80 > /tmp/pf/ws.log 2>&1
cd $R
KT=$(find /tmp/pf/kt -name "*.db" | head -1); FS=$(find /tmp/pf/fs -name "*.db" | head -1); WS=$(find /tmp/pf/ws -name "*.db" | head -1)
python tools/rocprof_summary.py "$KT" gpurun_out/${TAG}_kernel_trace.md "bench.py (N=1, default warm-up + 200 timed steps), rocprofv3 --kernel-trace --stats" 200
python tools/rocprof_pmc.py "$FS" gpurun_out/${TAG}_pmc_fetch_size.md "bench.py --steps 40 --warmup 80, rocprofv3 --pmc FETCH_SIZE --kernel-trace"
python tools/rocprof_pmc.py "$WS" gpurun_out/${TAG}_pmc_write_size.md "bench.py --steps 40 --warmup 80, rocprofv3 --pmc WRITE_SIZE --kernel-trace"
python tools/rocprof_pmc_json.py "$FS" "$WS" gpurun_out/${TAG}_pmc.json "python bench.py --no-cpu-baseline --no-psnr --steps 40 --warmup 80"
