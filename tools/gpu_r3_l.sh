#!/bin/bash
set -u
R=$PWD; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/api && mkdir -p /tmp/api
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --hip-trace -d /tmp/api -o api -- python $R/bench.py --no-cpu-baseline --no-psnr --no-fox --config lego --steps 64 --warmup 32 > /tmp/api/log 2>&1
echo "rc=$?"; tail -2 /tmp/api/log | cut -c1-300
DB=$(find /tmp/api -name "*.db" | head -1)
cd $R && python tools/rocprof_api_timeline.py "$DB" 32 --schema --api > gpurun_out/r3l_api_timeline.txt 2>&1
head -c 6000 gpurun_out/r3l_api_timeline.txt | tail -c 3000
