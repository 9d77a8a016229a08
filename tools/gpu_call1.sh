#!/bin/bash
# first GPU call of round 2: parity tests + the re-pointed bench (lego fp32, module path) + kernel trace of the same command
set -u
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/c1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/c1_pytest.log
tail -5 gpurun_out/c1_pytest.log
timeout 600 python bench.py > gpurun_out/c1_bench_lego.json 2> gpurun_out/c1_bench_lego.err; echo "bench rc=$?"
tail -c 3000 gpurun_out/c1_bench_lego.json
cd /tmp && rm -rf /tmp/pf && mkdir -p /tmp/pf/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-psnr --no-fox --steps 100 --warmup 16 > /tmp/pf/kt.log 2>&1
cd $R
grep "^{\"metric" /tmp/pf/kt.log | tail -1 > gpurun_out/c1_bench_under_rocprof.json
KT=$(find /tmp/pf/kt -name "*.db" | head -1)
python tools/rocprof_summary.py "$KT" gpurun_out/c1_kernel_trace.md "bench.py lego fp32 (module path), rocprofv3 --kernel-trace --stats" 100 | tail -1
head -40 gpurun_out/c1_kernel_trace.md
