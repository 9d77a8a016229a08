#!/bin/bash
set -u
mkdir -p gpurun_out
{
python tools/probe_determinism.py 2>&1 | tail -8
for i in 1 2; do python bench.py --no-fox --no-cpu-baseline --no-neus --no-psnr --no-kernel-events --steps 100 --burn-in 256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('single process split', d['extra']['param_signature'][:3], d['loss'])"; done
for i in 1 2; do BENCH_EXTRA_CFG='{"pipeline_sampling": false}' python bench.py --no-fox --no-cpu-baseline --no-neus --no-psnr --no-kernel-events --steps 100 --burn-in 256 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('single process split, no pipelined sampling', d['extra']['param_signature'][:3], d['loss'])"; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3w_determinism.txt
