#!/bin/bash
set -u
mkdir -p gpurun_out
{
NGP_FIELD32_BWD=2 python tools/probe_split_bwd.py
for p in 0 1 2 3 4; do NGP_FIELD32_BWD=3 NGP_SPLIT_PROBE=$p python tools/probe_split_bwd.py; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3u_split_probe.txt
