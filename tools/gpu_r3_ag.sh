#!/bin/bash
# round 3: full GPU suite after the test restructuring; A/B of the split backward without spills
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r3ag_tests_gpu.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/r3ag_tests_gpu.log | cut -c1-300
NGP_FIELD32_BWD=2 python tools/probe_split_bwd.py 2>&1 | tail -1
NGP_FIELD32_BWD=3 python tools/probe_split_bwd.py 2>&1 | tail -1
