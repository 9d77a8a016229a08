#!/bin/bash
# round 3: NeuS kernel tests (no training runs), then the two-stream overlap probe
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_neus_gpu.py -m gpu -q -s -k "not trains" --durations=8 > gpurun_out/r3o_neus.log 2>&1; echo "neus rc=$?"; tail -40 gpurun_out/r3o_neus.log
timeout 300 python tools/probe_overlap.py > gpurun_out/r3o_overlap.txt 2>&1; echo "overlap rc=$?"; cat gpurun_out/r3o_overlap.txt
