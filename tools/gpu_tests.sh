#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/tests_gpu.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/tests_gpu.log
