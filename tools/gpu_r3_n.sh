#!/bin/bash
# round 3, NeuS: the new GPU tests (compositing kernels, second-order hash kernels, training runs), then the whole GPU suite of HEAD
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_neus_gpu.py -m gpu -q -x -s --durations=8 > gpurun_out/r3n_neus.log 2>&1; echo "neus rc=$?"; tail -25 gpurun_out/r3n_neus.log
timeout 1200 python -m pytest tests -m gpu -q --durations=5 --deselect tests/test_neus_gpu.py > gpurun_out/r3n_tests_gpu.log 2>&1; echo "pytest rc=$?"; tail -9 gpurun_out/r3n_tests_gpu.log
