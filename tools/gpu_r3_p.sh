#!/bin/bash
# round 3: NeuS module / training tests, the two-rank sharded-sweep test
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_neus_gpu.py -m gpu -q -s -k "trains or module" --durations=8 > gpurun_out/r3p_neus.log 2>&1; echo "neus rc=$?"; grep -E "neus freq|neus hash|passed|failed|Error|assert" gpurun_out/r3p_neus.log | tail -30
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -k "sharded_sweep" --durations=4 > gpurun_out/r3p_sharded.log 2>&1; echo "sharded rc=$?"; tail -30 gpurun_out/r3p_sharded.log
