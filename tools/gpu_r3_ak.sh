#!/bin/bash
mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_neus_gpu.py -m gpu -q -s -k "trains" 2>&1 | grep -E "neus freq|neus hash|passed|failed" | cut -c1-700 | tee gpurun_out/r3ak_neus.txt
