#!/bin/bash
# marcher check on the GPU box: parity tests of the count passes, then exclusive timings of each pass on both samplings
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "march or count_pass" 2>&1 | tail -3
for cfg in lego fox; do
  for mode in wave g serial; do
    NGP_MARCH_COUNT=$mode timeout 300 python tools/bench_march.py --config $cfg 2>/dev/null | tail -1 | sed "s/^/$cfg $mode: /"
  done
done
