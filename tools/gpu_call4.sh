#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_oracle_golden.py -m gpu -q -x -k "march or golden" 2>&1 | tail -1
for cfg in lego; do timeout 300 python tools/bench_march.py --config $cfg 2>/dev/null | tail -1 | cut -c1-330; done
NGP_MARCH_COUNT=coop timeout 300 python tools/bench_march.py --config fox 2>/dev/null | tail -1 | cut -c1-330
