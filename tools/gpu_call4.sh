#!/bin/bash
set -u
mkdir -p gpurun_out
for c in 0 1 2 3 4 5; do
export NGP_MC_CFG=$c
echo "== cfg $c"
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_oracle_golden.py -m gpu -q -x -k "march or golden" > gpurun_out/c4_pytest_$c.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/c4_pytest_$c.log
for cfg in lego fox; do timeout 300 python tools/bench_march.py --config $cfg 2>/dev/null | tail -1 | cut -c1-330; done
done
