#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/probe_determinism.py > gpurun_out/r3z_a.txt 2>&1 &
python tools/probe_determinism.py > gpurun_out/r3z_b.txt 2>&1 &
wait
echo "--- two concurrent processes, split"; tail -6 gpurun_out/r3z_a.txt | cut -c1-200; tail -6 gpurun_out/r3z_b.txt | cut -c1-200
NGP_FIELD32_FWD=mfma32 python tools/probe_determinism.py > gpurun_out/r3z_c.txt 2>&1 &
NGP_FIELD32_FWD=mfma32 python tools/probe_determinism.py > gpurun_out/r3z_d.txt 2>&1 &
wait
echo "--- two concurrent processes, mfma32"; tail -3 gpurun_out/r3z_c.txt | cut -c1-200; tail -3 gpurun_out/r3z_d.txt | cut -c1-200
