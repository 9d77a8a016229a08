#!/bin/bash
set -u
mkdir -p gpurun_out
{
echo "== alone"; python tools/probe_determinism2.py split 4 2>&1 | tail -1
for kind in split mfma32 fp16; do
  echo "== two concurrent processes: $kind"
  if [ $kind = mfma32 ]; then export NGP_FIELD32_FWD=mfma32; else unset NGP_FIELD32_FWD; fi
  python tools/probe_determinism2.py $kind 6 2>&1 | tail -1 &
  python tools/probe_determinism2.py $kind 6 2>&1 | tail -1 &
  wait
done
} 2>&1 | grep -v amdgpu | tee gpurun_out/r3ac_determinism.txt
