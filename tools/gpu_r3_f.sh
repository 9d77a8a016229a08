#!/bin/bash
# round-3 GPU call F: is the 30 us step-boundary gap HW-queue sharing?  (HIP maps streams onto GPU_MAX_HW_QUEUES = 4 hardware queues by default; we have 5 streams)
set -u
mkdir -p gpurun_out
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox"
run() { name=$1; shift; timeout 300 env "$@" python bench.py $Q $EXTRA > gpurun_out/r3f_$name.json 2> gpurun_out/r3f_$name.err; echo "$name rc=$?"; }
EXTRA="" run base NGP_FIELD32_BWD=2
EXTRA="" run hwq8 NGP_FIELD32_BWD=2 GPU_MAX_HW_QUEUES=8
EXTRA="" run noside NGP_FIELD32_BWD=2 NGP_HASH_BWD_NO_SIDE_STREAM=1
EXTRA="" run hwq8_noside NGP_FIELD32_BWD=2 GPU_MAX_HW_QUEUES=8 NGP_HASH_BWD_NO_SIDE_STREAM=1
EXTRA="" run hwq2 NGP_FIELD32_BWD=2 GPU_MAX_HW_QUEUES=2
EXTRA="--config fox" run fox_base X=1
EXTRA="--config fox" run fox_hwq8 GPU_MAX_HW_QUEUES=8 NGP_HASH_BWD_NO_SIDE_STREAM=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3f_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
        print(f.split("r3f_")[1][:-5].ljust(12), d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
