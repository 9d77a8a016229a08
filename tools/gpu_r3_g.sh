#!/bin/bash
# round-3 GPU call G: stream / hardware-queue topology sweep (every variant the same binary)
set -u
mkdir -p gpurun_out
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox"
run() { name=$1; cfg=$2; shift; shift; timeout 300 env NGP_FIELD32_BWD=2 BENCH_EXTRA_CFG="$cfg" "$@" python bench.py $Q > gpurun_out/r3g_$name.json 2> gpurun_out/r3g_$name.err; echo "$name rc=$?"; }
run base '{}' X=1
run d1 '{"pipeline_dummy_streams": 1}' X=1
run d2 '{"pipeline_dummy_streams": 2}' X=1
run d3 '{"pipeline_dummy_streams": 3}' X=1
run s1 '{"pipeline_side_streams": 1}' X=1
run s3 '{"pipeline_side_streams": 3}' X=1
run lowprio '{"pipeline_side_priority": 0}' X=1
run hiprio_side '{"pipeline_side_priority": -1}' X=1
run noside_d1 '{"pipeline_dummy_streams": 1}' NGP_HASH_BWD_NO_SIDE_STREAM=1
run depth3 '{"pipeline_depth": 3}' X=1
run depth1 '{"pipeline_depth": 1}' X=1
run base2 '{}' X=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3g_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
        print(f.split("r3g_")[1][:-5].ljust(12), d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
