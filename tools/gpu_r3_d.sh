#!/bin/bash
# round-3 GPU call D: device-flag batch hand-over A/B (lego + fox + real fox), quick subset of the GPU tests that exercise the training loop
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_trajectory_gpu.py -m gpu -q -x --durations=3 > gpurun_out/r3d_tests.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r3d_tests.log
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox"
run() { name=$1; shift; timeout 300 env "$@" python bench.py $Q $EXTRA > gpurun_out/r3d_$name.json 2> gpurun_out/r3d_$name.err; echo "$name rc=$?"; }
EXTRA="" run flag X=1
EXTRA="" run event BENCH_EXTRA_CFG='{"flag_handover": false}'
EXTRA="" run flag2 X=1
EXTRA="" run event2 BENCH_EXTRA_CFG='{"flag_handover": false}'
EXTRA="--config fox" run fox_flag X=1
EXTRA="--config fox" run fox_event BENCH_EXTRA_CFG='{"flag_handover": false}'
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3d_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
        r = d["roofline"] or {}
        print(f.split("r3d_")[1][:-5].ljust(10), d["value"], d["ms_per_step"], "dom", r.get("kernel"), r.get("avg_launch_ms"), "stage", (r.get("stage") or {}).get("ms"), d["extra"]["param_signature"][:2])
    except Exception as e:
        print(f, "failed", e)
PY
