#!/bin/bash
# round-3 GPU call C: GPU tests (with the ping-pong field backward as default) + A/B against the lock-step kernel
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r3c_tests.log 2>&1; echo "pytest rc=$?"; tail -22 gpurun_out/r3c_tests.log
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox"
run() { name=$1; shift; timeout 300 env "$@" python bench.py $Q $EXTRA > gpurun_out/r3c_$name.json 2> gpurun_out/r3c_$name.err; echo "$name rc=$?"; }
EXTRA="" run pp X=1
EXTRA="" run lockstep NGP_FIELD32_BWD=0
EXTRA="" run pp2 X=1
EXTRA="" run lockstep2 NGP_FIELD32_BWD=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3c_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
        r = d["roofline"] or {}
        pk = d["extra"].get("probe_kernels", {})
        print(f.split("r3c_")[1][:-5].ljust(10), d["value"], d["ms_per_step"], "dom", r.get("kernel"), r.get("avg_launch_ms"), r.get("frac"), r.get("executed_frac"), "| bwd", {k: v.get("avg_launch_ms") for k, v in pk.items() if "field32" in k},
              "stage", (r.get("stage") or {}).get("ms"), d["extra"]["param_signature"][:2])
    except Exception as e:
        print(f, "failed", e)
PY
