#!/bin/bash
# round-3 GPU call B: full GPU test suite + probes of the helper-XCD forward map
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3b_tests.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r3b_tests.log
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox"
run() { name=$1; shift; timeout 300 env "$@" python bench.py $Q $EXTRA > gpurun_out/r3b_$name.json 2> gpurun_out/r3b_$name.err; echo "$name rc=$?"; }
EXTRA="" run base X=1
EXTRA="" run help10 NGP_HASH_FWD_BALANCE=2 NGP_HASH_FWD_HELP=0.10
EXTRA="" run help18 NGP_HASH_FWD_BALANCE=2 NGP_HASH_FWD_HELP=0.18
EXTRA="" run help25 NGP_HASH_FWD_BALANCE=2 NGP_HASH_FWD_HELP=0.25
EXTRA="" run base2 X=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3b_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
        r = d["roofline"] or {}
        pk = d["extra"].get("probe_kernels", {})
        print(f.split("r3b_")[1][:-5].ljust(10), d["value"], d["ms_per_step"], "dom", r.get("kernel"), r.get("avg_launch_ms"), "| hash_fwd", {k: v.get("avg_launch_ms") for k, v in pk.items() if k.startswith("k_hash_fwd")},
              "stage", (r.get("stage") or {}).get("ms"))
    except Exception as e:
        print(f, "failed", e)
PY
