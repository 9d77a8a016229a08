#!/bin/bash
# round-3 GPU call A: full GPU test suite, then A/B probes of the round-3 changes (every variant is the same binary, selected by environment / flags)
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r3a_tests.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r3a_tests.log
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox"
run() { name=$1; shift; timeout 300 env "$@" python bench.py $Q $EXTRA > gpurun_out/r3a_$name.json 2> gpurun_out/r3a_$name.err; echo "$name rc=$?"; }
EXTRA="" run base X=1
EXTRA="" run nobalance NGP_HASH_FWD_BALANCE=0
EXTRA="" run light05 NGP_HASH_FWD_LIGHT=0.05
EXTRA="" run light25 NGP_HASH_FWD_LIGHT=0.25
EXTRA="" run noabsmax NGP_NO_FUSED_ABSMAX=1
EXTRA="" run nomlptail NGP_NO_FUSED_MLP_TAIL=1
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
EXTRA="--force-dist" run dist X=1
EXTRA="--force-dist --dp-overlap" run distoverlap X=1
unset MASTER_ADDR MASTER_PORT RANK WORLD_SIZE LOCAL_RANK
EXTRA="--config fox" run fox X=1
EXTRA="--config fox" run fox_nobalance NGP_HASH_FWD_BALANCE=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3a_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
        r = d["roofline"] or {}
        k = r.get("ms_per_step_by_kernel", {})
        print(f.split("r3a_")[1][:-5].ljust(14), d["value"], d["ms_per_step"], "dom", r.get("kernel"), r.get("avg_launch_ms"), "| hash_fwd", (d["extra"].get("probe_kernels", {}).get("k_hash_fwd") or {}).get("avg_launch_ms"),
              "stage", (r.get("stage") or {}).get("ms"), "native", d["extra"].get("native_step"))
    except Exception as e:
        print(f, "failed", e)
PY
