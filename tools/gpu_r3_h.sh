#!/bin/bash
# round-3 GPU call H: lego-difficulty stand-in (bricks): 40 000-step curve; fp16 two-group backward probe on fox
set -u
mkdir -p gpurun_out
timeout 900 python tools/train_curve.py gpurun_out/r03_train_curve_bricks.md 40000 bricks > gpurun_out/r3h_bricks.log 2>&1; echo "bricks rc=$?"; tail -16 gpurun_out/r03_train_curve_bricks.md
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox --config fox"
run() { name=$1; shift; timeout 300 env "$@" python bench.py $Q > gpurun_out/r3h_$name.json 2> gpurun_out/r3h_$name.err; echo "$name rc=$?"; }
run fox_g0 X=1
run fox_g2 NGP_FIELD_BWD_GROUPS=2
NGP_FIELD_BWD_GROUPS=2 timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "field_bwd or field_fwd_bwd" 2>&1 | tail -3
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3h_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
        pk = d["extra"].get("probe_kernels", {})
        print(f.split("r3h_")[1][:-5].ljust(10), d["value"], d["ms_per_step"], {k: v.get("avg_launch_ms") for k, v in pk.items() if "field" in k})
    except Exception as e:
        print(f, "failed", e)
PY
