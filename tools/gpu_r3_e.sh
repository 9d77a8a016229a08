#!/bin/bash
# round-3 GPU call E: two-group field backward (parity tests of the kernel under NGP_FIELD32_BWD=2, then A/B), faster fused MLP tail, timeline of the current tree
set -u
mkdir -p gpurun_out
NGP_FIELD32_BWD=2 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_train_gpu.py -m gpu -q -x -k "field32 or fast_path or training_converges or fp32_fused" > gpurun_out/r3e_tests.log 2>&1; echo "pytest(2g) rc=$?"; tail -5 gpurun_out/r3e_tests.log
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox"
run() { name=$1; shift; timeout 300 env "$@" python bench.py $Q $EXTRA > gpurun_out/r3e_$name.json 2> gpurun_out/r3e_$name.err; echo "$name rc=$?"; }
EXTRA="" run lock X=1
EXTRA="" run twog NGP_FIELD32_BWD=2
EXTRA="" run lock2 X=1
EXTRA="" run twog2 NGP_FIELD32_BWD=2
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3e_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
        r = d["roofline"] or {}
        pk = d["extra"].get("probe_kernels", {})
        print(f.split("r3e_")[1][:-5].ljust(8), d["value"], d["ms_per_step"], "dom", r.get("kernel"), r.get("avg_launch_ms"), "| ", {k: v.get("avg_launch_ms") for k, v in pk.items() if "field32" in k or "mlp32" in k}, d["extra"]["param_signature"][:2])
    except Exception as e:
        print(f, "failed", e)
PY
bash tools/gpu_timeline.sh > gpurun_out/r3e_timeline.txt 2>&1; head -30 gpurun_out/r3e_timeline.txt | tail -28
