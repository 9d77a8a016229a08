#!/bin/bash
# round-3 GPU call I: occupied-bounds culling in the marcher - equivalence tests, then A/B (lego, bricks-like via --scene n/a, fox, real fox through the test suite)
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "bounds or march or golden or noisy" > gpurun_out/r3i_tests.log 2>&1; echo "pytest(march) rc=$?"; tail -4 gpurun_out/r3i_tests.log
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_trajectory_gpu.py -m gpu -q -x > gpurun_out/r3i_tests2.log 2>&1; echo "pytest(train) rc=$?"; tail -4 gpurun_out/r3i_tests2.log
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox"
run() { name=$1; shift; timeout 300 env "$@" python bench.py $Q $EXTRA > gpurun_out/r3i_$name.json 2> gpurun_out/r3i_$name.err; echo "$name rc=$?"; }
EXTRA="" run cull X=1
EXTRA="" run nocull NGP_MARCH_NO_BOUNDS=1
EXTRA="" run cull2 X=1
EXTRA="" run nocoarse NGP_MARCH_NO_COARSE=1
EXTRA="--config fox" run fox_cull X=1
EXTRA="" run nocoarse2 NGP_MARCH_NO_COARSE=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3i_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
        pk = d["extra"].get("probe_kernels", {})
        print(f.split("r3i_")[1][:-5].ljust(10), d["value"], d["ms_per_step"], {k: v.get("avg_launch_ms") for k, v in pk.items() if "march_wave" in k or "march_count" in k or "hash_fwd" in k or "records_runs" in k}, d["extra"]["param_signature"][:2])
    except Exception as e:
        print(f, "failed", e)
PY
