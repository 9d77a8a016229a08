#!/bin/bash
set -u
mkdir -p gpurun_out
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox"
run() { name=$1; shift; timeout 300 env "$@" python bench.py $Q > gpurun_out/r3m_$name.json 2> gpurun_out/r3m_$name.err; echo "$name rc=$?"; }
run old NGP_PIPELINE_ALLOC_RAYS=1
run new X=1
run old2 NGP_PIPELINE_ALLOC_RAYS=1
run new2 X=1
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox --config fox"
run fox_old NGP_PIPELINE_ALLOC_RAYS=1
run fox_new X=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3m_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
        print(f.split("r3m_")[1][:-5].ljust(8), d["value"], d["ms_per_step"], d["extra"]["param_signature"][:1])
    except Exception as e:
        print(f, "ERR", e); print(open(f.replace(".json", ".err")).read()[-1500:])
PY
timeout 600 python -m pytest tests/test_train_gpu.py -x -q 2>&1 | tail -3
