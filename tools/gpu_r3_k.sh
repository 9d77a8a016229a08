#!/bin/bash
set -u
mkdir -p gpurun_out
Q="--steps 200 --warmup 32 --no-cpu-baseline --no-psnr --no-fox"
run() { name=$1; shift; timeout 300 env "$@" python bench.py $Q > gpurun_out/r3k_$name.json 2> gpurun_out/r3k_$name.err; echo "$name rc=$?"; }
run base X=1
run map3 NGP_HASH_FWD_BALANCE=3
run base2 X=1
run map3b NGP_HASH_FWD_BALANCE=3
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3k_*.json")):
    d = json.loads([l for l in open(f) if l.startswith('{"metric')][-1])
    pk = d["extra"].get("probe_kernels", {})
    print(f.split("r3k_")[1][:-5].ljust(8), d["value"], d["ms_per_step"], {k: v.get("avg_launch_ms") for k, v in pk.items() if "hash_fwd" in k}, d["extra"]["param_signature"][:1])
PY
