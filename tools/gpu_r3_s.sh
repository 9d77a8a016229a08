#!/bin/bash
# round 3: the split-operand fp32 forward (field_split.hip): parity tests, then A/B of the bench line (split vs exact-product kernel) on one box
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_train_gpu.py -m gpu -q -k "field32 or fp32 or fused_network or fast_path or full_size" --durations=5 > gpurun_out/r3s_tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r3s_tests.log
for v in split mfma32 split mfma32; do
  NGP_FIELD32_FWD=$v timeout 600 python bench.py --no-fox --no-cpu-baseline --no-neus --no-psnr > gpurun_out/r3s_bench_$v.json 2> gpurun_out/r3s_bench_$v.err; echo "bench $v rc=$?"
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3s_bench_$v.json") if l.startswith('{"metric')][-1])
    k = d["roofline"]["ms_per_step_by_kernel"]
    print("$v", d["value"], d["ms_per_step"], {n: k[n] for n in k if "field32" in n or "hash_fwd" in n}, d["extra"].get("render_Msamples_per_s"))
except Exception as e:
    print("$v failed", e)
PY
done
