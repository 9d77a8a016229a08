#!/bin/bash
# round 3: the split-operand fp32 BACKWARD (field_split.hip): parity tests, then A/B of the bench line (NGP_FIELD32_BWD=3 split vs 2 two-group fp32 MFMA) on one box
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_train_gpu.py -m gpu -q -k "field32 or fp32 or fused_network or fast_path or full_size or converges" --durations=5 > gpurun_out/r3t_tests.log 2>&1; echo "tests rc=$?"; tail -30 gpurun_out/r3t_tests.log | cut -c1-400
for v in 3 2 3 2; do
  NGP_FIELD32_BWD=$v timeout 600 python bench.py --no-fox --no-cpu-baseline --no-neus > gpurun_out/r3t_bench_$v.json 2> gpurun_out/r3t_bench_$v.err; echo "bench bwd=$v rc=$?"
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3t_bench_$v.json") if l.startswith('{"metric')][-1])
    k = d["roofline"]["ms_per_step_by_kernel"]
    print("bwd=$v", d["value"], d["ms_per_step"], d["loss"], {n: k[n] for n in k if "field32" in n}, {a: b for a, b in d["extra"].items() if a.startswith("psnr")}, d["roofline"]["kernel"])
except Exception as e:
    print("bwd=$v failed", e); print(open("gpurun_out/r3t_bench_$v.err").read()[-1500:])
PY
done
