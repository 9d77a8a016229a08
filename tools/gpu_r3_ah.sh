#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -k "tiles_the_replicated" 2>&1 | tail -4 | cut -c1-300
for v in 3 2 3 2; do
  NGP_FIELD32_BWD=$v timeout 600 python bench.py --no-fox --no-cpu-baseline --no-neus > gpurun_out/r3ah_bench_$v.json 2> gpurun_out/r3ah_bench_$v.err; echo "bench bwd=$v rc=$?"
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r3ah_bench_$v.json") if l.startswith('{"metric')][-1])
    k = d["roofline"]["ms_per_step_by_kernel"]; r = d["roofline"]
    print("bwd=$v", d["value"], d["ms_per_step"], d["loss"], {n: k[n] for n in k if "field32" in n}, {a: b for a, b in d["extra"].items() if a.startswith("psnr")}, r["kernel"], r["frac"], r.get("executed_frac"), r.get("fp16_pipe"))
except Exception as e:
    print("bwd=$v failed", e); print(open("gpurun_out/r3ah_bench_$v.err").read()[-1500:])
PY
done
