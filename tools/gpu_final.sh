#!/bin/bash
# round-end check on the GPU box: full GPU test suite, smoke(), the two bench lines that go to profiles/
set -u
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/tests_gpu.log 2>&1; echo "pytest rc=$?"; tail -9 gpurun_out/tests_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r02_bench_lego.json 2> gpurun_out/r02_bench_lego.err; echo "bench lego rc=$?"
timeout 900 python bench.py --config fox --no-fox > gpurun_out/r02_bench_fox.json 2> gpurun_out/r02_bench_fox.err; echo "bench fox rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-fox --no-cpu-baseline > gpurun_out/r02_bench_driver_style.json 2>/dev/null; echo "driver-style rc=$?"
python - <<'PY'
import json
for f in ("r02_bench_lego", "r02_bench_fox", "r02_bench_driver_style"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith('{"metric')][-1])
        r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], d["dtype"], r["kernel"], r["bound"], r["achieved"], r["frac"], r.get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "fox", (d["extra"].get("fox") or {}))
    except Exception as e:
        print(f, "failed", e)
PY
