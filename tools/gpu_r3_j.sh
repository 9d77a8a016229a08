#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/probe_host_block.py 2>&1 | tail -5
cp gpurun_out/r03_pmc.json profiles/r03_pmc.json 2>/dev/null
timeout 900 python bench.py > gpurun_out/r03_bench_lego.json 2> gpurun_out/r03_bench_lego.err; echo "bench lego rc=$?"
timeout 900 python bench.py --config fox --no-fox > gpurun_out/r03_bench_fox.json 2> gpurun_out/r03_bench_fox.err; echo "bench fox rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-fox --no-cpu-baseline > gpurun_out/r03_bench_driver_style.json 2>/dev/null; echo "driver-style rc=$?"
BENCH_EXTRA_CFG='{"scene": "bricks"}' timeout 600 python bench.py --no-fox --no-cpu-baseline > gpurun_out/r03_bench_bricks.json 2> gpurun_out/r03_bench_bricks.err; echo "bench bricks rc=$?"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --force-dist --no-fox --no-cpu-baseline --no-psnr > gpurun_out/r03_bench_dist_world1.json 2>/dev/null; echo "dist rc=$?"
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "field_bwd_vs_oracle" 2>&1 | tail -2
python - <<'PY'
import json
for f in ("r03_bench_lego", "r03_bench_fox", "r03_bench_driver_style", "r03_bench_bricks", "r03_bench_dist_world1"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith('{"metric')][-1])
        r = d["roofline"] or {}
        print(f, d["value"], d["ms_per_step"], d["dtype"], r.get("kernel"), r.get("bound"), r.get("achieved"), r.get("frac"), r.get("executed_frac"), r.get("issued_frac"), r.get("pipe_util"), r.get("traffic"), r.get("avg_launch_ms"))
    except Exception as e:
        print(f, "failed", e)
PY
