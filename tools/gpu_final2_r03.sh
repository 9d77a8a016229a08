#!/bin/bash
# round 3, second end-of-round run (after the split backward became the default): full GPU suite, smoke, rocprofv3 passes of the final tree, the bench lines for profiles/
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r03_tests_gpu.log 2>&1; echo "pytest rc=$?"; tail -9 gpurun_out/r03_tests_gpu.log | cut -c1-300
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/collect_profiles_r03.sh all > gpurun_out/r03_collect.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/r03_collect.log
cp gpurun_out/r03_pmc.json profiles/r03_pmc.json 2>/dev/null
timeout 900 python bench.py > gpurun_out/r03_bench_lego.json 2> gpurun_out/r03_bench_lego.err; echo "bench lego rc=$?"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-fox --no-neus --no-cpu-baseline > gpurun_out/r03_bench_driver_style.json 2>/dev/null; echo "driver-style rc=$?"
BENCH_EXTRA_CFG='{"scene": "bricks"}' timeout 600 python bench.py --no-fox --no-neus --no-cpu-baseline > gpurun_out/r03_bench_bricks.json 2> gpurun_out/r03_bench_bricks.err; echo "bench bricks rc=$?"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --force-dist --no-fox --no-neus --no-cpu-baseline --no-psnr > gpurun_out/r03_bench_dist_world1.json 2>/dev/null; echo "dist rc=$?"
bash tools/gpu_timeline.sh > gpurun_out/r03_lego_timeline_raw.txt 2>&1; cp gpurun_out/timeline_step.txt gpurun_out/r03_lego_timeline_step.txt
python - <<'PY'
import json
for f in ("r03_bench_lego", "r03_bench_driver_style", "r03_bench_bricks", "r03_bench_dist_world1"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith('{"metric')][-1])
        r = d["roofline"] or {}
        print(f, d["value"], d["ms_per_step"], d["dtype"], r.get("kernel"), r.get("achieved"), r.get("frac"), r.get("executed_frac"), r.get("issued_frac"), r.get("pipe_util"), r.get("traffic"), r.get("fp16_pipe"),
              "stage", {k: v for k, v in (r.get("stage") or {}).items() if k in ("ms", "frac")}, "cpu", (d.get("cpu_baseline") or {}).get("value"), "fox", {k: v for k, v in (d["extra"].get("fox") or {}).items() if k in ("iters_per_s", "psnr_test_split_after_3000_steps")},
              "neus", {k: v for k, v in (d["extra"].get("neus") or {}).items() if k != "config"}, {k: v for k, v in d["extra"].items() if k.startswith(("psnr", "render"))})
    except Exception as e:
        print(f, "failed", e)
PY
