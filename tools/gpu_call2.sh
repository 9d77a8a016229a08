#!/bin/bash
# GPU call 2: full parity suite (no -x), then the lego bench on the native fp32 path
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_train_gpu.py > gpurun_out/c2_pytest_parity.log 2>&1; echo "parity rc=$?"
tail -15 gpurun_out/c2_pytest_parity.log
timeout 1500 python -m pytest tests/test_train_gpu.py -m gpu -q > gpurun_out/c2_pytest_train.log 2>&1; echo "train rc=$?"
tail -25 gpurun_out/c2_pytest_train.log
timeout 600 python bench.py --no-fox > gpurun_out/c2_bench_lego.json 2> gpurun_out/c2_bench_lego.err; echo "bench rc=$?"
tail -c 1500 gpurun_out/c2_bench_lego.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/c2_bench_lego.json') if l.startswith('{"metric')][-1])
    print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['ms_per_step_by_kernel'])
    print(d['extra'])
except Exception as e: print('no bench line', e)
PY
