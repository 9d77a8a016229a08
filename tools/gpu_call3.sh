#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/c3_pytest.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/c3_pytest.log
for cfg in lego fox; do
timeout 600 python bench.py --no-fox --no-cpu-baseline --config $cfg > gpurun_out/c3_bench_$cfg.json 2> gpurun_out/c3_bench_$cfg.err; echo "bench $cfg rc=$?"
python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/c3_bench_$cfg.json') if l.startswith('{"metric')][-1])
    print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['ms_per_step_by_kernel'])
    print({k:v for k,v in d['extra'].items() if k!='probe_kernels'})
except Exception as e: print('no bench line', e)
PY
done
