#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for cfg in lego fox; do
timeout 600 python bench.py --no-fox --no-cpu-baseline --steps 100 --config $cfg > gpurun_out/c6_bench_$cfg.json 2> gpurun_out/c6_bench_$cfg.err; echo "bench $cfg rc=$?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/c6_bench_$cfg.json') if l.startswith('{"metric')][-1])
print(d['value'], d['ms_per_step'], {k:v for k,v in d['roofline']['ms_per_step_by_kernel'].items() if v>0.015})
print({k:v for k,v in d['extra'].items() if k!='probe_kernels'})
PY
done
