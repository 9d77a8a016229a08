#!/bin/bash
# quick check on the GPU box: a pytest selection, then both bench configurations (short), one summary line each
set -u
SEL=${1:-hash}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -k "$SEL" 2>&1 | tail -2
for cfg in lego fox; do
timeout 600 python bench.py --no-fox --no-cpu-baseline --no-psnr --steps 100 --config $cfg > gpurun_out/quick_bench_$cfg.json 2> gpurun_out/quick_bench_$cfg.err; echo "bench $cfg rc=$?"
python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/quick_bench_$cfg.json') if l.startswith('{"metric')][-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], {k:v for k,v in d['roofline']['ms_per_step_by_kernel'].items() if v>0.015})
PY
done
