#!/bin/bash
# A/B of environment switches on the bench: tools/gpu_ab.sh <lego|fox> "VAR=1 VAR2=x" "VAR=2" ...   (first run is always the plain one)
set -u
cfg=$1; shift
mkdir -p gpurun_out
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)" 2>/dev/null
for variant in "A=1" "$@"; do
env $variant timeout 600 python bench.py --no-fox --no-cpu-baseline --no-psnr --steps 200 --config $cfg > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/ab.json') if l.startswith('{"metric')][-1])
    k=d['roofline']['ms_per_step_by_kernel']
    print('$cfg', '[$variant]', d['value'], d['ms_per_step'], {x:k[x] for x in list(k)[:6]})
except Exception as e:
    print('$cfg [$variant] failed', e, open('gpurun_out/ab.err').read()[-400:])
PY
done
