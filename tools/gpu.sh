#!/bin/bash
# The ONE script for GPU-box runs (through gpurun; replaces round 3's 39 one-off tools/gpu_*.sh).  Everything it writes goes to gpurun_out/.
#   tools/gpu.sh tests [pytest args]                 pytest -m gpu with durations                      -> gpurun_out/tests_gpu.log
#   tools/gpu.sh ab <lego|fox> "VAR=1 VAR2=x" ...    bench.py once plain, then once per environment set -> one line each (it/s, ms/step, top kernels)
#   tools/gpu.sh bench <tag> [bench args]            python bench.py ...                               -> gpurun_out/<tag>.json (+ .err)
#   tools/gpu.sh py <tag> <script> [args]            python <script> ...                               -> gpurun_out/<tag>.txt
set -u
mkdir -p gpurun_out
cmd=${1:-tests}; shift || true
case $cmd in
tests)
  timeout 3000 python -m pytest tests -m gpu -q -x --durations=15 "$@" 2>&1 | tail -60 | tee gpurun_out/tests_gpu.log
  ;;
ab)
  cfg=$1; shift
  for variant in "A=1" "$@"; do
    env $variant timeout 600 python bench.py --no-fox --no-neus --no-cpu-baseline --no-psnr --steps 200 --config $cfg > gpurun_out/ab.json 2> gpurun_out/ab.err
    python - <<PY
import json
try:
    d = json.loads([l for l in open('gpurun_out/ab.json') if l.startswith('{"metric')][-1])
    k = d['roofline']['ms_per_step_by_kernel']
    print('$cfg', '[$variant]', d['value'], d['ms_per_step'], {x: k[x] for x in list(k)[:8]}, flush=True)
except Exception as e:
    print('$cfg [$variant] failed', e, open('gpurun_out/ab.err').read()[-600:], flush=True)
PY
  done 2>&1 | tee -a gpurun_out/ab_lines.txt
  ;;
bench)
  tag=$1; shift
  timeout 1500 python bench.py "$@" > gpurun_out/$tag.json 2> gpurun_out/$tag.err
  grep '^{"metric' gpurun_out/$tag.json | tail -1 | cut -c1-600
  ;;
py)
  tag=$1; shift
  timeout 1500 python "$@" 2>&1 | tee gpurun_out/$tag.txt | tail -80
  ;;
*)
  echo "unknown command $cmd"; exit 2
  ;;
esac
