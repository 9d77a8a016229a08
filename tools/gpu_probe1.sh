#!/bin/bash
set -u
mkdir -p gpurun_out
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py --no-fox --no-cpu-baseline --no-psnr --steps 100 --config lego > gpurun_out/p1_$name.json 2> gpurun_out/p1_$name.err; 
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/p1_$name.json') if l.startswith('{"metric')][-1])
print('$name', d['value'], d['ms_per_step'], d['roofline']['kernel'], {k:v for k,v in d['roofline']['ms_per_step_by_kernel'].items() if v>0.02})
PY
}
run base A=1
run noowner NGP_PROBE_LEVEL_MASK=0
run noside NGP_HASH_BWD_NO_SIDE_STREAM=1
run oldmarch NGP_MARCH_COUNT=g
