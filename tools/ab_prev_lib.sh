#!/bin/bash
# A/B of the library in the tree against the sources in _prev_csrc/ (a copy of an earlier commit's csrc + header, made with `git show` before the call: the GPU box has no .git):
# builds both, then runs bench.py alternately with each library in place.   usage: tools/ab_prev_lib.sh <lego|fox> [repeats]
set -u
cfg=${1:-lego}; reps=${2:-2}
R=$(cd "$(dirname "$0")/.." && pwd)
cp $R/jnerf_amd/csrc/libngp_hip.so /tmp/lib_new.so
(cd $R/_prev_csrc/jnerf_amd/csrc && chmod +x build.sh && bash build.sh > /tmp/prev_build.log 2>&1 && cp libngp_hip.so /tmp/lib_prev.so) || { echo "building _prev_csrc failed"; tail -5 /tmp/prev_build.log; exit 2; }
mkdir -p $R/gpurun_out
for i in $(seq 1 $reps); do
  for v in prev new; do
    cp /tmp/lib_$v.so $R/jnerf_amd/csrc/libngp_hip.so
    (cd $R && timeout 600 python bench.py --no-fox --no-neus --no-cpu-baseline --no-psnr --no-spheres --no-lego-gate --steps 200 --config $cfg > gpurun_out/ab.json 2> gpurun_out/ab.err)
    python - <<PY
import json
try:
    d = json.loads([l for l in open('$R/gpurun_out/ab.json') if l.startswith('{"metric')][-1])
    k = d['roofline']['ms_per_step_by_kernel']
    print('$cfg', '[$v]', d['value'], d['ms_per_step'], {x: k[x] for x in list(k)[:9]}, flush=True)
except Exception as e:
    print('$cfg [$v] failed', e, open('$R/gpurun_out/ab.err').read()[-400:], flush=True)
PY
  done
done
cp /tmp/lib_new.so $R/jnerf_amd/csrc/libngp_hip.so
