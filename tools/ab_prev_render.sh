#!/bin/bash
# Render throughput (bench.py's extra.render_Msamples_per_s: four 800 x 800 views incl. ray generation and the device->host copy) of the tree's library against the library
# of an earlier commit whose ABI is the same (sources copied to $PREV_DIR with `git archive` before the call), alternately in ONE gpurun call.   usage: PREV_DIR=_prev_r5 tools/ab_prev_render.sh [reps]
set -u
reps=${1:-2}
R=$(cd "$(dirname "$0")/.." && pwd)
PREV_DIR=${PREV_DIR:-_prev_csrc}
cp $R/jnerf_amd/csrc/libngp_hip.so /tmp/lib_new.so
(cd $R/$PREV_DIR/jnerf_amd/csrc && chmod +x build.sh && bash build.sh > /tmp/prev_build.log 2>&1 && cp libngp_hip.so /tmp/lib_prev.so) || { echo "building $PREV_DIR failed"; tail -5 /tmp/prev_build.log; exit 2; }
for i in $(seq 1 $reps); do
  for v in prev new; do
    cp /tmp/lib_$v.so $R/jnerf_amd/csrc/libngp_hip.so
    (cd $R && timeout 600 python bench.py --no-fox --no-neus --no-cpu-baseline --no-spheres --no-lego-gate --steps 100 > gpurun_out/ab.json 2> gpurun_out/ab.err)
    python - <<PY
import json
try:
    d = json.loads([l for l in open('$R/gpurun_out/ab.json') if l.startswith('{"metric')][-1])
    e = d['extra']
    print('render [$v]', d['value'], 'it/s;', e.get('render_Msamples_per_s'), 'Msamples/s', {k: v for k, v in e.items() if k.startswith('render_ms') or k.startswith('psnr')}, flush=True)
except Exception as ex:
    print('render [$v] failed', ex, open('$R/gpurun_out/ab.err').read()[-400:], flush=True)
PY
  done
done
cp /tmp/lib_new.so $R/jnerf_amd/csrc/libngp_hip.so
