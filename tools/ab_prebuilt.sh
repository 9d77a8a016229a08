#!/bin/bash
# The tree's library against a PREBUILT earlier library (_prev_csrc/jnerf_amd/csrc/libngp_hip.so: `git archive <commit> jnerf_amd/csrc include | tar -x -C _prev_csrc`, built in
# the container - the GPU box spends no minutes compiling): output bits of the field kernels under both (tools/probe_lib_bits.py), then bench.py / tools/fox_leg.py alternately.
# usage: [PREV_LIB=<path of the other library>] tools/ab_prebuilt.sh <lego reps> <fox reps>     (writes the faster lego variant's name, prev | new, to /tmp/ab_winner)
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
lreps=${1:-2}; freps=${2:-1}
cp $R/jnerf_amd/csrc/libngp_hip.so /tmp/lib_new.so
cp ${PREV_LIB:-$R/_prev_csrc/jnerf_amd/csrc/libngp_hip.so} /tmp/lib_prev.so || exit 2
rm -f /tmp/ab_lego_*.txt
mkdir -p $R/gpurun_out
for v in prev new; do
  cp /tmp/lib_$v.so $R/jnerf_amd/csrc/libngp_hip.so
  (cd $R && timeout 300 python tools/probe_lib_bits.py 2>&1 | grep "^bits\|Error\|error" > /tmp/bits_$v.txt)
done
echo "== field kernels' output hashes, previous library | this tree (identical lines: $(comm -12 <(sort /tmp/bits_prev.txt) <(sort /tmp/bits_new.txt) | wc -l) of $(wc -l < /tmp/bits_new.txt))"
diff /tmp/bits_prev.txt /tmp/bits_new.txt && echo "all bits identical"
for i in $(seq 1 $lreps); do
  for v in prev new; do
    cp /tmp/lib_$v.so $R/jnerf_amd/csrc/libngp_hip.so
    (cd $R && timeout 600 python bench.py --no-fox --no-neus --no-cpu-baseline --no-psnr --no-spheres --no-lego-gate --steps 200 --config lego > gpurun_out/ab.json 2> gpurun_out/ab.err)
    python - <<PY
import json
try:
    d = json.loads([l for l in open('$R/gpurun_out/ab.json') if l.startswith('{"metric')][-1])
    k = d['roofline']['ms_per_step_by_kernel']
    print('lego', '[$v]', d['value'], d['ms_per_step'], {x: k[x] for x in list(k)[:9]}, 'render', d.get('extra', {}).get('render_Msamples_per_s'), flush=True)
    open('/tmp/ab_lego_$v.txt', 'a').write('%f\n' % d['value'])
except Exception as e:
    print('lego [$v] failed', e, open('$R/gpurun_out/ab.err').read()[-400:], flush=True)
PY
  done
done
for i in $(seq 1 $freps); do
  for v in prev new; do
    cp /tmp/lib_$v.so $R/jnerf_amd/csrc/libngp_hip.so
    echo "realfox [$v] $(cd $R && timeout 600 python tools/fox_leg.py 2>&1 | tail -1)"
  done
done
cp /tmp/lib_new.so $R/jnerf_amd/csrc/libngp_hip.so
python - <<PY
import os
m = lambda v: (lambda a: sum(a) / len(a) if a else 0.0)([float(x) for x in open('/tmp/ab_lego_%s.txt' % v)] if os.path.exists('/tmp/ab_lego_%s.txt' % v) else [])
w = 'prev' if m('prev') > m('new') else 'new'
open('/tmp/ab_winner', 'w').write(w)
print('lego mean it/s: prev %.1f, new %.1f -> %s' % (m('prev'), m('new'), w))
PY
