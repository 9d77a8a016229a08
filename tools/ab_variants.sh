#!/bin/bash
# A/B of compile-time variants of the library in ONE gpurun call: builds jnerf_amd/csrc once per variant (EXTRA="<flags>" of build.sh), then runs bench.py alternately with
# each library in place.   usage: tools/ab_variants.sh <lego|fox> <repeats> "name=flags" ...
# ("name=" = the tree as it is; "name=@VAR=1" = the tree's library with VAR=1 in the environment)
set -u
cfg=$1; reps=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out
cp $R/jnerf_amd/csrc/libngp_hip.so /tmp/lib_base.so
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  if [ "${flags:0:1}" = "@" ] || [ -z "$flags" ]; then continue; fi
  (cd $R/jnerf_amd/csrc && rm -rf build && EXTRA="$flags" bash build.sh > /tmp/build_$name.log 2>&1 && cp libngp_hip.so /tmp/lib_$name.so) || { echo "building $name failed"; tail -5 /tmp/build_$name.log; exit 2; }
done
(cd $R/jnerf_amd/csrc && rm -rf build)
for i in $(seq 1 $reps); do
  for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}
    envs="A=$i"; lib=/tmp/lib_base.so
    if [ "${flags:0:1}" = "@" ]; then envs="${flags:1}"; elif [ -n "$flags" ]; then lib=/tmp/lib_$name.so; fi
    cp $lib $R/jnerf_amd/csrc/libngp_hip.so
    (cd $R && env $envs timeout 600 python bench.py --no-fox --no-neus --no-cpu-baseline --no-psnr --no-spheres --no-lego-gate --steps 200 --config $cfg > gpurun_out/ab.json 2> gpurun_out/ab.err)
    python - <<PY
import json
try:
    d = json.loads([l for l in open('$R/gpurun_out/ab.json') if l.startswith('{"metric')][-1])
    k = d['roofline']['ms_per_step_by_kernel']
    print('$cfg', '[$name]', d['value'], d['ms_per_step'], {x: k[x] for x in list(k)[:9]}, flush=True)
except Exception as e:
    print('$cfg [$name] failed', e, open('$R/gpurun_out/ab.err').read()[-400:], flush=True)
PY
  done
done
cp /tmp/lib_base.so $R/jnerf_amd/csrc/libngp_hip.so
