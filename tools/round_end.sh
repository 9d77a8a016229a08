#!/bin/bash
# End-of-round runs (round 4 as committed: the ${TAG}_ prefixes) on the GPU box (through gpurun): full GPU suite, bench lines, rocprofv3 passes, render / NeuS traces, the initialisation A/B.  Everything -> gpurun_out/.
set -u
R=$PWD
mkdir -p gpurun_out
TAG=${TAG:-r06z}; export TAG
WHAT=${1:-all}     # all | tests | bench | profiles | parts | curve | counters (the last one never as part of `all`)
if [ $WHAT = all ] || [ $WHAT = tests ]; then
bash tools/gpu.sh tests > /dev/null 2>&1; tail -6 gpurun_out/tests_gpu.log; cp gpurun_out/tests_gpu.log gpurun_out/${TAG}_tests_gpu.log
fi
if [ $WHAT = all ] || [ $WHAT = bench ]; then
bash tools/gpu.sh bench ${TAG}_bench_lego
bash tools/gpu.sh bench ${TAG}_bench_fox --config fox --no-fox --no-neus
bash tools/gpu.sh bench ${TAG}_bench_driver_style --gpus 1 --steps 20 --warmup 5
fi
if [ $WHAT = all ] || [ $WHAT = profiles ]; then
bash tools/collect_profiles.sh trace > gpurun_out/${TAG}_collect.log 2>&1; tail -12 gpurun_out/${TAG}_collect.log
fi
# counter passes ONLY on request and in a call of their own: ~7 minutes EACH on this stack (every dispatch is serialised under --pmc); `all` does not include them
if [ $WHAT = counters ]; then
bash tools/collect_profiles.sh pmc > gpurun_out/${TAG}_collect_pmc.log 2>&1; tail -12 gpurun_out/${TAG}_collect_pmc.log
bash tools/collect_profiles.sh mfma > gpurun_out/${TAG}_collect_mfma.log 2>&1; tail -12 gpurun_out/${TAG}_collect_mfma.log
fi
if [ $WHAT = all ] || [ $WHAT = parts ]; then
cd /tmp && export TMPDIR=/tmp
for part in ${PARTS:-render neus}; do
  rm -rf /tmp/pp_$part && mkdir -p /tmp/pp_$part
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pp_$part -o kt -- python $R/tools/profile_part.py $part > /tmp/pp_$part/log 2>&1
  grep "^render:\|^neus:" /tmp/pp_$part/log | tee $R/gpurun_out/${TAG}_${part}_wall.txt
  KT=$(find /tmp/pp_$part -name "*.db" | head -1)
  (cd $R && python tools/rocprof_summary.py "$KT" gpurun_out/${TAG}_${part}_kernel_trace.md "python tools/profile_part.py $part, rocprofv3 --kernel-trace --stats - dispatches after the marker kernel only" 0 triu_tril_kernel)
done
cd $R
fi
if [ $WHAT = all ] || [ $WHAT = curve ]; then
INVARIANT_UNIFORM_GAIN=3.0 timeout 600 python tools/train_curve.py gpurun_out/${TAG}_train_curve_bricks_gain3.md 40000 bricks > gpurun_out/${TAG}_curve.log 2>&1
INVARIANT_UNIFORM_GAIN=1.0 timeout 600 python tools/train_curve.py gpurun_out/${TAG}_train_curve_bricks_gain1.md 40000 bricks >> gpurun_out/${TAG}_curve.log 2>&1
tail -4 gpurun_out/${TAG}_train_curve_bricks_gain3.md gpurun_out/${TAG}_train_curve_bricks_gain1.md
fi
