#!/bin/bash
# Runs on the GPU box (through gpurun): the rocprofv3 passes of this round's bench commands, post-processed into gpurun_out/ (copy the results into profiles/).
#   trace    --kernel-trace --stats of `bench.py` (lego / headline) and `bench.py --config fox` (procedural) and the real-fox leg
#   pmc      --pmc FETCH_SIZE / --pmc WRITE_SIZE (own passes, kernel trace only) of the lego and fox commands -> per-kernel HBM bytes per launch -> <tag>_pmc.json
#   mfma     --pmc MFMA counters of the lego (split fp16 MFMA) and fox (fp16 MFMA) commands
# Counter passes collect ONLY the library's kernels (--kernel-include-regex k_): the procedural scene is rendered by ~10^5 torch dispatches that would otherwise each be
# serialised under the counters (round 4 lost 35 GPU-minutes to that), and run a short schedule: 64 burn-in steps reach the adapted ray count, 16 + 16 steps are measured.
# usage: tools/collect_profiles.sh [what] [configs]      what = all | trace | pmc | mfma      configs = "lego fox" (default)
set -u
WHAT=${1:-all}
CFGS=${2:-"lego fox"}
TAG=${TAG:-r06z}
R=$PWD
mkdir -p $R/gpurun_out
BASE="python $R/bench.py --no-cpu-baseline --no-psnr --no-fox --no-neus --no-spheres --no-lego-gate"
PMC="--burn-in 64 --steps 16 --warmup 16"
INC_RE="k_(hash|bin|field|adam|march|mscan|composite|grid|occ|generate|reduce|mlp32|pack|level|bitfield|refresh)"
scene_of() { [ $1 = lego ] && echo bricks || echo spheres; }
cd /tmp && export TMPDIR=/tmp
if [ $WHAT = all ] || [ $WHAT = trace ]; then
for cfg in $CFGS; do
  rm -rf /tmp/pf_$cfg && mkdir -p /tmp/pf_$cfg
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_$cfg -o kt -- $BASE --config $cfg > /tmp/pf_$cfg/log 2>&1
  grep "^{\"metric" /tmp/pf_$cfg/log | tail -1 > $R/gpurun_out/${TAG}_${cfg}_bench_under_rocprof.json
  KT=$(find /tmp/pf_$cfg -name "*.db" | head -1)
  (cd $R && python tools/rocprof_summary.py "$KT" gpurun_out/${TAG}_${cfg}_kernel_trace.md "bench.py --config $cfg (N=1, 1024 burn-in + 64 warm-up + 200 timed steps), rocprofv3 --kernel-trace --stats" 200 && python tools/rocprof_gaps.py "$KT" 128 > gpurun_out/${TAG}_${cfg}_timeline.txt)
done
fi
# (r6) the real-fox leg (bench.py's extra.fox: ngp_fox.py on data/fox) under the kernel trace
if [ ${NO_REALFOX:-0} = 0 ] && { [ $WHAT = all ] || [ $WHAT = trace ] || [ $WHAT = realfox ]; }; then
  rm -rf /tmp/pf_realfox && mkdir -p /tmp/pf_realfox
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_realfox -o kt -- python $R/tools/fox_leg.py 200 > /tmp/pf_realfox/log 2>&1
  grep "^fox_leg" /tmp/pf_realfox/log | tail -1 > $R/gpurun_out/${TAG}_realfox_wall.txt
  KT=$(find /tmp/pf_realfox -name "*.db" | head -1)
  (cd $R && python tools/rocprof_summary.py "$KT" gpurun_out/${TAG}_realfox_kernel_trace.md "python tools/fox_leg.py 200 (ngp_fox.py on data/fox: 1024 burn-in + 200 timed steps), rocprofv3 --kernel-trace --stats" 200 && python tools/rocprof_gaps.py "$KT" 128 > gpurun_out/${TAG}_realfox_timeline.txt)
fi
if [ $WHAT = all ] || [ $WHAT = pmc ]; then
for cfg in $CFGS; do
for pm in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_${cfg}_$pm && mkdir -p /tmp/pm_${cfg}_$pm
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $pm --kernel-trace --kernel-include-regex "$INC_RE" -d /tmp/pm_${cfg}_$pm -o pm -- $BASE --config $cfg $PMC > /tmp/pm_${cfg}_$pm/log 2>&1
  echo "pmc $cfg $pm rc=$? $(tail -c 300 /tmp/pm_${cfg}_$pm/log | tr '\n' ' ')"
  DB=$(find /tmp/pm_${cfg}_$pm -name "*.db" | head -1)
  (cd $R && python tools/rocprof_pmc.py "$DB" gpurun_out/${TAG}_${cfg}_pmc_$(echo $pm | tr A-Z a-z).md "bench.py --config $cfg $PMC, rocprofv3 --pmc $pm --kernel-trace --kernel-include-regex <library kernels>" 16)
done
(cd $R && python tools/rocprof_pmc_json.py "$(find /tmp/pm_${cfg}_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pm_${cfg}_WRITE_SIZE -name '*.db' | head -1)" gpurun_out/${TAG}_pmc.json "python bench.py --no-cpu-baseline --no-psnr --no-fox --no-neus --no-spheres --config $cfg $PMC" $cfg $(scene_of $cfg))
done
fi
if [ $WHAT = all ] || [ $WHAT = mfma ]; then
for cfg in $CFGS; do
  rm -rf /tmp/pq_$cfg && mkdir -p /tmp/pq_$cfg
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace --kernel-include-regex k_field -d /tmp/pq_$cfg -o pq -- $BASE --config $cfg $PMC > /tmp/pq_$cfg/log 2>&1
  echo "mfma $cfg rc=$? $(tail -c 300 /tmp/pq_$cfg/log | tr '\n' ' ')"
  DB=$(find /tmp/pq_$cfg -name "*.db" | head -1)
  (cd $R && python tools/rocprof_pmc.py "$DB" gpurun_out/${TAG}_${cfg}_pmc_mfma.md "bench.py --config $cfg $PMC, rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace --kernel-include-regex k_field" 16
   python tools/rocprof_mfma_json.py "$DB" gpurun_out/${TAG}_pmc.json gpurun_out/${TAG}_mfma.md $cfg $([ $cfg = fox ] && echo 1 || echo 0))
done
fi
ls -la $R/gpurun_out | grep $TAG
