#!/bin/bash
# Runs on the GPU box (through gpurun): the rocprofv3 passes of the round-4 bench commands, post-processed into gpurun_out/ (copy the results into profiles/).
#   pass 1  --kernel-trace --stats of `bench.py` (lego / headline) and `bench.py --config fox`
#   pass 2,3  --pmc FETCH_SIZE / --pmc WRITE_SIZE (own passes, kernel trace only) of the lego command -> per-kernel HBM bytes per launch
#   pass 4  --pmc MFMA counters of the lego (fp32-MFMA) and fox (fp16-MFMA) commands
# usage: tools/collect_profiles_r02.sh [what]      what = all | trace | pmc | mfma
set -u
WHAT=${1:-all}
TAG=r04
R=$PWD
mkdir -p $R/gpurun_out
BASE="python $R/bench.py --no-cpu-baseline --no-psnr --no-fox --no-neus --no-spheres"
# counter passes serialise every dispatch (~25 ms each under SQ counters): a short run - 64 burn-in steps reach the adapted ray count, 16 + 16 steps are measured
PMC="--burn-in 64 --steps 16 --warmup 16"
cd /tmp && export TMPDIR=/tmp
if [ $WHAT = all ] || [ $WHAT = trace ]; then
for cfg in lego fox; do
  rm -rf /tmp/pf_$cfg && mkdir -p /tmp/pf_$cfg
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_$cfg -o kt -- $BASE --config $cfg > /tmp/pf_$cfg/log 2>&1
  grep "^{\"metric" /tmp/pf_$cfg/log | tail -1 > $R/gpurun_out/${TAG}_${cfg}_bench_under_rocprof.json
  KT=$(find /tmp/pf_$cfg -name "*.db" | head -1)
  (cd $R && python tools/rocprof_summary.py "$KT" gpurun_out/${TAG}_${cfg}_kernel_trace.md "bench.py --config $cfg (N=1, 1024 burn-in + 64 warm-up + 200 timed steps), rocprofv3 --kernel-trace --stats" 200 && python tools/rocprof_gaps.py "$KT" 128 > gpurun_out/${TAG}_${cfg}_timeline.txt)
done
fi
if [ $WHAT = all ] || [ $WHAT = pmc ]; then
for pm in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$pm && mkdir -p /tmp/pm_$pm
  timeout 420 rocprofv3 --pmc $pm --kernel-trace -d /tmp/pm_$pm -o pm -- $BASE --config lego $PMC > /tmp/pm_$pm/log 2>&1
  DB=$(find /tmp/pm_$pm -name "*.db" | head -1)
  (cd $R && python tools/rocprof_pmc.py "$DB" gpurun_out/${TAG}_lego_pmc_$(echo $pm | tr A-Z a-z).md "bench.py --config lego $PMC, rocprofv3 --pmc $pm --kernel-trace" 16)
done
(cd $R && python tools/rocprof_pmc_json.py "$(find /tmp/pm_FETCH_SIZE -name '*.db' | head -1)" "$(find /tmp/pm_WRITE_SIZE -name '*.db' | head -1)" gpurun_out/${TAG}_pmc.json "python bench.py --no-cpu-baseline --no-psnr --no-fox --no-neus --no-spheres --config lego $PMC" lego)
fi
if [ $WHAT = all ] || [ $WHAT = mfma ]; then
for cfg in lego fox; do
  rm -rf /tmp/pq_$cfg && mkdir -p /tmp/pq_$cfg
  timeout 420 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace -d /tmp/pq_$cfg -o pq -- $BASE --config $cfg $PMC > /tmp/pq_$cfg/log 2>&1
  DB=$(find /tmp/pq_$cfg -name "*.db" | head -1)
  (cd $R && python tools/rocprof_pmc.py "$DB" gpurun_out/${TAG}_${cfg}_pmc_mfma.md "bench.py --config $cfg --steps 32 --warmup 16, rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --kernel-trace" 16
   python tools/rocprof_mfma_json.py "$DB" gpurun_out/${TAG}_pmc.json gpurun_out/${TAG}_mfma.md $cfg $([ $cfg = fox ] && echo 1 || echo 0))
done
fi
ls -la $R/gpurun_out | grep $TAG
