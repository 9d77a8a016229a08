#!/bin/bash
set -u
R=$PWD
CFG=${1:-fox}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for pm in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_WAIT_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pm | tr ' ' '_' | cut -c1-24)
  rm -rf /tmp/pm && mkdir -p /tmp/pm
  timeout 300 rocprofv3 --pmc $pm --kernel-trace -d /tmp/pm -o pm -- python $R/tools/bench_march.py --config $CFG --reps 4 > /tmp/pm.log 2>&1
  DB=$(find /tmp/pm -name "*.db" | head -1)
  python $R/tools/rocprof_pmc.py "$DB" $R/gpurun_out/c5_${CFG}_pmc_$tag.md "bench_march $CFG: $pm" 4 | tail -1
  grep "march_coop" $R/gpurun_out/c5_${CFG}_pmc_$tag.md
done
