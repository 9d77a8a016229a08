set -u
mkdir -p gpurun_out
bash tools/rocprof_pmc_sq.sh "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" gpurun_out/r4j_sq.md > gpurun_out/r4j_sq.log 2>&1
cat gpurun_out/r4j_sq.md | cut -c1-220
