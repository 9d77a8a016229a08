#!/bin/bash
set -u
mkdir -p gpurun_out
run() {
  port=$((29600 + RANDOM % 300))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --backend gloo --steps 20 --warmup 8 --burn-in 32 --config lego --images 4 --res 64 --no-psnr --no-kernel-events "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$TAG', d['extra']['param_signature'][:2], d['loss'])"
}
{
TAG=mfma32; export NGP_FIELD32_FWD=mfma32; run; run; run; run
TAG=split; export NGP_FIELD32_FWD=split; run; run; run; run
TAG=split-nopipe; export BENCH_EXTRA_CFG='{"pipeline_sampling": false}'; run; run; run
} 2>&1 | tee gpurun_out/r3x_repro.txt
