#!/bin/bash
set -u
mkdir -p gpurun_out
run2() {
  port=$((29600 + RANDOM % 300))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --backend gloo --steps 20 --warmup 8 --burn-in 32 --config lego --images 4 --res 64 --no-psnr --no-kernel-events "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$TAG', d['extra']['param_signature'][:2], d['loss'])"
}
{
TAG="train split, density mfma32"; export NGP_FIELD32_FWD=split NGP_DENSITY32_FWD=mfma32; run2; run2; run2
TAG="train mfma32, density split"; export NGP_FIELD32_FWD=mfma32 NGP_DENSITY32_FWD=split; run2; run2; run2
} 2>&1 | tee gpurun_out/r3aa_repro.txt
