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
run1() {
  port=$((29600 + RANDOM % 300))
  MASTER_ADDR=127.0.0.1 MASTER_PORT=$port RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --gpus 1 --force-dist --backend gloo --steps 20 --warmup 8 --burn-in 32 --config lego --images 4 --res 64 --no-psnr --no-kernel-events --no-fox --no-neus --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$TAG', d['extra']['param_signature'][:2], d['loss'])"
}
{
TAG="2 ranks, no fused tail"; export NGP_NO_FUSED_MLP_TAIL=1; run2; run2; run2; unset NGP_NO_FUSED_MLP_TAIL
TAG="1 rank gloo two-phase"; run1; run1; run1
TAG="2 ranks, split bwd too"; export NGP_FIELD32_BWD=3; run2; run2; unset NGP_FIELD32_BWD
} 2>&1 | tee gpurun_out/r3y_repro.txt
