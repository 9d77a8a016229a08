#!/bin/bash
# is the two-rank (gloo, one GPU) lego run reproducible run to run, sharded and not?
set -u
mkdir -p gpurun_out
run() {
  port=$((29600 + RANDOM % 300))
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --backend gloo --steps 20 --warmup 8 --burn-in 32 --config lego --images 4 --res 64 --no-psnr --no-kernel-events "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$*', d['extra']['param_signature'][:3], d['loss'], d['extra']['replicas_identical'])"
}
{
run; run
run --dp-host-sharded; run --dp-host-sharded
run --dp-overlap; run --dp-host-sharded --dp-overlap; run --dp-host-sharded --dp-overlap
NGP_FIELD32_FWD=mfma32 run; NGP_FIELD32_FWD=mfma32 run --dp-host-sharded --dp-overlap
} 2>&1 | tee gpurun_out/r3v_repro.txt
