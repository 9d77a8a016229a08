#!/bin/bash
# are two-rank runs (two processes on ONE GPU, gloo) reproducible run to run?  Runs the two-rank lego bench N times and prints the parameter signatures.
# usage: tools/probe_two_rank_repro.sh [runs]
set -u
N=${1:-3}
R=$(cd "$(dirname "$0")/.." && pwd)
for k in $(seq 1 $N); do
  PORT=$((29600 + k))
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT $R/bench.py --gpus 2 --backend gloo --steps 20 --warmup 8 --burn-in 32 \
    --config lego --images 4 --res 64 --no-psnr --no-kernel-events 2> /tmp/two_rank_$k.err | grep '^{"metric' | python -c "
import json, sys
d = json.loads(sys.stdin.read())
print('run $k: loss', d['loss'], 'replicas_identical', d['extra']['replicas_identical'], 'signature', [round(x, 9) for x in d['extra']['param_signature'][:6]])"
done
