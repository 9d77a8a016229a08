#!/bin/bash
# where does the run-to-run difference of two-rank runs on ONE GPU come from?  A matrix of tiny lego runs, each printing its parameter signature:
#   two2      two ranks (two processes, one GPU, gloo)                                       - the known irreproducible case
#   two2_nopipe   the same, pipeline_sampling = false (no side streams, no buffer-set reuse)
#   two2_serial   the same, every kernel and copy serialised by the runtime (AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3): no intra-process concurrency at all
#   one_gloo  ONE rank through the same gloo two-phase step, alone on the GPU
#   one_gloo_hog  the same while an UNRELATED process (a plain single-GPU bench) shares the GPU
#   plain_hog a plain single-process run (no process group at all) while an unrelated process shares the GPU
#   plain     the same alone (control)
# If the runs beside a stranger process differ from each other while the same runs alone agree, the cause is two processes time-sharing one GPU - not the host
# sequencing of this package - and cannot occur with one process per GPU.      usage: tools/probe_repro_matrix.sh [runs per case] [cases...]
set -u
N=${1:-3}; shift || true
CASES=${*:-"plain plain_hog one_gloo one_gloo_hog two2 two2_nopipe two2_serial"}
R=$(cd "$(dirname "$0")/.." && pwd)
TINY="--steps 20 --warmup 8 --burn-in 32 --config lego --images 4 --res 64 --no-psnr --no-kernel-events --no-fox --no-neus --no-spheres --no-cpu-baseline"
sig() { grep '^{"metric' | tail -1 | python -c "
import json, sys
t = sys.stdin.read()
if not t.strip():
    print('$1: FAILED (no line)'); sys.exit(0)
d = json.loads(t)
print('$1: loss', d['loss'], 'replicas_identical', d['extra'].get('replicas_identical'), 'signature', [round(x, 9) for x in d['extra']['param_signature'][:6]])"; }
hog_start() { python $R/bench.py --steps 60000 --warmup 8 --burn-in 32 --config lego --images 4 --res 64 --no-psnr --no-kernel-events --no-fox --no-neus --no-spheres --no-cpu-baseline > /tmp/hog.out 2> /tmp/hog.err & HOG=$!; sleep ${HOG_WARM:-12}; }
hog_stop() { kill $HOG 2>/dev/null; wait $HOG 2>/dev/null; }
PORT=29700
for c in $CASES; do
  for k in $(seq 1 $N); do
    PORT=$((PORT + 1))
    case $c in
    plain)        timeout 200 python $R/bench.py $TINY 2> /tmp/rm.err | sig "$c run $k" ;;
    plain_hog)    hog_start; timeout 200 python $R/bench.py $TINY 2> /tmp/rm.err | sig "$c run $k"; hog_stop ;;
    one_gloo)     timeout 200 python $R/bench.py --gpus 1 --force-dist --backend gloo $TINY 2> /tmp/rm.err | sig "$c run $k" ;;
    one_gloo_hog) hog_start; timeout 200 python $R/bench.py --gpus 1 --force-dist --backend gloo $TINY 2> /tmp/rm.err | sig "$c run $k"; hog_stop ;;
    two2)         timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT $R/bench.py --gpus 2 --backend gloo $TINY 2> /tmp/rm.err | sig "$c run $k" ;;
    two2_nopipe)  BENCH_EXTRA_CFG='{"pipeline_sampling": false}' timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT $R/bench.py --gpus 2 --backend gloo $TINY 2> /tmp/rm.err | sig "$c run $k" ;;
    two2_serial)  AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 BENCH_EXTRA_CFG='{"pipeline_sampling": false}' timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT $R/bench.py --gpus 2 --backend gloo $TINY 2> /tmp/rm.err | sig "$c run $k" ;;
    esac
  done
done
