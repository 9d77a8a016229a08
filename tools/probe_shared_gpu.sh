#!/bin/bash
# VERDICT r4 item 2, third step: the forward gather of THIS library returns different values for identical inputs while an unrelated training process shares the GPU
# (profiles/r05c_repro_plain_hip.txt: plain-HIP victim, one stream, no torch; a library-independent random gather does not).  Which property of the kernel, which kind of
# neighbour?  Builds diagnosis variants of the library (-DNGP_PROBE_WIDE_LOADS_F32: the 16-byte gathers of rounds 1-4 for the fp32 table - the product has used 8-byte loads there since this probe; -DNGP_PROBE_LINEAR_MAP: no XCD-aware block map, with the 16-byte gathers), computes
# every variant's reference result ALONE (stored in /tmp), then runs the victims beside (a) a training process of this package, (b) a torch process that never loads this
# library (matrix products), (c) a torch process issuing small kernels on four streams.  Prints where the results differ from the reference.      usage: tools/probe_shared_gpu.sh [seconds per victim]
set -u
SEC=${1:-4}
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -w"
OTHERS=$(ls $R/jnerf_amd/csrc/build/*.o | grep -v hash_encode.o)
for v in base wide128 linear; do
  mkdir -p /tmp/v_$v
  case $v in base) D="";; wide128) D="-DNGP_PROBE_WIDE_LOADS_F32";; linear) D="-DNGP_PROBE_LINEAR_MAP -DNGP_PROBE_WIDE_LOADS_F32";; esac
  ( cd $R/jnerf_amd/csrc && /opt/rocm/bin/hipcc $FLAGS $D -c hash_encode.hip -o /tmp/v_$v/hash_encode.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/v_$v/hash_encode.o -o /tmp/v_$v/libngp_hip.so ) &
done
wait
for v in base wide128 linear; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/repro_two_process.hip -I$R/include -L/tmp/v_$v -lngp_hip -Wl,-rpath,/tmp/v_$v -o /tmp/repro_$v || exit 2
  rm -f /tmp/ref_$v.bin; REPRO_REF=/tmp/ref_$v.bin /tmp/repro_$v 1 0 "$v alone"          # reference computed with the GPU to itself
done
victim() { REPRO_REF=/tmp/ref_$1.bin /tmp/repro_$1 $SEC 0 "$1 beside $2"; }
python $R/bench.py --steps 200000 --warmup 8 --burn-in 32 --config lego --images 4 --res 64 --no-psnr --no-kernel-events --no-fox --no-neus --no-spheres --no-cpu-baseline --no-lego-gate > /tmp/hog.out 2> /tmp/hog.err &
HOG=$!; sleep 12
for v in base wide128 linear; do victim $v "a training process of this package"; done
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
python - > /tmp/hogb.out 2>&1 <<'PY' &
import torch, time
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
t0 = time.time()
while time.time() - t0 < 40:
    for _ in range(50): c = a @ b
    torch.cuda.synchronize()
PY
HOG=$!; sleep 8
victim wide128 "a torch process doing matrix products (never loads this library)"
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
python - > /tmp/hogc.out 2>&1 <<'PY' &
import torch, time
st = [torch.cuda.Stream() for _ in range(4)]
xs = [torch.randn(1 << 16, device="cuda") for _ in range(4)]
t0 = time.time()
while time.time() - t0 < 40:
    for k in range(200):
        with torch.cuda.stream(st[k % 4]): xs[k % 4].mul_(1.0001).add_(1e-3)
    torch.cuda.synchronize()
PY
HOG=$!; sleep 8
victim wide128 "a torch process issuing small kernels on four streams"
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
