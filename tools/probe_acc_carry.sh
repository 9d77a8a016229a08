#!/bin/bash
# VERDICT r5 item 1(b), ONE bounded experiment: the hash backward still differs in 2 of 300 launches (12-25 table entries) while a second process trains on the same GPU, and
# narrow record loads did not change that (profiles/r05f_*).  Remaining suspect: the 64-bit LDS atomics (ds_add_u64) of k_bin_accumulate2 across a wave save / restore.  This
# builds the library with -DNGP_PROBE_ACC_U32_CARRY (two 32-bit LDS atomics with carry, same integer sums) and repeats the hash-backward stage `reps` times on frozen inputs,
# alone and beside a training process, with the product build and with the variant.  0 differing launches with the variant and > 0 with the product = the platform note;
# the same rate = unexplained, one process per GPU stays a requirement.      usage: tools/probe_acc_carry.sh [reps]
set -u
REPS=${1:-300}
R=$(cd "$(dirname "$0")/.." && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -w"
cd $R/jnerf_amd/csrc
cp libngp_hip.so /tmp/libngp_product.so
OTHERS=$(ls build/*.o | grep -v hash_encode.o)
/opt/rocm/bin/hipcc $FLAGS -DNGP_PROBE_ACC_U32_CARRY -c hash_encode.hip -o /tmp/hash_encode_carry.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/hash_encode_carry.o -o /tmp/libngp_carry.so || exit 2
cd $R
export PROBE_STAGES="hash backward"
probe() { cp /tmp/libngp_$1.so jnerf_amd/csrc/libngp_hip.so; python tools/probe_kernels_under_hog.py $REPS 2>&1 | grep -v "^(" | sed "s/^/[$1, $2] /"; }
probe product alone
probe carry alone
# the neighbour trains from a COPY of the tree: the probes below swap jnerf_amd/csrc/libngp_hip.so, and a process that has the old file mapped dies with it (r06a: the hog segfaulted at the first swap)
rm -rf /tmp/hogrepo && mkdir -p /tmp/hogrepo && cp -r bench.py jnerf_amd jnerf tests oracle projects tools include __graft_entry__.py /tmp/hogrepo/ 2>/dev/null
(cd /tmp/hogrepo && cp /tmp/libngp_product.so jnerf_amd/csrc/libngp_hip.so)
python /tmp/hogrepo/bench.py --steps 200000 --warmup 8 --burn-in 32 --config lego --images 4 --res 64 --no-psnr --no-kernel-events --no-fox --no-neus --no-spheres --no-cpu-baseline --no-lego-gate > /tmp/hog.out 2> /tmp/hog.err &
HOG=$!; sleep 20
probe product "beside a training process"
kill -0 $HOG 2>/dev/null && echo "(neighbour alive after the first swap)" || echo "(NEIGHBOUR DIED - the 'beside' lines are not valid)"
probe carry "beside a training process"
probe product "beside a training process, second pass"
probe carry "beside a training process, second pass"
kill -0 $HOG 2>/dev/null && echo "(neighbour alive at the end)" || echo "(NEIGHBOUR DIED - the 'beside' lines are not valid)"
kill $HOG 2>/dev/null; wait $HOG 2>/dev/null
cp /tmp/libngp_product.so jnerf_amd/csrc/libngp_hip.so
