#!/bin/bash
# Builds jnerf_amd/csrc/libngp_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -Wall -Wno-unused-function -Wno-unused-variable"
mkdir -p build
pids=()
for f in *.hip; do
  o=build/${f%.hip}.o
  stale=0
  for h in *.h ../../include/ngp_hip.h; do [ "$h" -nt "$o" ] && stale=1; done
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ $stale = 1 ]; then
    /opt/rocm/bin/hipcc $FLAGS -c "$f" -o "$o" $EXTRA &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o libngp_hip.so
echo "built $(pwd)/libngp_hip.so"
