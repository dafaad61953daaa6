#!/bin/bash
# Builds gp_amd/libgpx.so for gfx950 (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libgpx.so
mkdir -p obj
pids=()
for f in *.hip; do
  o=obj/${f%.hip}.o
  stale=0
  for h in *.h ../../include/gpx.h; do [ "$h" -nt "$o" ] && stale=1; done   # every header: tile_order.h, gemm_flags.h ... too
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ $stale = 1 ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" obj/*.o
echo "built $(realpath $OUT)"
