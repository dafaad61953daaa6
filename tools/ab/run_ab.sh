#!/bin/bash
# A/B of two builds of libgpx.so on ONE box (box-to-box variance is +-2 %): old, new, old, new
set -e
cd "$(dirname "$0")/../.."
cp gp_amd/libgpx.so /tmp/libgpx_keep.so
for v in old new old new; do
  cp tools/ab/libgpx_$v.so gp_amd/libgpx.so
  echo "== $v"
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'])"
done
cp /tmp/libgpx_keep.so gp_amd/libgpx.so
