#!/bin/bash
# A/B of two builds of libgpx.so on ONE box, sparse workloads: old, new, old, new
set -e
cd "$(dirname "$0")/../.."
cp gp_amd/libgpx.so /tmp/libgpx_keep.so
for v in old new old new; do
  cp tools/ab/libgpx_$v.so gp_amd/libgpx.so
  echo "== $v"
  for w in fitc vfe; do
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['ms_per_step'])"
  done
done
cp /tmp/libgpx_keep.so gp_amd/libgpx.so
