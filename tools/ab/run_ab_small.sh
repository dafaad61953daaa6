#!/bin/bash
# A/B of two builds of libgpx.so on ONE box, latency-bound configurations: exact n=8192 and the FITC shard size
set -e
cd "$(dirname "$0")/../.."
cp gp_amd/libgpx.so /tmp/libgpx_keep.so
for v in old new old new; do
  cp tools/ab/libgpx_$v.so gp_amd/libgpx.so
  echo "== $v"
  timeout -k 10 200 python bench.py --workload exact --n 8192 --d 10 --steps 30 --warmup 3 --no-cpu-baseline --predict-t 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exact8192', d['ms_per_step'], d['nll'])"
  timeout -k 10 200 python bench.py --workload fitc --fitc-n 131072 --sparse-steps 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fitc131072', d['ms_per_step'], d['fitc']['nll'])"
done
cp /tmp/libgpx_keep.so gp_amd/libgpx.so
