#!/bin/bash
# A/B of builds of libgpx.so on ONE box; usage: run_ab_any.sh "<bench.py args>" <tag> <tag> ... (tools/ab/libgpx_<tag>.so)
set -e
cd "$(dirname "$0")/../.."
args=$1; shift
cp gp_amd/libgpx.so /tmp/libgpx_keep.so
for v in "$@"; do
  cp tools/ab/libgpx_$v.so gp_amd/libgpx.so
  echo "== $v"
  timeout -k 10 300 python bench.py $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
cp /tmp/libgpx_keep.so gp_amd/libgpx.so
