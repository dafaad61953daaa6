#!/bin/bash
# paired small products of the recursion: off / thresholds on the total workgroup count of a pair (one box)
cd "$(dirname "$0")/../.."
run() { timeout -k 10 200 python bench.py --workload exact --n 8192 --d 10 --steps 30 --warmup 3 --no-cpu-baseline --predict-t 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['nll'])"; }
for rep in 1 2; do
  GPX_NO_GEMM_PAIR=1 run off
  GPX_PAIR_MAX_GRID=128 run max128
  GPX_PAIR_MAX_GRID=512 run max512
  GPX_PAIR_MAX_GRID=2048 run max2048
  run all
done
