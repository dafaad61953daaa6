#!/bin/bash
# A/B of an environment switch of libgpx.so on ONE box; usage: run_ab_env.sh VAR "<bench.py args>" [rounds]
# prints ms_per_step with VAR=1 ("old") and without ("new"), alternating
set -e
cd "$(dirname "$0")/../.."
var=$1; args=$2; rounds=${3:-2}
for i in $(seq $rounds); do
  for v in 1 0; do
    if [ $v = 1 ]; then export $var=1; else unset $var; fi
    timeout -k 10 300 python bench.py $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', '$args', d['ms_per_step'], d.get('nll'))"
  done
done
