set -e
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in 0 224 192 160; do
  export GPX_BUILD_THROTTLE=$v
  timeout -k 10 300 python bench.py --workload fitc --fitc-n 131072 --sparse-steps 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('throttle=$v', d['ms_per_step'], d['fitc']['nll'])"
done
done
