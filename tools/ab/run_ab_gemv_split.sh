set -e
cd $GRAFT_REPO_ROOT
for sp in 1 4 8 1 4 8; do
  export GPX_GEMV_SPLIT=$sp
  for args in "--workload fitc --fitc-n 131072 --sparse-steps 20 --no-cpu-baseline" "--workload fitc --sparse-steps 8 --no-cpu-baseline"; do
    timeout -k 10 300 python bench.py $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split=$sp', d['fitc']['config']['n'], d['ms_per_step'], d['fitc']['nll'])"
  done
done
for sp in 1 2 4; do
  export GPX_GEMV_SPLIT=$sp
  timeout -k 10 300 python bench.py --workload vfe --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split=$sp vfe', d['ms_per_step'], d['vfe']['nll'])"
done
