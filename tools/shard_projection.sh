#!/bin/bash
# One GPU: the sparse legs at the shard sizes of N = 1, 2, 4, 8 ranks with the stream-ordered RCCL hook active (world 1), i.e.
# everything of an N-rank evaluation except the wire time of the collectives and the skew between ranks.
# Prints t(n / N) and the bound t(n) / (N t(n / N)) on the parallel efficiency.  usage: tools/shard_projection.sh > out.json
cd "$(dirname "$0")/.."
export GPX_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
python3 - <<'PY'
import json, subprocess, sys
out = {}
for model, n0, arg, steps in (("fitc", 1 << 20, "--fitc-n", ["--sparse-steps", "10"]), ("vfe", 1 << 22, "--vfe-n", ["--vfe-steps", "4"])):
    rows = {}
    for N in (1, 2, 4, 8):
        cmd = [sys.executable, "bench.py", "--workload", model, arg, str(n0 // N), "--no-cpu-baseline"] + steps
        line = subprocess.run(cmd, capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1]
        d = json.loads(line)[model]
        rows[N] = {"rows_per_rank": n0 // N, "ms_per_eval": d["ms_per_step"], "allreduces_per_eval": d.get("allreduces_per_eval"),
                   "allreduce_doubles": d.get("allreduce_doubles"), "allreduce_ms_world1": d.get("allreduce_ms_per_eval")}
        print(model, N, rows[N], file=sys.stderr, flush=True)
    t1 = rows[1]["ms_per_eval"]
    for N in rows:
        rows[N]["efficiency_bound_before_wire_time"] = t1 / (N * rows[N]["ms_per_eval"])
    out[model] = rows
out["note"] = ("one GPU, RCCL world size 1 through torch.distributed (the hook path of bench.py): t(n/N) is one rank's evaluation at the "
               "N-rank shard size with every hook call made; NOT a multi-GPU measurement - the wire time of the collectives and the "
               "skew between ranks come on top")
print(json.dumps(out, indent=1))
PY
