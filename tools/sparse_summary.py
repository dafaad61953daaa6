"""one line per bench.py JSON file: sparse-model timing, GEMM share and the pairwise-kernel HBM rates"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    for k in ("fitc", "vfe"):
        if k in d:
            s = d[k]
            print(f, k, "ms", round(s["ms_per_step"], 2), "nll", s["nll"], "gemm_ms", round(s["gemm_ms_per_eval_rank0"], 2),
                  [(r["kernel"][:12], round(r["kernel_ms_per_eval"], 2), round(r["achieved"])) for r in s["hbm_rooflines"]])
