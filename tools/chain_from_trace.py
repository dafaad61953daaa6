#!/usr/bin/env python3
"""Duration of every factor-and-invert chain (gpx_potrf_inv) in one evaluation of a rocprofv3 kernel trace: first
potrf_inv_base_kernel start to the end of the last GEMM before the next non-chain kernel, per chain.
usage: tools/chain_from_trace.py <run_kernel_trace.csv> [eval_index=-1]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
starts = [i for i, r in enumerate(rows) if "scale_inputs_kernel" in r["Kernel_Name"]]
evals = [s for k, s in enumerate(starts) if k == 0 or s - starts[k - 1] > 8]
lo = evals[which]
hi = evals[evals.index(lo) + 1] if evals.index(lo) + 1 < len(evals) else len(rows)
sel = rows[lo:hi]
# a chain: the dispatches of ONE stream between its first base block and the last gemm<2,2> / gemm<1,1> that follows a base block
by_stream = {}
for r in sel:
    by_stream.setdefault(r["Stream_Id"], []).append(r)
for sid, rs in by_stream.items():
    chain = None
    for r in rs:
        n = r["Kernel_Name"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "potrf_inv_base_kernel" in n:
            if chain is None:
                chain = [s, e, 0, 0, 0.0, 0.0]
            chain[2] += 1
            chain[4] += e - s
            chain[1] = e
        elif chain is not None and ("gemm_kernel<2, 2" in n or "gemm_kernel<1, 1" in n):
            chain[3] += 1
            chain[5] += e - s
            chain[1] = e
        elif chain is not None:
            print(f"stream {sid}: chain of {chain[2]} base blocks + {chain[3]} small GEMMs: wall {(chain[1] - chain[0]) / 1e3:.1f} us "
                  f"(base blocks {chain[4] / 1e3:.1f} us, GEMMs {chain[5] / 1e3:.1f} us)")
            chain = None
    if chain is not None:
        print(f"stream {sid}: chain of {chain[2]} base blocks + {chain[3]} small GEMMs: wall {(chain[1] - chain[0]) / 1e3:.1f} us "
              f"(base blocks {chain[4] / 1e3:.1f} us, GEMMs {chain[5] / 1e3:.1f} us)")
