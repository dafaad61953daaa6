#!/usr/bin/env python3
"""Is a sharded-vs-single gradient difference rounding under an ill-conditioned Kuu, or a sharding error?  FITC at the
two-rank rehearsal's conditions (d = 6, inducing points = the first m training rows, jitter 1e-6): single context, two
row shards (threads stand in for ranks), the CPU oracle (literal reference order) and the NumPy/LAPACK evaluation.
usage: python tools/study_shard_grad.py [n=4000] [m=1024]"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import np_gp  # noqa: E402
import oracle  # noqa: E402
from gp_amd import lib as gpx, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
X, y, hyp, _ = synth.make_problem(3, n, 6)
U = X[:m].copy()
jit = 1e-6


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / np.max(np.abs(b))


ref = gpx.Context(); ref.set_data(X, y); ref.set_inducing(U)
nll1, g1 = ref.nll_grad(gpx.MODEL_FITC, 0, hyp, jitter=jit)
h = n // 2
halves = [(X[:h], y[:h]), (X[h:], y[h:])]
bar, slots, out = threading.Barrier(2), [None, None], [None, None]


def make_cb(rank):
    def cb(ptr, count):
        host = np.empty(count)
        gpx.check(gpx.lib.gpx_memcpy_d2h(host.ctypes.data, ptr, count * 8), "d2h")
        slots[rank] = host
        bar.wait()
        tot = slots[0] + slots[1]
        bar.wait()
        gpx.check(gpx.lib.gpx_memcpy_h2d(ptr, tot.ctypes.data, count * 8), "h2d")
    return cb


def worker(rank):
    ctx = gpx.Context(); ctx.set_data(*halves[rank]); ctx.set_inducing(U); ctx.set_allreduce(make_cb(rank))
    out[rank] = ctx.nll_grad(gpx.MODEL_FITC, 0, hyp, jitter=jit)
    ctx.close()


th = [threading.Thread(target=worker, args=(r,)) for r in range(2)]
[t.start() for t in th]; [t.join() for t in th]
nll2, g2 = out[0]
print(f"n={n} m={m}: sharded vs single: NLL {abs(nll2 - nll1) / abs(nll1):.2e} grad {rel(g2, g1):.2e}")
print("grad single :", np.array2string(g1, precision=6))
print("grad sharded:", np.array2string(g2, precision=6))
Kuu = np_gp.cov_k(0, hyp, U, U) + jit * np.eye(m)
print(f"cond(Kuu + j I) = {np.linalg.cond(Kuu):.2e}")
nll_n, g_n = np_gp.fitc_nll(0, X, y, U, jit, hyp)
print(f"NumPy/LAPACK: NLL single {abs(nll1 - nll_n) / abs(nll_n):.2e}  grad single {rel(g1, g_n):.2e}  grad sharded {rel(g2, g_n):.2e}")
if n * m * m <= 8e9:
    nll_o, g_o = oracle.fitc_nll_all_cores(0, X, y, U, jit, hyp)
    print(f"oracle: NLL single {abs(nll1 - nll_o) / abs(nll_o):.2e}  grad single {rel(g1, g_o):.2e}  grad sharded {rel(g2, g_o):.2e}")
