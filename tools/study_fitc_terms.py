"""Where the FITC NLL loses its digits at the benchmark's own conditions (n = 4096, m = 1024, d = 20, jitter 1e-8,
inducing points = the first m training rows), term by term against an 80-bit evaluation.

  python tools/study_fitc_terms.py cpu     NumPy/LAPACK variants (A as the reference forms it, A formed in 80 bits and
                                           rounded, the whitened M = sn2 I + V^T G^-1 V) against the 80-bit terms below
  python tools/study_fitc_terms.py gpu     the library (GPX_DEBUG_TERMS prints its terms); run once per switch:
                                           GPX_FITC_NO_WHITEN=1, GPX_EXACT_PIVOT=1, GPX_BASE64=1

80-bit terms (tools/longdouble_fitc.py arithmetic, printed per term): the data-fit term (rGr - Fa) / sn2 multiplies the
error of Fa = F^T A^-1 F by 1 / sn2 = 100, and cond(A) = 1.2e9."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import np_gp
from gp_amd import synth

LD = dict(ldA=-4183.4892511746902004, ldU=-2297.2097041136928852, slg=5823.8252689483502587,
          rGr=1345.4209295916302364, Fa=1308.8910850442738195, nll=485.69591464012831117)
n, m, d, seed, jit = 4096, 1024, 20, 20240601 + 3, 1e-8
hyp = synth.eval_hyps(d)
X = synth.make_inputs(seed, n, d)
y = synth.make_targets(seed, X, hyp)
U = X[:m].copy()
sn2 = np.exp(2 * hyp[d + 1])


def report(tag, ldAU, slg, rGr, Fa):
    nll = 0.5 * ((rGr - Fa) / sn2 + ldAU + slg + (n - m) * np.log(sn2) + n * np.log(2 * np.pi))
    print(f"{tag:34s} ldA-ldU {ldAU - (LD['ldA'] - LD['ldU']):+.2e} slg {slg - LD['slg']:+.2e} rGr {rGr - LD['rGr']:+.2e} "
          f"Fa {Fa - LD['Fa']:+.2e} | NLL abs {nll - LD['nll']:+.2e} rel {abs(nll - LD['nll']) / LD['nll']:.2e}")


if sys.argv[1] == "cpu":
    from scipy.linalg import solve_triangular
    ell, sf2, mean = np.exp(hyp[:d]), np.exp(2 * hyp[d]), hyp[d + 2]
    Kuu = sf2 * np.exp(-0.5 * np_gp.sqd_per_dim(U, U, ell).sum(0)) + jit * np.eye(m)
    K = sf2 * np.exp(-0.5 * np_gp.sqd_per_dim(X, U, ell).sum(0))
    r = y - mean
    Lu = np.linalg.cholesky(Kuu)
    V = solve_triangular(Lu, K.T, lower=True).T
    Gam = (sn2 + sf2 - (V * V).sum(1)) / sn2
    g_ = 1 / Gam
    p = g_ * r
    F = K.T @ p
    ldU = 2 * np.log(np.diag(Lu)).sum()
    L_ = np.longdouble
    A64 = sn2 * Kuu + K.T @ (g_[:, None] * K)
    A80 = (L_(sn2) * Kuu.astype(L_) + K.astype(L_).T @ (g_.astype(L_)[:, None] * K.astype(L_))).astype(np.float64)
    for tag, A in (("A formed in fp64 (reference)", A64), ("A formed in 80 bits, rounded", A80)):
        LA = np.linalg.cholesky(A)
        z = solve_triangular(LA, F, lower=True)
        report(tag, 2 * np.log(np.diag(LA)).sum() - ldU, np.log(Gam).sum(), r @ p, z @ z)
    M = sn2 * np.eye(m) + V.T @ (g_[:, None] * V)
    LM = np.linalg.cholesky(M)
    z = solve_triangular(LM, V.T @ p, lower=True)
    report("whitened M = sn2 I + V^T G^-1 V", 2 * np.log(np.diag(LM)).sum(), np.log(Gam).sum(), r @ p, z @ z)
    print(f"cond(A) {np.linalg.cond(A64):.2e}  cond(M) {np.linalg.cond(M):.2e}  cond(Kuu + jI) {np.linalg.cond(Kuu):.2e}")
else:
    os.environ["GPX_DEBUG_TERMS"] = "1"
    from gp_amd import lib
    ctx = lib.Context()
    ctx.set_data(X, y)
    ctx.set_inducing(U)
    nll, g = ctx.nll_grad(lib.MODEL_FITC, 0, hyp, jitter=jit)
    ctx.close()
    nll_n, g_n = np_gp.fitc_nll(0, X, y, U, jit, hyp)
    sw = {k: os.environ.get(k) for k in ("GPX_FITC_NO_WHITEN", "GPX_EXACT_PIVOT", "GPX_BASE64") if os.environ.get(k)}
    print(f"switches {sw}: NLL {nll!r} rel err vs 80-bit {abs(nll - LD['nll']) / LD['nll']:.2e} vs LAPACK "
          f"{abs(nll - nll_n) / abs(nll_n):.2e}; gradient vs LAPACK {np.max(np.abs(g - g_n)) / np.max(np.abs(g_n)):.2e}")
