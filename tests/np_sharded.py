"""NumPy mirror of the PHASE structure of gp_amd/csrc/sparse.hip (test helper): what each rank
computes from its own rows, what is all-reduced, and what is replicated.  `allreduce(vec)` is
any in-place sum over ranks (identity for one rank; torch.distributed/gloo in the CPU test)."""
import numpy as np
from scipy.linalg import cho_solve, solve_triangular

import np_gp


def _trace_sums(cov, W, K, D):
    WK = W * K
    per_dim = (WK[None] * D).sum(axis=(1, 2))
    return per_dim, WK.sum()


def sparse_nll_grad_sharded(model, cov, X, y, U, jitter, hyp, allreduce):
    n_loc, d = X.shape
    m = U.shape[0]
    nk, ell, sf2, sn2, mean = np_gp.split_hyp(cov, d, hyp)
    Duu = np_gp.sqd_per_dim(U, U, ell)
    Dxu = np_gp.sqd_per_dim(X, U, ell)
    Kuu0 = sf2 * np.exp(-0.5 * Duu.sum(0))
    Kuu = Kuu0 + jitter * np.eye(m)
    K = sf2 * np.exp(-0.5 * Dxu.sum(0))
    r = y - mean
    Lu = np.linalg.cholesky(Kuu)
    fitc = model == "fitc"
    # ---- phase 1 (row-local) + all-reduce #1: [S (m*m) | F (m) | 3 scalars] -------------------
    if fitc:
        V = solve_triangular(Lu, K.T, lower=True).T
        Gam = (sn2 + sf2 - (V * V).sum(1)) / sn2
        g_ = 1.0 / Gam
        p = g_ * r
        Ksc = g_[:, None] * K
        # whitened partials (sparse.hip): V^T Gamma^-1 V and F_v = V^T Gamma^-1 r
        buf = np.concatenate([(V.T @ (g_[:, None] * V)).ravel(), V.T @ p, [np.log(Gam).sum(), r @ p, n_loc]])
    else:
        buf = np.concatenate([(K.T @ K).ravel(), K.T @ r, [r @ r, 0.0, n_loc]])
    allreduce(buf)
    S, F, tail = buf[:m * m].reshape(m, m), buf[m * m:m * m + m], buf[m * m + m:]
    n = tail[2]
    # ---- phase 2 (replicated) -------------------------------------------------------------------
    Kuuinv = cho_solve((Lu, True), np.eye(m))
    if fitc:
        # M = sn2 I + V^T Gamma^-1 V;  A = Lu M Lu^T, so log|A| - log|Kuu| = log|M|, F^T A^-1 F = |L_M^-1 F_v|^2,
        # a = Lu^-T M^-1 F_v and A^-1 = (L_M^-1 Lu^-1)^T (L_M^-1 Lu^-1)
        LM = np.linalg.cholesky(sn2 * np.eye(m) + S)
        z = solve_triangular(LM, F, lower=True)
        a = solve_triangular(Lu, solve_triangular(LM, z, lower=True, trans="T"), lower=True, trans="T")
        WA = solve_triangular(LM, np.linalg.inv(Lu), lower=True)
        Ainv = WA.T @ WA
        nll = 0.5 * ((tail[1] - z @ z) / sn2 + 2 * np.log(np.diag(LM)).sum() + tail[0] + (n - m) * np.log(sn2) + n * np.log(2 * np.pi))
    else:
        A = sn2 * Kuu + S
        LA = np.linalg.cholesky(A)
        a = cho_solve((LA, True), F)
        ldA, ldU = 2 * np.log(np.diag(LA)).sum(), 2 * np.log(np.diag(Lu)).sum()
        Ainv = cho_solve((LA, True), np.eye(m))
        f_tr = 0.5 * (n * sf2 - np.sum(Kuuinv * S)) / sn2
        nll = 0.5 * n * np.log(2 * np.pi) + 0.5 * (ldA - ldU + (n - m) * np.log(sn2)) + 0.5 * (tail[0] - F @ a) / sn2 + f_tr
    # ---- phase 3 (row-local) + all-reduce #2 ------------------------------------------------------
    Ka = K @ a
    if fitc:
        C = Ksc @ Ainv
        B = V @ np.linalg.inv(Lu)
        pq = p - g_ * Ka
        c1 = (C * Ksc).sum(1)
        w = g_ - c1 - pq * pq / sn2
        Whalf = C - np.outer(pq, a) / sn2 - (w[:, None] * B) / sn2
        per_dim, sK = _trace_sums(cov, Whalf, K, Dxu)
        buf2 = np.concatenate([(B.T @ (w[:, None] * B)).ravel(), per_dim, [sK, (g_ - c1).sum(), pq @ pq, pq.sum(), w.sum()]])
        allreduce(buf2)
        Mw, per_dim, t2 = buf2[:m * m].reshape(m, m), buf2[m * m:m * m + d], buf2[m * m + d:]
        Wuu = sn2 * Ainv - Kuuinv + np.outer(a, a) + Mw / sn2
        pu, sKu = _trace_sums(cov, Wuu, Kuu0, Duu)
        kern = 0.5 * (pu + 2 * per_dim)
        gsf = 0.5 * (2 * sKu + 2 * jitter * np.trace(Wuu) + 2 * 2 * t2[0]) + 0.5 * (2 * sf2 * t2[4] / sn2)
        gsn = t2[1] - t2[2] / sn2
        gmean = -t2[3] / sn2
    else:
        P = Ainv - Kuuinv / sn2
        alpha = (r - Ka) / sn2
        Wxu = K @ P - np.outer(alpha, a)
        per_dim, sK = _trace_sums(cov, Wxu, K, Dxu)
        buf2 = np.concatenate([per_dim, [sK, alpha @ alpha, alpha.sum()]])
        allreduce(buf2)
        per_dim, t2 = buf2[:d], buf2[d:]
        Wuu = 0.5 * (np.outer(a, a) + sn2 * Ainv - Kuuinv + Kuuinv @ S @ Kuuinv / sn2)
        pu, sKu = _trace_sums(cov, Wuu, Kuu0, Duu)
        kern = pu + per_dim
        gsf = 2 * sKu + 2 * jitter * np.trace(Wuu) + 2 * t2[0] + 0.5 * n * 2 * sf2 / sn2
        gsn = -sn2 * t2[1] + (n - np.sum(Ainv * S)) - 2 * f_tr
        gmean = -t2[2]
    g = np.empty(nk + 2)
    if cov == np_gp.ARD:
        g[:d] = kern
    else:
        g[0] = kern.sum()
    g[nk - 1], g[nk], g[nk + 1] = gsf, gsn, gmean
    return nll, g
