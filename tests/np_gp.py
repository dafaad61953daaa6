"""Independent NumPy/SciPy evaluation of the GP objectives in closed form (test helper).

This is NOT a restatement of the reference's operation order (that is oracle/gp_oracle.c).  It
evaluates the same quantities through the restructured O(n m^2) weight-matrix forms that the HIP
path uses, with SciPy's LAPACK Cholesky, so that oracle <-> numpy <-> finite
differences form a three-way check.  Inputs: X (n, d) row = point.
"""
import numpy as np
from scipy.linalg import cho_factor, cho_solve, solve_triangular

ARD, ISO = 0, 1


def split_hyp(cov, d, hyp):
    nk = d + 1 if cov == ARD else 2
    ell = np.exp(hyp[:d]) if cov == ARD else np.full(d, np.exp(hyp[0]))
    return nk, ell, np.exp(2 * hyp[nk - 1]), np.exp(2 * hyp[nk]), hyp[nk + 1]


class _LeanSqd:
    """Stand-in for the (d, p, q) array of sqd_per_dim when that array would not fit in memory (the
    benchmark's own shapes: n x m x d = 66000 x 2048 x 50): the same direct differences, one dimension
    at a time, p x q temporaries only."""

    def __init__(self, X1, X2, ell):
        self.a = np.ascontiguousarray((X1 / ell).T)
        self.b = np.ascontiguousarray((X2 / ell).T)

    def dim(self, i):
        t = self.a[i][:, None] - self.b[i][None, :]
        t *= t
        return t

    def sum(self, axis=0):
        assert axis == 0
        s = self.dim(0)
        for i in range(1, len(self.a)):
            s += self.dim(i)
        return s


def sqd_per_dim(X1, X2, ell):
    """(d, p, q) array of ((x1_i - x2_i)/l_i)^2 (a lazy per-dimension object above 2e8 elements)"""
    if X1.shape[0] * X2.shape[0] * X1.shape[1] > 2e8:
        return _LeanSqd(X1, X2, ell)
    a = (X1 / ell).T[:, :, None]
    b = (X2 / ell).T[:, None, :]
    return (a - b) ** 2


def cov_k(cov, hyp, X1, X2):
    d = X1.shape[1]
    _, ell, sf2, _, _ = split_hyp(cov, d, hyp)
    return sf2 * np.exp(-0.5 * sqd_per_dim(X1, X2, ell).sum(0))


def _kernel_grad_sums(cov, W, K, D):
    """sum_ab W_ab dK_ab/dtheta_i for every kernel hyper, K and D = sqd_per_dim given."""
    WK = W * K
    if isinstance(D, _LeanSqd):
        per_dim = np.array([np.sum(WK * D.dim(i)) for i in range(len(D.a))])
    else:
        per_dim = (WK[None] * D).sum(axis=(1, 2))
    if cov == ARD:
        return np.concatenate([per_dim, [2 * WK.sum()]])
    return np.array([per_dim.sum(), 2 * WK.sum()])


def exact_nll(cov, X, y, hyp):
    n, d = X.shape
    nk, ell, sf2, sn2, mean = split_hyp(cov, d, hyp)
    D = sqd_per_dim(X, X, ell)
    K = sf2 * np.exp(-0.5 * D.sum(0))
    Ky = K + sn2 * np.eye(n)
    try:
        c = cho_factor(Ky, lower=True)
    except np.linalg.LinAlgError:
        return np.inf, np.full(nk + 2, np.inf)
    r = y - mean
    alpha = cho_solve(c, r)
    nll = 0.5 * r @ alpha + np.log(np.diag(c[0])).sum() + 0.5 * n * np.log(2 * np.pi)
    Q = cho_solve(c, np.eye(n)) - np.outer(alpha, alpha)
    g = np.empty(nk + 2)
    g[:nk] = 0.5 * _kernel_grad_sums(cov, Q, K, D)
    g[nk] = sn2 * np.trace(Q)
    g[nk + 1] = -alpha.sum()
    return nll, g


def exact_predict(cov, X, y, hyp, Xs):
    n, d = X.shape
    nk, ell, sf2, sn2, mean = split_hyp(cov, d, hyp)
    Ky = cov_k(cov, hyp, X, X) + sn2 * np.eye(n)
    L = np.linalg.cholesky(Ky)
    alpha = cho_solve((L, True), y - mean)
    Ks = cov_k(cov, hyp, Xs, X)
    V = solve_triangular(L, Ks.T, lower=True)
    mu = mean + Ks @ alpha
    s2 = np.maximum(sf2 - (V * V).sum(0), 0) + sn2
    return mu, s2


def fitc_nll(cov, X, y, U, jitter, hyp):
    n, d = X.shape
    m = U.shape[0]
    nk, ell, sf2, sn2, mean = split_hyp(cov, d, hyp)
    Duu = sqd_per_dim(U, U, ell)
    Dxu = sqd_per_dim(X, U, ell)
    Kuu0 = sf2 * np.exp(-0.5 * Duu.sum(0))
    Kuu = Kuu0 + jitter * np.eye(m)
    K = sf2 * np.exp(-0.5 * Dxu.sum(0))
    r = y - mean
    Lu = np.linalg.cholesky(Kuu)
    V = solve_triangular(Lu, K.T, lower=True).T            # n x m, V V^T = K Kuu^-1 K^T
    Gam = (sn2 + sf2 - (V * V).sum(1)) / sn2
    g_ = 1.0 / Gam
    A = sn2 * Kuu + K.T @ (g_[:, None] * K)
    LA = np.linalg.cholesky(A)
    p = g_ * r
    F = K.T @ p
    a = cho_solve((LA, True), F)
    nll = 0.5 * ((r @ p - F @ a) / sn2 + 2 * np.log(np.diag(LA)).sum() - 2 * np.log(np.diag(Lu)).sum()
                 + np.log(Gam).sum() + (n - m) * np.log(sn2) + n * np.log(2 * np.pi))
    # gradient through the weight matrices
    Ainv = cho_solve((LA, True), np.eye(m))
    Kuuinv = cho_solve((Lu, True), np.eye(m))
    B = solve_triangular(Lu, V.T, lower=True, trans="T").T  # K Kuu^-1
    C = K @ Ainv
    q = g_ * (K @ a)
    pq = p - q
    c1 = g_ * g_ * (C * K).sum(1)
    w = g_ - c1 - pq * pq / sn2
    Wxu = 2 * g_[:, None] * C - (2 / sn2) * np.outer(pq, a) - (2 / sn2) * w[:, None] * B
    Wuu = sn2 * Ainv - Kuuinv + np.outer(a, a) + (B.T @ (w[:, None] * B)) / sn2
    g = np.empty(nk + 2)
    # dKuu is taken of the jittered Kuu (FITC.cpp:271): the sf hyper sees 2*(Kuu0 + jitter*I)
    g[:nk] = 0.5 * (_kernel_grad_sums(cov, Wuu, Kuu0, Duu) + _kernel_grad_sums(cov, Wxu, K, Dxu))
    g[nk - 1] += 0.5 * (2 * jitter * np.trace(Wuu) + 2 * sf2 * w.sum() / sn2)
    g[nk] = (g_ - c1).sum() - pq @ pq / sn2
    g[nk + 1] = -pq.sum() / sn2
    return nll, g


def vfe_nll(cov, X, y, U, jitter, hyp):
    n, d = X.shape
    m = U.shape[0]
    nk, ell, sf2, sn2, mean = split_hyp(cov, d, hyp)
    Duu = sqd_per_dim(U, U, ell)
    Dxu = sqd_per_dim(X, U, ell)
    Kuu0 = sf2 * np.exp(-0.5 * Duu.sum(0))
    Kuu = Kuu0 + jitter * np.eye(m)
    K = sf2 * np.exp(-0.5 * Dxu.sum(0))
    r = y - mean
    S = K.T @ K
    A = sn2 * Kuu + S
    Lu = np.linalg.cholesky(Kuu)
    LA = np.linalg.cholesky(A)
    a = cho_solve((LA, True), K.T @ r)
    alpha = (r - K @ a) / sn2
    Kuuinv = cho_solve((Lu, True), np.eye(m))
    Ainv = cho_solve((LA, True), np.eye(m))
    f_tr = 0.5 * (n * sf2 - np.sum(Kuuinv * S)) / sn2
    nll = (0.5 * n * np.log(2 * np.pi)
           + 0.5 * (2 * np.log(np.diag(LA)).sum() - 2 * np.log(np.diag(Lu)).sum() + (n - m) * np.log(sn2))
           + 0.5 * r @ alpha + f_tr)
    P = Ainv - Kuuinv / sn2
    Wxu = K @ P - np.outer(alpha, a)
    Wuu = 0.5 * (np.outer(a, a) + sn2 * Ainv - Kuuinv + Kuuinv @ S @ Kuuinv / sn2)
    g = np.empty(nk + 2)
    g[:nk] = _kernel_grad_sums(cov, Wuu, Kuu0, Duu) + _kernel_grad_sums(cov, Wxu, K, Dxu)
    g[nk - 1] += 2 * jitter * np.trace(Wuu) + 0.5 * n * 2 * sf2 / sn2
    g[nk] = -sn2 * alpha @ alpha + (n - np.sum(Ainv * S)) - 2 * f_tr
    g[nk + 1] = -alpha.sum()
    return nll, g


def finite_diff_grad(f, hyp, eps):
    """central differences, as GP.cpp:486-507"""
    g = np.empty_like(hyp)
    for i in range(len(hyp)):
        hp, hm = hyp.copy(), hyp.copy()
        hp[i] += eps
        hm[i] -= eps
        g[i] = (f(hp) - f(hm)) / (2 * eps)
    return g
