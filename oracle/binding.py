"""ctypes binding of oracle/libgporacle.so (built by `make -C oracle`).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgporacle.so")

COV_SE_ARD, COV_SE_ISO = 0, 1


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


_lib = _load()
_dp = C.POINTER(C.c_double)
_l = C.c_long
_i = C.c_int
_d = C.c_double


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


_lib.orc_cov_num_hyp.restype = _i
_lib.orc_cov_k.argtypes = [_i, _i, _dp, _dp, _l, _dp, _l, _dp]
_lib.orc_cov_dk_dhyp.argtypes = [_i, _i, _dp, _i, _dp, _l, _dp, _l, _dp, _dp]
_lib.orc_cov_dk_dx1.argtypes = [_i, _i, _dp, _dp, _dp, _l, _dp]
_lib.orc_default_hyps.argtypes = [_i, _i, _l, _dp, _dp, _d, _dp]
_lib.orc_hyp_range.argtypes = [_i, _i, _l, _dp, _dp, _d, _dp, _dp]
_lib.orc_chol.argtypes = [_dp, _l]
_lib.orc_chol.restype = _i
_lib.orc_chol_solve.argtypes = [_dp, _l, _dp, _l]
_lib.orc_chol_logdet.argtypes = [_dp, _l]
_lib.orc_chol_logdet.restype = _d
_lib.orc_chol_inverse.argtypes = [_dp, _l, _dp]
_lib.orc_exact_nll.argtypes = [_i, _i, _l, _dp, _dp, _dp, _i, _dp]
_lib.orc_exact_nll.restype = _d
_lib.orc_exact_setk.argtypes = [_i, _i, _l, _dp, _dp, _dp, _i]
_lib.orc_exact_setk.restype = C.c_void_p
_lib.orc_exact_free.argtypes = [C.c_void_p]
_lib.orc_exact_predict.argtypes = [C.c_void_p, _dp, _l, _dp, _dp, _dp, _dp]
for _n in ("orc_fitc_nll", "orc_vfe_nll"):
    getattr(_lib, _n).argtypes = [_i, _i, _l, _dp, _dp, _dp, _l, _d, _dp, _i, _dp]
    getattr(_lib, _n).restype = _d
for _n in ("orc_fitc_setk", "orc_vfe_setk"):
    getattr(_lib, _n).argtypes = [_i, _i, _l, _dp, _dp, _dp, _l, _dp, _dp]
    getattr(_lib, _n).restype = C.c_void_p
_lib.orc_sparse_free.argtypes = [C.c_void_p]
_lib.orc_fitc_predict.argtypes = [C.c_void_p, _dp, _l, _dp, _dp, _dp, _dp]
_lib.orc_vfe_predict.argtypes = [C.c_void_p, _dp, _l, _dp, _dp, _dp]


def num_hyp(cov, d):
    """number of kernel hypers + 2 (noise, mean)"""
    return _lib.orc_cov_num_hyp(cov, d) + 2


def cov_k(cov, hyp, x1, x2):
    """x1: (p, d) row = point (C-contiguous == d x p column-major).  Returns K (p, q)."""
    x1, x2, hyp = _f(x1), _f(x2), _f(hyp)
    p, d = x1.shape
    q = x2.shape[0]
    K = np.empty((q, p))  # column-major p x q
    _lib.orc_cov_k(cov, d, _p(hyp), _p(x1), p, _p(x2), q, _p(K))
    return K.T


def cov_dk_dhyp(cov, hyp, idx, x1, x2, K):
    x1, x2, hyp = _f(x1), _f(x2), _f(hyp)
    p, d = x1.shape
    q = x2.shape[0]
    Kc = _f(np.asarray(K).T)
    dK = np.empty((q, p))
    _lib.orc_cov_dk_dhyp(cov, d, _p(hyp), idx, _p(x1), p, _p(x2), q, _p(Kc), _p(dK))
    return dK.T


def default_hyps(cov, X, y, noise_lb=1e-3):
    X, y = _f(X), _f(y)
    n, d = X.shape
    hyp = np.empty(num_hyp(cov, d))
    _lib.orc_default_hyps(cov, d, n, _p(X), _p(y), noise_lb, _p(hyp))
    return hyp


def hyp_range(cov, X, y, noise_lb=1e-3):
    X, y = _f(X), _f(y)
    n, d = X.shape
    lb = np.empty(num_hyp(cov, d))
    ub = np.empty_like(lb)
    _lib.orc_hyp_range(cov, d, n, _p(X), _p(y), noise_lb, _p(lb), _p(ub))
    return lb, ub


def chol(A):
    """Returns (info, L) with L lower-triangular (numpy row/col convention)."""
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    Ac = _f(A.T)  # column-major copy
    info = _lib.orc_chol(_p(Ac), n)
    return info, np.tril(Ac.T)


def exact_nll(cov, X, y, hyp, calc_grad=True):
    X, y, hyp = _f(X), _f(y), _f(hyp)
    n, d = X.shape
    g = np.zeros(len(hyp))
    nll = _lib.orc_exact_nll(cov, d, n, _p(X), _p(y), _p(hyp), int(calc_grad), _p(g))
    return (nll, g) if calc_grad else nll


class ExactState:
    def __init__(self, cov, X, y, hyp, literal_spd_check=False):
        X, y = _f(X), _f(y)
        self.hyp = _f(hyp).copy()
        n, self.d = X.shape
        self._h = _lib.orc_exact_setk(cov, self.d, n, _p(X), _p(y), _p(self.hyp), int(literal_spd_check))

    def predict(self, Xs, grad=False):
        Xs = _f(Xs)
        t = Xs.shape[0]
        mu, s2 = np.empty(t), np.empty(t)
        gy = np.empty((t, self.d)) if grad else None
        gs2 = np.empty((t, self.d)) if grad else None
        _lib.orc_exact_predict(self._h, _p(Xs), t, _p(mu), _p(s2), _p(gy), _p(gs2))
        return (mu, s2, gy, gs2) if grad else (mu, s2)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.orc_exact_free(self._h)
            self._h = None


def _sparse_nll(fn, cov, X, y, U, jitter, hyp, calc_grad):
    X, y, U, hyp = _f(X), _f(y), _f(U), _f(hyp)
    n, d = X.shape
    m = U.shape[0]
    g = np.zeros(len(hyp))
    nll = fn(cov, d, n, _p(X), _p(y), _p(U), m, jitter, _p(hyp), int(calc_grad), _p(g))
    return (nll, g) if calc_grad else nll


def fitc_nll(cov, X, y, U, jitter, hyp, calc_grad=True):
    return _sparse_nll(_lib.orc_fitc_nll, cov, X, y, U, jitter, hyp, calc_grad)


def vfe_nll(cov, X, y, U, jitter, hyp, calc_grad=True):
    return _sparse_nll(_lib.orc_vfe_nll, cov, X, y, U, jitter, hyp, calc_grad)


class SparseState:
    def __init__(self, model, cov, X, y, U, jitter, hyp):
        X, y, U = _f(X), _f(y), _f(U)
        self.hyp = _f(hyp).copy()
        self.model = model
        n, self.d = X.shape
        j = C.c_double(jitter)
        fn = _lib.orc_fitc_setk if model == "fitc" else _lib.orc_vfe_setk
        self._h = fn(cov, self.d, n, _p(X), _p(y), _p(U), U.shape[0], C.byref(j), _p(self.hyp))
        self.jitter = j.value

    def predict(self, Xs, grad=False):
        Xs = _f(Xs)
        t = Xs.shape[0]
        mu, s2 = np.empty(t), np.empty(t)
        gy = np.empty((t, self.d)) if grad else None
        if self.model == "fitc":
            gs2 = np.empty((t, self.d)) if grad else None
            _lib.orc_fitc_predict(self._h, _p(Xs), t, _p(mu), _p(s2), _p(gy), _p(gs2))
            return (mu, s2, gy, gs2) if grad else (mu, s2)
        _lib.orc_vfe_predict(self._h, _p(Xs), t, _p(mu), _p(s2), _p(gy))
        return (mu, s2, gy) if grad else (mu, s2)

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.orc_sparse_free(self._h)
            self._h = None


# ---- the OpenMP build of the SAME restatement: bench.py's "all host cores" CPU timing only ---------
_SO_OMP = os.path.join(_HERE, "libgporacle_omp.so")
_omp = None


def _load_omp():
    global _omp
    if _omp is None:
        if not os.path.exists(_SO_OMP):
            build()
        _omp = C.CDLL(_SO_OMP)
        _omp.orc_exact_nll.argtypes = _lib.orc_exact_nll.argtypes
        _omp.orc_exact_nll.restype = _d
        for name in ("orc_fitc_nll", "orc_vfe_nll"):
            getattr(_omp, name).argtypes = _lib.orc_fitc_nll.argtypes
            getattr(_omp, name).restype = _d
    return _omp


def exact_nll_all_cores(cov, X, y, hyp):
    """exact_nll from libgporacle_omp.so (thread count: OMP_NUM_THREADS, else every core the process may use)."""
    lib = _load_omp()
    X, y, hyp = _f(X), _f(y), _f(hyp)
    n, d = X.shape
    g = np.zeros(len(hyp))
    return lib.orc_exact_nll(cov, d, n, _p(X), _p(y), _p(hyp), 1, _p(g)), g


def fitc_nll_all_cores(cov, X, y, U, jitter, hyp):
    return _sparse_nll(_load_omp().orc_fitc_nll, cov, X, y, U, jitter, hyp, True)


def vfe_nll_all_cores(cov, X, y, U, jitter, hyp):
    return _sparse_nll(_load_omp().orc_vfe_nll, cov, X, y, U, jitter, hyp, True)
