"""ctypes binding of gp_amd/libgpx.so (the C ABI declared in include/gpx.h).

There is no CPU fallback: importing this module raises if the HIP library is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libgpx.so")
if not os.path.exists(SO_PATH):
    raise ImportError(f"{SO_PATH} not found: build it with gp_amd/csrc/build.sh (no CPU fallback exists)")
lib = C.CDLL(SO_PATH)

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_i64 = C.c_int64
_i = C.c_int
_d = C.c_double

GEMM_LOWER, GEMM_KLO_M, GEMM_KLO_N, GEMM_KHI_M, GEMM_KHI_N = 1, 2, 4, 8, 16
GEMM_HEAVY_LAST, GEMM_TILE64, GEMM_TILE128 = 32, 64, 128
GEMM_NO_SQUARES, GEMM_FORCE_SQUARES, GEMM_NO_BAND, GEMM_NO_TILE32, GEMM_NO_HALF_SKIP = 1024, 2048, 4096, 8192, 16384


class GpxError(RuntimeError):
    pass


def check(status, what=""):
    if status != 0:
        raise GpxError(f"gpx call {what} failed with status {status}")


lib.gpx_malloc.argtypes = [C.POINTER(_vp), _i64]
lib.gpx_free.argtypes = [_vp]
lib.gpx_memcpy_h2d.argtypes = [_vp, _vp, _i64]
lib.gpx_memcpy_d2h.argtypes = [_vp, _vp, _i64]
lib.gpx_memset_zero.argtypes = [_vp, _i64]
lib.gpx_set_device.argtypes = [_i]
lib.gpx_device_count.argtypes = [C.POINTER(_i)]
lib.gpx_dgemm.argtypes = [_i, _i, _i64, _i64, _i64, _d, _vp, _i64, _vp, _i64, _d, _vp, _i64, _i]
lib.gpx_dgemm_splitk.argtypes = lib.gpx_dgemm.argtypes + [_vp, _i64]
lib.gpx_dgemm_kscale.argtypes = [_i, _i64, _i64, _i64, _d, _vp, _i64, _vp, _vp, _i64, _d, _vp, _i64, _i, _vp, _i64]


def device_count():
    n = _i(0)
    check(lib.gpx_device_count(C.byref(n)), "device_count")
    return n.value


class DeviceBuffer:
    """A hipMalloc'd buffer of float64, filled from / read back to NumPy arrays."""

    def __init__(self, count):
        self.count = int(count)
        p = _vp()
        check(lib.gpx_malloc(C.byref(p), self.count * 8), "malloc")
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = cls(a.size)
        check(lib.gpx_memcpy_h2d(b.ptr, a.ctypes.data, a.size * 8), "h2d")
        return b

    def zero(self):
        check(lib.gpx_memset_zero(self.ptr, self.count * 8), "memset")

    def to_numpy(self, shape=None):
        out = np.empty(self.count, dtype=np.float64)
        check(lib.gpx_memcpy_d2h(out.ctypes.data, self.ptr, self.count * 8), "d2h")
        return out if shape is None else out.reshape(shape)

    def free(self):
        if self.ptr:
            lib.gpx_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def colmajor(a):
    """NumPy (rows, cols) array -> flat column-major copy."""
    return np.asfortranarray(a, dtype=np.float64).ravel(order="K")


def dgemm(A, B, C_in=None, a_kc=False, b_kc=False, alpha=1.0, beta=0.0, flags=0, splitk_work=0, kscale=None):
    """Test helper: A is (M, K), B is (N, K) as NumPy arrays; computes alpha*A@B.T + beta*C on the GPU.
    a_kc / b_kc choose the device storage of the operand (K-contiguous or M/N-contiguous)."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    M, K = A.shape
    N = B.shape[0]
    dA = DeviceBuffer.from_numpy(np.ascontiguousarray(A) if a_kc else colmajor(A))
    dB = DeviceBuffer.from_numpy(np.ascontiguousarray(B) if b_kc else colmajor(B))
    Cm = np.zeros((M, N)) if C_in is None else np.asarray(C_in, dtype=np.float64)
    dC = DeviceBuffer.from_numpy(colmajor(Cm))
    lda = K if a_kc else M
    ldb = K if b_kc else N
    if kscale is not None:
        assert a_kc, "kscale needs a K-contiguous A"
        ds = DeviceBuffer.from_numpy(np.ascontiguousarray(kscale, dtype=np.float64))
        w = DeviceBuffer(splitk_work) if splitk_work else None
        check(lib.gpx_dgemm_kscale(int(b_kc), M, N, K, alpha, dA.ptr, lda, ds.ptr, dB.ptr, ldb, beta, dC.ptr, M, flags,
                                   w.ptr if w else None, splitk_work), "dgemm_kscale")
    elif splitk_work:
        w = DeviceBuffer(splitk_work)
        check(lib.gpx_dgemm_splitk(int(a_kc), int(b_kc), M, N, K, alpha, dA.ptr, lda, dB.ptr, ldb, beta,
                                   dC.ptr, M, flags, w.ptr, splitk_work), "dgemm_splitk")
    else:
        check(lib.gpx_dgemm(int(a_kc), int(b_kc), M, N, K, alpha, dA.ptr, lda, dB.ptr, ldb, beta, dC.ptr, M,
                            flags), "dgemm")
    return dC.to_numpy().reshape(N, M).T.copy()


lib.gpx_dgemm_rowdot.argtypes = [_i64, _i64, _i64, _d, _vp, _i64, _vp, _i64, _vp, _i64, _i, _vp, _i64, _vp, _vp, _vp, _i64]


def dgemm_rowdot(A, B, E=None, evec=None, alpha=1.0, flags=0):
    """Test helper: C = alpha*A@B.T (A (M, K), B (N, K)) with the row-dot epilogue; returns C, R1 (parts, M) and
    R2 (parts, M) or None."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    M, K = A.shape
    N = B.shape[0]
    parts = 2 * ((N + 127) // 128)
    dA, dB = DeviceBuffer.from_numpy(colmajor(A)), DeviceBuffer.from_numpy(colmajor(B))
    dC, dR1 = DeviceBuffer(M * N), DeviceBuffer(parts * M)
    dE = DeviceBuffer.from_numpy(colmajor(E)) if E is not None else None
    dv = DeviceBuffer.from_numpy(np.asarray(evec, dtype=np.float64)) if evec is not None else None
    dR2 = DeviceBuffer(parts * M) if E is not None else None
    check(lib.gpx_dgemm_rowdot(M, N, K, alpha, dA.ptr, M, dB.ptr, N, dC.ptr, M, flags, dE.ptr if dE else None, M,
                               dv.ptr if dv else None, dR1.ptr, dR2.ptr if dR2 else None, M), "dgemm_rowdot")
    Cm = dC.to_numpy().reshape(N, M).T.copy()
    return Cm, dR1.to_numpy((parts, M)), (dR2.to_numpy((parts, M)) if dR2 else None)


def dgemm_opdot(A, B, evec, alpha=1.0, flags=0):
    """Test helper: C = alpha*A@B.T together with the operand dot R2 = A @ evec (gpx_dgemm_rowdot with R1 = E = NULL)."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    M, K = A.shape
    N = B.shape[0]
    dA, dB = DeviceBuffer.from_numpy(colmajor(A)), DeviceBuffer.from_numpy(colmajor(B))
    dC, dR2 = DeviceBuffer(M * N), DeviceBuffer(M)
    dv = DeviceBuffer.from_numpy(np.asarray(evec, dtype=np.float64))
    check(lib.gpx_dgemm_rowdot(M, N, K, alpha, dA.ptr, M, dB.ptr, N, dC.ptr, M, flags, None, 0, dv.ptr, None, dR2.ptr, 0),
          "dgemm_rowdot (operand dot)")
    return dC.to_numpy().reshape(N, M).T.copy(), dR2.to_numpy()


# ---- high-level API ---------------------------------------------------------------------------
MODEL_EXACT, MODEL_FITC, MODEL_VFE = 0, 1, 2
COV_SE_ARD, COV_SE_ISO = 0, 1
ALLREDUCE_FN = C.CFUNCTYPE(_i, _vp, _i64, _vp)

lib.gpx_ctx_create.argtypes = [_i, C.POINTER(_vp)]
lib.gpx_ctx_destroy.argtypes = [_vp]
lib.gpx_set_data.argtypes = [_vp, _vp, _vp, _i64, _i64]
lib.gpx_set_inducing.argtypes = [_vp, _vp, _i64]
lib.gpx_set_allreduce.argtypes = [_vp, ALLREDUCE_FN, _vp]
ALLREDUCE_STREAM_FN = C.CFUNCTYPE(_i, _vp, _i64, _vp, _vp)
lib.gpx_set_allreduce_stream.argtypes = [_vp, ALLREDUCE_STREAM_FN, _vp]
lib.gpx_dscal_stream.argtypes = [_vp, _i64, _d, _vp]
lib.gpx_nll_grad.argtypes = [_vp, _i, _i, _vp, _i, _d, C.POINTER(_d), _vp]
lib.gpx_prepare.argtypes = [_vp, _i, _i, _vp, C.POINTER(_d)]
lib.gpx_predict.argtypes = [_vp, _vp, _i64, _vp, _vp]
lib.gpx_set_spd_check.argtypes = [_vp, _i]
lib.gpx_last_spd_residual.argtypes = [_vp, C.POINTER(_d)]
lib.gpx_predict_with_grad.argtypes = [_vp, _vp, _i64, _vp, _vp, _vp, _vp]
lib.gpx_cov_k.argtypes = [_i, _i64, _vp, _vp, _i64, _vp, _i64, _vp]
lib.gpx_chol_solve.argtypes = [_i64, _vp, C.POINTER(_i), C.POINTER(_d), _vp, _i64, _vp]


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def num_hyp(cov, d):
    return (d + 1 if cov == COV_SE_ARD else 2) + 2


class Context:
    """One GPU context = one GP object's device state.  X is (n, d) with one point per row
    (C order), which is the reference's d x n column-major layout byte for byte."""

    def __init__(self, device=0):
        h = _vp()
        check(lib.gpx_ctx_create(device, C.byref(h)), "ctx_create")
        self._h = h
        self.d = 0
        self._cb = None

    def set_data(self, X, y):
        X, y = _c(X), _c(y)
        n, self.d = X.shape
        assert y.shape == (n,)
        check(lib.gpx_set_data(self._h, X.ctypes.data, y.ctypes.data, n, self.d), "set_data")

    def set_inducing(self, U):
        U = _c(U)
        assert U.shape[1] == self.d
        check(lib.gpx_set_inducing(self._h, U.ctypes.data, U.shape[0]), "set_inducing")

    def set_allreduce(self, fn):
        """fn(device_ptr:int, count:int) -> None: in-place sum all-reduce of `count` doubles."""
        def tramp(buf, count, user):
            try:
                fn(buf, count)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("gpx allreduce callback failed:", e)
                return 1
        self._cb = ALLREDUCE_FN(tramp)
        check(lib.gpx_set_allreduce(self._h, self._cb, None), "set_allreduce")

    def set_allreduce_stream(self, fn):
        """fn(device_ptr:int, count:int, hip_stream:int) -> None: ENQUEUE the in-place sum all-reduce of
        `count` doubles on that stream and return without waiting."""
        def tramp(buf, count, stream, user):
            try:
                fn(buf, count, stream or 0)
                return 0
            except Exception as e:  # never let an exception cross the C boundary
                print("gpx allreduce callback failed:", e)
                return 1
        self._cb = ALLREDUCE_STREAM_FN(tramp)
        check(lib.gpx_set_allreduce_stream(self._h, self._cb, None), "set_allreduce_stream")

    def nll_grad(self, model, cov, hyp, calc_grad=True, jitter=0.0):
        hyp = _c(hyp)
        assert hyp.shape == (num_hyp(cov, self.d),)
        nll = _d(0.0)
        g = np.zeros_like(hyp)
        check(lib.gpx_nll_grad(self._h, model, cov, hyp.ctypes.data, int(calc_grad), jitter, C.byref(nll),
                               g.ctypes.data if calc_grad else None), "nll_grad")
        return (nll.value, g) if calc_grad else nll.value

    def prepare(self, model, cov, hyp, jitter=0.0):
        hyp = _c(hyp).copy()
        j = _d(jitter)
        check(lib.gpx_prepare(self._h, model, cov, hyp.ctypes.data, C.byref(j)), "prepare")
        return hyp, j.value

    def set_spd_check(self, cols):
        """residual part of _check_SPD (GP.cpp:455-457) inside prepare() of the exact model; 0 = off"""
        check(lib.gpx_set_spd_check(self._h, int(cols)), "set_spd_check")

    def last_spd_residual(self):
        v = _d(0.0)
        check(lib.gpx_last_spd_residual(self._h, C.byref(v)), "last_spd_residual")
        return v.value

    def predict(self, Xs, want_mu=True, want_s2=True):
        Xs = _c(Xs)
        t = Xs.shape[0]
        mu = np.empty(t) if want_mu else None
        s2 = np.empty(t) if want_s2 else None
        check(lib.gpx_predict(self._h, Xs.ctypes.data, t, mu.ctypes.data if want_mu else None,
                              s2.ctypes.data if want_s2 else None), "predict")
        return mu, s2

    def predict_with_grad(self, Xs, want_gs2=True):
        """mu, s2 and their gradients w.r.t. the test inputs, each gradient (t, d) with one test
        point per row (GP.cpp:284-296, 305-313, 324-333; FITC.cpp:118-129; VFE.cpp:118-123)."""
        Xs = _c(Xs)
        t = Xs.shape[0]
        mu, s2 = np.empty(t), np.empty(t)
        gmu = np.empty((t, self.d))
        gs2 = np.empty((t, self.d)) if want_gs2 else None
        check(lib.gpx_predict_with_grad(self._h, Xs.ctypes.data, t, mu.ctypes.data, s2.ctypes.data, gmu.ctypes.data,
                                        gs2.ctypes.data if want_gs2 else None), "predict_with_grad")
        return mu, s2, gmu, gs2

    def close(self):
        if self._h:
            lib.gpx_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def cov_k(cov, hyp, x1, x2):
    x1, x2, hyp = _c(x1), _c(x2), _c(hyp)
    p, d = x1.shape
    q = x2.shape[0]
    K = np.empty((q, p))
    check(lib.gpx_cov_k(cov, d, hyp.ctypes.data, x1.ctypes.data, p, x2.ctypes.data, q, K.ctypes.data), "cov_k")
    return K.T


lib.gpx_cov_dk_dhyp.argtypes = [_i, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]


def cov_dk_dhyp(cov, hyp, idx, x1, x2, K=None):
    """d k(x1, x2) / d hyp[idx] (p, q); K (p, q) as returned by cov_k, or None (computed first)."""
    x1, x2, hyp = _c(x1), _c(x2), _c(hyp)
    p, d = x1.shape
    q = x2.shape[0]
    Kc = None if K is None else np.ascontiguousarray(np.asarray(K, dtype=np.float64).T)   # column-major p x q
    dK = np.empty((q, p))
    check(lib.gpx_cov_dk_dhyp(cov, d, hyp.ctypes.data, idx, x1.ctypes.data, p, x2.ctypes.data, q,
                              Kc.ctypes.data if Kc is not None else None, dK.ctypes.data), "cov_dk_dhyp")
    return dK.T


def chol_solve(A, B=None, want_inverse=False):
    """Returns (spd, logdet, X, Inv) for symmetric A (lower triangle read)."""
    A = np.asarray(A, dtype=np.float64)
    n = A.shape[0]
    Ac = colmajor(A)
    spd, ld = _i(0), _d(0.0)
    Bc = None
    nrhs = 0
    if B is not None:
        B = np.asarray(B, dtype=np.float64).reshape(n, -1)
        nrhs = B.shape[1]
        Bc = colmajor(B).copy()
    Inv = np.empty(n * n) if want_inverse else None
    check(lib.gpx_chol_solve(n, Ac.ctypes.data, C.byref(spd), C.byref(ld), Bc.ctypes.data if nrhs else None, nrhs,
                             Inv.ctypes.data if want_inverse else None), "chol_solve")
    X = Bc.reshape(nrhs, n).T.copy() if nrhs else None
    return bool(spd.value), ld.value, X, (Inv.reshape(n, n).T.copy() if want_inverse else None)
