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


def dgemm(A, B, C_in=None, a_kc=False, b_kc=False, alpha=1.0, beta=0.0, flags=0, splitk_work=0):
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
    if splitk_work:
        w = DeviceBuffer(splitk_work)
        check(lib.gpx_dgemm_splitk(int(a_kc), int(b_kc), M, N, K, alpha, dA.ptr, lda, dB.ptr, ldb, beta,
                                   dC.ptr, M, flags, w.ptr, splitk_work), "dgemm_splitk")
    else:
        check(lib.gpx_dgemm(int(a_kc), int(b_kc), M, N, K, alpha, dA.ptr, lda, dB.ptr, ldb, beta, dC.ptr, M,
                            flags), "dgemm")
    return dC.to_numpy().reshape(N, M).T.copy()
