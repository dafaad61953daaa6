"""gpx_dgemm_splitk_noalloc (the entry point that allocates nothing: split-K slabs in a static device
array of the library) against gpx_dgemm_splitk with a caller-provided workspace of the same capacity.
fp64 sums in the same order: the bar is bit-exact.  Two calls in a row, the second with a smaller input,
so that whatever the first call left in the shared static array must not leak into the second."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WORK_DOUBLES = 8 * 1024 * 1024      # GPX_NOALLOC_WORK_DOUBLES (include/gpx.h)
MAX_MN = 1024 * 1024                # GPX_NOALLOC_MAX_MN


@pytest.fixture(scope="module")
def gpx():
    from gp_amd import lib
    lib.lib.gpx_dgemm_splitk_noalloc.argtypes = lib.lib.gpx_dgemm.argtypes + [C.c_void_p]
    return lib


def _gram(gpx, n, m, flags, noalloc, seed, beta=0.0):
    """C = Kux Kxu (+ beta*C0) for a K-contiguous n x m operand, as sparse.hip calls it (FITC.cpp:217, VFE.cpp:173)"""
    rng = np.random.default_rng(seed)
    Kxu = rng.normal(size=(m, n))                       # device: column stride n == K-contiguous (k + m*lda)
    C0 = rng.normal(size=(m, m))
    dA = gpx.DeviceBuffer.from_numpy(Kxu)
    dC = gpx.DeviceBuffer.from_numpy(gpx.colmajor(C0))
    if noalloc:
        gpx.check(gpx.lib.gpx_dgemm_splitk_noalloc(1, 1, m, m, n, 1.0, dA.ptr, n, dA.ptr, n, beta, dC.ptr, m, flags, None), "noalloc")
    else:
        w = gpx.DeviceBuffer(WORK_DOUBLES)
        w_fill = np.full(1 << 16, np.nan)               # a caller's workspace holds garbage too
        gpx.check(gpx.lib.gpx_memcpy_h2d(w.ptr, w_fill.ctypes.data, w_fill.nbytes), "h2d")
        gpx.check(gpx.lib.gpx_dgemm_splitk(1, 1, m, m, n, 1.0, dA.ptr, n, dA.ptr, n, beta, dC.ptr, m, flags, w.ptr,
                                           WORK_DOUBLES), "splitk")
        w.free()
    gpx.check(gpx.lib.gpx_device_synchronize(), "sync")
    out = dC.to_numpy().reshape(m, m).T.copy()
    dA.free()
    dC.free()
    return out, Kxu, C0


def test_noalloc_matches_workspace_variant_twice_second_smaller(gpx):
    # first call: the m = 1024 Gram product at the size limit (lower tiles only, as the library runs it);
    # second call, right after it on the same GPU: smaller and ragged in every dimension, full output, beta != 0
    for n, m, flags, beta in ((40000, 1024, gpx.GEMM_LOWER, 0.0), (9001, 333, 0, 0.5)):
        got, Kxu, C0 = _gram(gpx, n, m, flags, True, seed=n, beta=beta)
        ref, _, _ = _gram(gpx, n, m, flags, False, seed=n, beta=beta)
        # GEMM_LOWER: tiles above the diagonal are never computed, compare what the product defines
        keep = np.tril(np.ones((m, m), bool)) if flags & gpx.GEMM_LOWER else np.ones((m, m), bool)
        assert np.array_equal(got[keep], ref[keep])      # same kernels, same slab count, same order: bit-exact
        want = Kxu @ Kxu.T + beta * C0
        assert np.max(np.abs(got - want)[keep]) <= 1e-11 * np.max(np.abs(want))


def test_noalloc_rejects_sizes_beyond_the_static_workspace(gpx):
    m = 1056                                             # m*m > GPX_NOALLOC_MAX_MN
    assert m * m > MAX_MN
    d = gpx.DeviceBuffer(m * 64)
    c = gpx.DeviceBuffer(m * m)
    assert gpx.lib.gpx_dgemm_splitk_noalloc(1, 1, m, m, 64, 1.0, d.ptr, 64, d.ptr, 64, 0.0, c.ptr, m, 0, None) == 1
    # sizes whose product overflows int64 are refused by the same test, the product is never formed
    assert gpx.lib.gpx_dgemm_splitk_noalloc(1, 1, 1 << 40, 1 << 40, 64, 1.0, d.ptr, 64, d.ptr, 64, 0.0, c.ptr, 1 << 40, 0, None) == 1
    d.free()
    c.free()
