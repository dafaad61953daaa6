"""Probe: the split-K Gram product S = Kux [G] Kxu (m x m lower tiles, K = n training rows, both operands K-contiguous)
with the n x m operand stored plain column-major (column stride n * 8 bytes) against the row-panel layout (column stride
pan * 8 bytes).  Checks the panelled result against the plain one and prints both times.
usage: python tools/probe/gram_panel_probe.py [n] [m]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from gp_amd import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
L = lib.lib
_vp, _i64, _i, _d = C.c_void_p, C.c_int64, C.c_int, C.c_double
L.gpx_dgemm_splitk_panelled.argtypes = [_i64, _i64, _i64, _d, _vp, _vp, _i64, _i64, _d, _vp, _i64, _i, _vp, _i64, _vp]
L.gpx_dgemm_splitk_panel_rows.argtypes = [_i64, _i64, _i64, _i, _i64]
L.gpx_dgemm_splitk_panel_rows.restype = _i64
work_doubles = 64 * m * m
pan = L.gpx_dgemm_splitk_panel_rows(m, m, n, lib.GEMM_LOWER, work_doubles)
npan = (n + pan - 1) // pan
pst = pan * m
print(f"n {n} m {m}: panels of {pan} rows x {npan}, column stride {pan * 8 / 1024:.0f} KB against {n * 8 / 1024:.0f} KB")
rng = np.random.default_rng(0)
K = rng.normal(size=(m, n))                      # row b = column b of Kxu: plain K-contiguous layout, lda = n
g = rng.uniform(0.5, 1.5, size=n)
dA = lib.DeviceBuffer.from_numpy(K)
P = np.zeros((npan, m, pan))                     # panelled: [panel][column b][row in panel]
for p in range(npan):
    r0, r1 = p * pan, min(n, (p + 1) * pan)
    P[p, :, :r1 - r0] = K[:, r0:r1]
dP = lib.DeviceBuffer.from_numpy(P)
del P
dg = lib.DeviceBuffer.from_numpy(g)
w = lib.DeviceBuffer(work_doubles)
c0, c1 = lib.DeviceBuffer(m * m), lib.DeviceBuffer(m * m)
c0.zero(); c1.zero()


def plain(scale):
    if scale:
        return L.gpx_dgemm_kscale(1, m, m, n, 1.0, dA.ptr, n, dg.ptr, dA.ptr, n, 0.0, c0.ptr, m, lib.GEMM_LOWER, w.ptr, work_doubles)
    return L.gpx_dgemm_splitk(1, 1, m, m, n, 1.0, dA.ptr, n, dA.ptr, n, 0.0, c0.ptr, m, lib.GEMM_LOWER, w.ptr, work_doubles)


def panelled(scale):
    return L.gpx_dgemm_splitk_panelled(m, m, n, 1.0, dP.ptr, dP.ptr, pan, pst, 0.0, c1.ptr, m, lib.GEMM_LOWER, w.ptr, work_doubles,
                                        dg.ptr if scale else None)


for scale in (0, 1):
    for name, fn in (("plain", plain), ("panelled", panelled)):
        lib.check(fn(scale), name)
        lib.check(L.gpx_device_synchronize(), "sync")
        t0 = time.perf_counter()
        for _ in range(5):
            lib.check(fn(scale), name)
        lib.check(L.gpx_device_synchronize(), "sync")
        dt = (time.perf_counter() - t0) / 5
        print(f"kscale {scale} {name:9s} {1e3 * dt:8.3f} ms  {m * m * n / dt / 1e12:6.1f} TFLOP/s useful (lower half)")
    a, b = c0.to_numpy().reshape(m, m).T, c1.to_numpy().reshape(m, m).T
    low = np.tril(np.ones((m, m), bool))
    print(f"kscale {scale}: max |panelled - plain| / max|plain| on the lower triangle = {np.max(np.abs(a - b)[low]) / np.max(np.abs(a[low])):.2e}")
