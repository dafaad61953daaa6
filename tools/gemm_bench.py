#!/usr/bin/env python3
"""Micro-benchmark of the fp64 MFMA GEMM: TFLOP/s for the operand layouts / shapes the GP path uses."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_amd import lib  # noqa: E402


def bench(M, N, K, a_kc, b_kc, flags=0, reps=3, name=""):
    rng = np.random.default_rng(0)
    ld = max(M, N, K) + 16
    base = rng.uniform(-1, 1, 1 << 20)
    A = lib.DeviceBuffer.from_numpy(np.resize(base, ld * max(M, K)))
    B = lib.DeviceBuffer.from_numpy(np.resize(base[::-1], ld * max(N, K)))
    Cd = lib.DeviceBuffer(ld * N)
    Cd.zero()

    def run():
        lib.check(lib.lib.gpx_dgemm(a_kc, b_kc, M, N, K, 1.0, A.ptr, ld, B.ptr, ld, 0.0, Cd.ptr, ld, flags), "dgemm")
    run()
    lib.check(lib.lib.gpx_device_synchronize(), "sync")
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    lib.check(lib.lib.gpx_device_synchronize(), "sync")
    dt = (time.perf_counter() - t0) / reps
    fl = 2.0 * M * N * K
    if flags & lib.GEMM_LOWER:
        fl *= 0.5
    if flags & (lib.GEMM_KLO_M | lib.GEMM_KLO_N | lib.GEMM_KHI_M | lib.GEMM_KHI_N):
        fl *= (2.0 / 3.0) if flags & lib.GEMM_LOWER else 0.5   # triangular operand: n^3/3 resp. half
    print(f"{name:28s} M={M:6d} N={N:6d} K={K:6d} akc={a_kc} bkc={b_kc} flags={flags:3d}  {dt * 1e3:9.2f} ms  {fl / dt / 1e12:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    if len(sys.argv) > 2:   # single case for PMC passes
        bench(n, n, n, 0, 0, name="NT", reps=2)
        sys.exit(0)
    if len(sys.argv) <= 1 or True:
        # the five GEMM forms of one exact-GP evaluation (chol.hip potrf_inv_rec + lauum), top level
        h = n // 2
        bench(h, h, h, 0, 0, lib.GEMM_KHI_N, name="L21 = A21 W11^T (KHI_N)")
        bench(h, h, h, 0, 0, lib.GEMM_LOWER, name="A22 -= L21 L21^T (lower)")
        bench(h, h, h, 0, 1, lib.GEMM_KLO_N, name="T = L21 W11 (KLO_N)")
        bench(h, h, h, 0, 1, lib.GEMM_KHI_M | lib.GEMM_HEAVY_LAST, name="W21 = -W22 T (KHI_M)")
        bench(n, n, n, 1, 1, lib.GEMM_LOWER | lib.GEMM_KLO_M, name="W^T W (lower, KLO_M)")
    bench(n, n, n, 0, 0, name="NT (potrf trsm/syrk form)")
    bench(n, n, n, 0, 0, lib.GEMM_LOWER, name="NT lower (syrk)")
    bench(n, n, n, 0, 1, name="NN (trtri)")
    bench(n, n, n, 1, 1, name="TT (lauum form, full)")
    bench(n, 1024, n, 0, 0, name="skinny N=1024")
    bench(min(8 * n, 1 << 20), 1024, 1024, 0, 0, name="tall FITC C=K*Ainv")
