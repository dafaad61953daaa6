#!/usr/bin/env python3
"""Mid-size products of the recursion (4096 and 2048 rows): tile size and tile order A/B.
usage: python tools/probe/gemm_mid_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gp_amd import lib  # noqa: E402

SQ, T64, T128, NOT32 = 2048, 64, 128, 8192


def bench(M, a_kc, b_kc, flags, name, reps=40):
    rng = np.random.default_rng(0)
    ld = M + 16
    base = rng.uniform(-1, 1, 1 << 20)
    A = lib.DeviceBuffer.from_numpy(np.resize(base, ld * M))
    B = lib.DeviceBuffer.from_numpy(np.resize(base[::-1], ld * M))
    Cd = lib.DeviceBuffer(ld * M)
    Cd.zero()

    def run():
        lib.check(lib.lib.gpx_dgemm(a_kc, b_kc, M, M, M, 1.0, A.ptr, ld, B.ptr, ld, 0.0, Cd.ptr, ld, flags), "dgemm")
    for _ in range(3):
        run()
    lib.check(lib.lib.gpx_device_synchronize(), "sync")
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    lib.check(lib.lib.gpx_device_synchronize(), "sync")
    dt = (time.perf_counter() - t0) / reps
    fl = 2.0 * M ** 3 * 0.5
    print(f"{name:34s} M={M:5d} flags={flags:6d} {dt * 1e6:9.1f} us {fl / dt / 1e12:6.1f} TF/s", flush=True)


forms = [(0, 0, lib.GEMM_KHI_N, "L21 = A21 W11^T (KHI_N)"), (0, 0, lib.GEMM_LOWER, "A22 -= L21 L21^T (lower)"),
         (0, 1, lib.GEMM_KLO_N, "T = L21 W11 (KLO_N)"), (0, 1, lib.GEMM_KHI_M | lib.GEMM_HEAVY_LAST, "W21 = -W22 T (KHI_M)")]
for M in (4096, 2048):
    for a, b, f, nm in forms:
        bench(M, a, b, f, nm)
        if M == 4096:
            bench(M, a, b, f | SQ, nm + " +squares")
        else:
            bench(M, a, b, f | T64, nm + " 64-tiles")
            bench(M, a, b, f | T128, nm + " 128-tiles")
            bench(M, a, b, f | T128 | SQ, nm + " 128 +squares")
