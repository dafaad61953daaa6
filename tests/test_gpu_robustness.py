"""Leak / lifetime guards of the context machinery (streams, events, pinned memory, device buffers): many contexts
created, used and destroyed in one process must leave the process able to evaluate — round 2 recorded a hang after
about 90 LEAKED side streams (a per-caller pool that was never released), so the count here is
well above that."""
import ctypes as C

import numpy as np
import pytest

from gp_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpx():
    from gp_amd import lib
    return lib


def test_300_contexts_created_used_destroyed_then_one_more_evaluation(gpx):
    X, y, hyp, Xs = synth.make_problem(6, 340, 3, 5)
    U, X, y = X[:40], X[40:], y[40:]

    def one():
        ctx = gpx.Context()
        ctx.set_data(X, y)
        ctx.set_inducing(U)
        out = ctx.nll_grad(gpx.MODEL_FITC, 0, hyp, jitter=1e-6)   # both streams, both events, the side-stream chains
        ctx.close()
        return out

    nll0, g0 = one()
    for i in range(300):
        nll, g = one()
        assert nll == nll0 and np.array_equal(g, g0), i            # deterministic: bit-identical every time
    # and the other entry points still work afterwards
    ctx = gpx.Context()
    ctx.set_data(X, y)
    nll_e, g_e = ctx.nll_grad(gpx.MODEL_EXACT, 0, hyp)
    ctx.prepare(gpx.MODEL_EXACT, 0, hyp)
    mu, s2 = ctx.predict(Xs)
    ctx.close()
    assert np.isfinite(nll_e) and np.all(np.isfinite(g_e)) and np.all(np.isfinite(mu)) and np.all(s2 > 0)


def test_context_on_a_missing_device_is_refused_and_leaves_nothing_behind(gpx):
    h = C.c_void_p()
    assert gpx.lib.gpx_ctx_create(gpx.device_count() + 3, C.byref(h)) == 1 and not h.value
    assert gpx.lib.gpx_ctx_create(-1, C.byref(h)) == 1 and not h.value


def test_chol_handle_names_its_device_and_restores_the_callers(gpx):
    """gpx_chol_factor(device, ...): the MatrixSolverLLT handle belongs to the device it was given (-1 = current), a
    missing device is refused, and solve / inverse reuse the handle's staging buffers call after call."""
    lib = gpx.lib
    lib.gpx_chol_factor.argtypes = [C.c_int, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    lib.gpx_chol_solve_factored.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.gpx_chol_inverse_factored.argtypes = [C.c_void_p, C.c_void_p]
    lib.gpx_chol_free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(1)
    n = 203
    Q = rng.normal(size=(n, n))
    A = Q @ Q.T + n * np.eye(n)
    Ac = gpx.colmajor(A)
    h, spd, ld = C.c_void_p(), C.c_int(0), C.c_double(0)
    assert lib.gpx_chol_factor(gpx.device_count() + 2, n, Ac.ctypes.data, C.byref(h), C.byref(spd), C.byref(ld)) == 1 and not h.value
    for dev in (-1, 0):
        assert lib.gpx_chol_factor(dev, n, Ac.ctypes.data, C.byref(h), C.byref(spd), C.byref(ld)) == 0 and spd.value == 1
        assert abs(ld.value - np.linalg.slogdet(A)[1]) <= 1e-12 * abs(ld.value)
        for nrhs in (1, 7, 3):                                   # the staging buffer grows once, then is reused
            B = rng.normal(size=(n, nrhs))
            Bc = gpx.colmajor(B).copy()
            assert lib.gpx_chol_solve_factored(h, Bc.ctypes.data, nrhs) == 0
            Xs = Bc.reshape(nrhs, n).T
            assert np.max(np.abs(A @ Xs - B)) <= 1e-10 * np.max(np.abs(B))
        Inv = np.empty(n * n)
        for _ in range(2):
            assert lib.gpx_chol_inverse_factored(h, Inv.ctypes.data) == 0
            Im = Inv.reshape(n, n)
            assert np.array_equal(Im, Im.T)                      # mirrored on the device
            assert np.max(np.abs(Im @ A - np.eye(n))) <= 1e-10
        assert lib.gpx_chol_free(h) == 0
