"""GPU parity tests of the exact-GP path against the CPU oracle (through the C ABI).

Tolerances are the north star's: NLL rel <= 1e-9, gradient rel <= 1e-7 (max-norm relative),
predictions rel <= 1e-9, all fp64.
"""
import os

import numpy as np
import pytest

import oracle
from gp_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "refdata_small.npz")


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope="module")
def gpx():
    from gp_amd import lib
    return lib


@pytest.mark.parametrize("shape", [(5, 3), (64, 64), (65, 130), (200, 77), (1000, 300)])
@pytest.mark.parametrize("cov", [0, 1])
def test_cov_k(gpx, cov, shape):
    p, q = shape
    d = 7
    X1 = synth.make_inputs(3, p, d)
    X2 = synth.make_inputs(4, q, d)
    hyp = np.concatenate([0.3 * np.sin(np.arange(d)) + 0.5, [0.2]]) if cov == 0 else np.array([0.7, 0.2])
    K = gpx.cov_k(cov, hyp, X1, X2)
    Ko = oracle.cov_k(cov, hyp, X1, X2)
    assert rel(K, Ko) < 1e-14
    Kd = gpx.cov_k(cov, hyp, X1, X1)
    assert np.all(np.diag(Kd) == np.exp(2 * hyp[-1]))   # direct differences: exact diagonal


@pytest.mark.parametrize("n", [1, 7, 64, 65, 200, 513, 1500])
def test_chol_solve_inverse_logdet(gpx, n):
    rng = np.random.default_rng(n)
    G = rng.normal(size=(n, n))
    A = G @ G.T / n + np.eye(n)
    B = rng.normal(size=(n, 3))
    spd, ld, X, Inv = gpx.chol_solve(A, B, want_inverse=True)
    assert spd
    sign, ld_ref = np.linalg.slogdet(A)
    assert abs(ld - ld_ref) <= 1e-11 * max(1.0, abs(ld_ref))
    assert rel(X, np.linalg.solve(A, B)) < 1e-11
    assert rel(Inv, np.linalg.inv(A)) < 1e-11
    info, L = oracle.chol(A)
    assert info == 0
    A[n // 2, n // 2] = -1.0
    assert not gpx.chol_solve(A)[0]                         # check_SPD() == false


@pytest.mark.parametrize("cov", [0, 1])
@pytest.mark.parametrize("n,d", [(50, 3), (300, 5), (1000, 10), (1537, 24)])
def test_exact_nll_grad_vs_oracle(gpx, cov, n, d):
    X, y, hyp, Xs = synth.make_problem(1, n, d, 33)
    if cov == 1:
        hyp = np.array([0.5 * np.log(d), 0.1, np.log(0.1), 0.05])
    else:
        hyp = hyp + np.concatenate([np.zeros(d), [0.1, 0.0, 0.05]])
    ctx = gpx.Context()
    ctx.set_data(X, y)
    nll, g = ctx.nll_grad(gpx.MODEL_EXACT, cov, hyp)
    nll_o, g_o = oracle.exact_nll(cov, X, y, hyp)
    assert abs(nll - nll_o) <= 1e-9 * abs(nll_o)
    assert rel(g, g_o) <= 1e-7
    nll_only = ctx.nll_grad(gpx.MODEL_EXACT, cov, hyp, calc_grad=False)   # TRSV path vs explicit-inverse path
    assert abs(nll_only - nll) <= 1e-11 * abs(nll) and abs(nll_only - nll_o) <= 1e-9 * abs(nll_o)
    hyp2, _ = ctx.prepare(gpx.MODEL_EXACT, cov, hyp)
    assert np.array_equal(hyp2, hyp)
    mu, s2 = ctx.predict(Xs)
    mu_o, s2_o = oracle.ExactState(cov, X, y, hyp).predict(Xs)
    assert rel(mu, mu_o) <= 1e-9 and rel(s2, s2_o) <= 1e-9
    # predictive input gradients (GP.cpp:284-296, 305-313, 324-333)
    mu_g, s2_g, gmu, gs2 = ctx.predict_with_grad(Xs)
    _, _, gmu_o, gs2_o = oracle.ExactState(cov, X, y, hyp).predict(Xs, grad=True)
    assert np.array_equal(mu_g, mu) and rel(s2_g, s2_o) <= 1e-9
    assert rel(gmu, gmu_o) <= 1e-9 and rel(gs2, gs2_o) <= 1e-8
    ctx.close()


@pytest.mark.parametrize("n,d", [(4230, 7)])
def test_exact_nll_grad_vs_oracle_mid_size(gpx, n, d):
    """The literal oracle (the same C restatement built with OpenMP, ~3 s on the box's cores) at a size whose recursion
    reaches the 2048-row level: 128 x 128-tile products with triangular K ranges, the paired launches of a node, ragged
    last blocks (n = 33 * 128 + 6) - the sizes between the n <= 1537 oracle tests and the property tests at 8192 / 65536."""
    X, y, hyp, _ = synth.make_problem(5, n, d, 4)
    hyp = hyp + np.concatenate([np.zeros(d), [0.05, 0.0, -0.02]])
    ctx = gpx.Context()
    ctx.set_data(X, y)
    nll, g = ctx.nll_grad(gpx.MODEL_EXACT, 0, hyp)
    nll_only = ctx.nll_grad(gpx.MODEL_EXACT, 0, hyp, calc_grad=False)
    ctx.close()
    nll_o, g_o = oracle.exact_nll_all_cores(0, X, y, hyp)
    print(f"\n[exact n={n} d={d}] NLL rel {abs(nll - nll_o) / abs(nll_o):.2e}, gradient rel {rel(g, g_o):.2e}, "
          f"NLL-only path rel {abs(nll_only - nll_o) / abs(nll_o):.2e}")
    assert abs(nll - nll_o) <= 1e-9 * abs(nll_o)
    assert rel(g, g_o) <= 1e-7
    assert abs(nll_only - nll_o) <= 1e-9 * abs(nll_o)


def test_exact_predict_grad_chunked_and_finite_difference(gpx):
    """more test points than one predict chunk, and the gradients against central differences"""
    n, d, t = 400, 4, 1100
    X, y, hyp, Xs = synth.make_problem(1, n, d, t)
    ctx = gpx.Context()
    ctx.set_data(X, y)
    ctx.prepare(gpx.MODEL_EXACT, 0, hyp)
    mu, s2, gmu, gs2 = ctx.predict_with_grad(Xs)
    mu_o, s2_o, gmu_o, gs2_o = oracle.ExactState(0, X, y, hyp).predict(Xs, grad=True)
    assert rel(mu, mu_o) <= 1e-9 and rel(s2, s2_o) <= 1e-9
    assert rel(gmu, gmu_o) <= 1e-9 and rel(gs2, gs2_o) <= 1e-8
    h = 1e-5
    for k in range(d):
        E = np.zeros_like(Xs)
        E[:, k] = h
        mp, sp = ctx.predict(Xs + E)
        mm, sm = ctx.predict(Xs - E)
        assert np.max(np.abs((mp - mm) / (2 * h) - gmu[:, k])) <= 1e-6 * max(1.0, np.max(np.abs(gmu)))
        assert np.max(np.abs((sp - sm) / (2 * h) - gs2[:, k])) <= 1e-6 * max(1.0, np.max(np.abs(gs2)))
    ctx.close()


def test_exact_golden_fixture(gpx):
    g = np.load(GOLD)
    ctx = gpx.Context()
    ctx.set_data(g["X"], g["y"])
    for cov, c in ((0, "ard"), (1, "iso")):
        hyp = g[f"{c}_hyp"]
        nll, gr = ctx.nll_grad(gpx.MODEL_EXACT, cov, hyp)
        assert abs(nll - g[f"{c}_exact_nll"]) <= 1e-9 * abs(nll)
        assert rel(gr, g[f"{c}_exact_grad"]) <= 1e-7
        ctx.prepare(gpx.MODEL_EXACT, cov, hyp)
        mu, s2 = ctx.predict(g["Xs"])
        assert rel(mu, g[f"{c}_exact_mu"]) <= 1e-9 and rel(s2, g[f"{c}_exact_s2"]) <= 1e-9
    ctx.close()


def test_exact_failure_and_noise_escalation(gpx):
    X, y, hyp, _ = synth.make_problem(2, 40, 3)
    Xd, yd = np.vstack([X, X]), np.concatenate([y, y])
    h = hyp.copy()
    h[-2] = -np.inf                                   # zero noise on duplicated points: singular
    ctx = gpx.Context()
    ctx.set_data(Xd, yd)
    nll, g = ctx.nll_grad(gpx.MODEL_EXACT, 0, h)
    assert nll == np.inf and np.all(np.isinf(g))      # GP.cpp:132-134
    h[-2] = np.log(1e-12)
    h2, _ = ctx.prepare(gpx.MODEL_EXACT, 0, h)
    st = oracle.ExactState(0, Xd, yd, h)
    k = (h2[-2] - h[-2]) / np.log(np.sqrt(10))
    assert h2[-2] > h[-2] and abs(k - round(k)) < 1e-9   # GP.cpp:431-440
    assert abs(h2[-2] - st.hyp[-2]) < 1e-12 + 2 * np.log(np.sqrt(10))  # same loop; pivots may differ by one step
    with pytest.raises(gpx.GpxError):
        gpx.Context().predict(X)                      # predict before _setK (assert(_trained), GP.cpp:275)
    ctx.close()


def test_predict_shards_over_test_points(gpx):
    """SURVEY.md §8(e) predict: test points are independent, so ranks shard X* with no collective — every shard of
    the test set, predicted by its own context (one per rank), must reproduce the rows of the single-context answer
    bit for bit (gp_amd.sharding.shard_range is bench.py's partition)."""
    from gp_amd.sharding import shard_range
    X, y, hyp, Xs = synth.make_problem(4, 900, 7, 301)
    ref = gpx.Context()
    ref.set_data(X, y)
    ref.prepare(gpx.MODEL_EXACT, 0, hyp)
    mu, s2, gmu, gs2 = ref.predict_with_grad(Xs)
    for world in (2, 3):
        for rank in range(world):
            r0, r1 = shard_range(len(Xs), rank, world)
            ctx = gpx.Context()
            ctx.set_data(X, y)
            ctx.prepare(gpx.MODEL_EXACT, 0, hyp)
            m_r, s_r, gm_r, gs_r = ctx.predict_with_grad(Xs[r0:r1])
            assert np.array_equal(m_r, mu[r0:r1]) and np.array_equal(s_r, s2[r0:r1])
            assert np.array_equal(gm_r, gmu[r0:r1]) and np.array_equal(gs_r, gs2[r0:r1])
            ctx.close()
    ref.close()


@pytest.mark.parametrize("n,d,ls,lsn", [(200, 2, 1.0, -14.0), (200, 2, 1.5, -16.0), (300, 3, 1.2, -15.0), (150, 1, 0.5, -5.0)])
def test_setk_residual_spd_check_escalates_like_the_literal_check(gpx, n, d, ls, lsn):
    """_check_SPD's residual test (GP.cpp:455-457: mean|I - K K^-1| < 1e-4) as the optional flag of gpx_prepare: with
    nearly singular K + sn2 I every Cholesky pivot is still positive, so the default policy accepts the start noise,
    while the residual test escalates it (x sqrt(10) per round, GP.cpp:431-440) exactly as far as the oracle's literal
    _check_SPD does — with all n columns and with 64 sampled ones.  (inv_err within a factor ~2 of the 1e-4 threshold is
    decided by rounding: e.g. n=150 d=1 l=0.5 log sn=-13 is accepted by the oracle and escalated once here; the cases
    below are decided the same way on MI355X, measured inv_err at the accepted noise 3.6e-5 .. 5.3e-5; the last one
    is well conditioned: no escalation on either side.)"""
    X, y, hyp, Xs = synth.make_problem(8, n, d, 5)
    hyp[:d] = ls
    hyp[d + 1] = lsn
    lit = oracle.ExactState(0, X, y, hyp, literal_spd_check=True)
    piv = oracle.ExactState(0, X, y, hyp, literal_spd_check=False)
    ctx = gpx.Context()
    ctx.set_data(X, y)
    h0, _ = ctx.prepare(gpx.MODEL_EXACT, 0, hyp)
    # default = Cholesky success only: stops earlier than the literal check (at log sn = -16 the matrix is singular to
    # rounding and whether the last pivot is > 0 depends on the summation order: no equality with the oracle there)
    assert h0[d + 1] < lit.hyp[d + 1] or lit.hyp[d + 1] == lsn
    assert lsn > -15.5 and np.array_equal(h0, piv.hyp) or lsn <= -15.5
    for cols in (n, 64):
        ctx.set_spd_check(cols)
        h1, _ = ctx.prepare(gpx.MODEL_EXACT, 0, hyp)
        print(f"\n[spd check n={n} cols={cols}] log sn {lsn} -> {h1[d + 1]:.6f} (oracle literal {lit.hyp[d + 1]:.6f}), "
              f"last inv_err {ctx.last_spd_residual():.3e}")
        assert ctx.last_spd_residual() < 1e-4
        assert np.allclose(h1, lit.hyp, rtol=0, atol=1e-12)
        mu, s2 = ctx.predict(Xs)
        mu_o, s2_o = lit.predict(Xs)
        # cond(K + sn2 I) ~ 1e13 at the accepted noise: two fp64 solves agree to a few digits only (measured 1e-3)
        assert np.all(np.isfinite(mu)) and np.max(np.abs(mu - mu_o)) <= 2e-2 * np.max(np.abs(mu_o))
    ctx.close()


@pytest.mark.parametrize("cov", [0, 1])
def test_cov_dk_dhyp_on_the_device_matches_the_oracle(gpx, cov):
    """Cov::dk_dhyp (Cov/Cov.h:15-16; CovSEard.cpp:12-27, CovSEiso.cpp:12-31) through gpx_cov_dk_dhyp: every kernel
    hyper, with the caller's K (5-argument overload) and without (4-argument overload: K computed first)."""
    rng = np.random.default_rng(4)
    d, p, q = 7, 131, 67
    x1, x2 = rng.normal(size=(p, d)), rng.normal(size=(q, d))
    hyp = np.concatenate([0.3 * rng.normal(size=d) + 0.5, [0.2]]) if cov == 0 else np.array([0.7, 0.2])
    K = oracle.cov_k(cov, hyp, x1, x2)
    for idx in range(len(hyp)):
        want = oracle.cov_dk_dhyp(cov, hyp, idx, x1, x2, K)
        got = gpx.cov_dk_dhyp(cov, hyp, idx, x1, x2, K)
        got4 = gpx.cov_dk_dhyp(cov, hyp, idx, x1, x2)
        assert rel(got, want) <= 1e-15, (cov, idx)
        assert rel(got4, want) <= 1e-14, (cov, idx)
    assert gpx.lib.gpx_cov_dk_dhyp(cov, d, hyp.ctypes.data, len(hyp), x1.ctypes.data, p, x2.ctypes.data, q, None,
                                   np.empty((q, p)).ctypes.data) == 1     # "Wrong index for dk_dhyp"
