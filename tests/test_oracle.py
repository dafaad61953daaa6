"""CPU tests of the oracle: literal restatement (oracle/gp_oracle.c) vs the independent NumPy
closed forms (tests/np_gp.py) vs finite differences (GP.cpp:486-507), and vs the committed
golden fixture built from the reference's sample data (tools/gen_golden.py).

PARITY UNPINNED: the reference holds no expected outputs for this path and cannot be built here.
"""
import os

import numpy as np
import pytest

import np_gp
import oracle
from gp_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "refdata_small.npz")


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))


def problem(cov, n=200, d=4, m=24, t=16):
    X, y, hyp, Xs = synth.make_problem(11, n, d, t)
    if cov == oracle.COV_SE_ISO:
        hyp = np.array([np.log(2.0), 0.1, np.log(0.1), 0.05])
    else:
        hyp = hyp + np.array([0.0] * d + [0.1, 0.0, 0.05])
    return X, y, hyp, Xs, X[::n // m][:m].copy()


@pytest.mark.parametrize("cov", [oracle.COV_SE_ARD, oracle.COV_SE_ISO])
def test_cov_k_and_derivatives(cov):
    X, y, hyp, Xs, U = problem(cov)
    K = oracle.cov_k(cov, hyp, X, U)
    assert rel(K, np_gp.cov_k(cov, hyp, X, U)) < 1e-14
    nk = oracle.num_hyp(cov, X.shape[1]) - 2
    for i in range(nk):
        dK = oracle.cov_dk_dhyp(cov, hyp, i, X, U, K)
        hp, hm = hyp.copy(), hyp.copy()
        hp[i] += 1e-6
        hm[i] -= 1e-6
        fd = (oracle.cov_k(cov, hp, X, U) - oracle.cov_k(cov, hm, X, U)) / 2e-6
        assert rel(dK, fd) < 1e-8


@pytest.mark.parametrize("cov", [oracle.COV_SE_ARD, oracle.COV_SE_ISO])
def test_exact_three_way(cov):
    X, y, hyp, Xs, U = problem(cov)
    nll, g = oracle.exact_nll(cov, X, y, hyp)
    nll2, g2 = np_gp.exact_nll(cov, X, y, hyp)
    assert abs(nll - nll2) <= 1e-12 * abs(nll2)
    assert rel(g, g2) < 1e-10
    fd = np_gp.finite_diff_grad(lambda h: oracle.exact_nll(cov, X, y, h, False), hyp, 1e-5)
    assert rel(fd, g) < 1e-6
    assert oracle.exact_nll(cov, X, y, hyp, False) == nll
    st = oracle.ExactState(cov, X, y, hyp)
    mu, s2 = st.predict(Xs)
    mu2, s22 = np_gp.exact_predict(cov, X, y, hyp, Xs)
    assert rel(mu, mu2) < 1e-11 and rel(s2, s22) < 1e-11


@pytest.mark.parametrize("cov", [oracle.COV_SE_ARD, oracle.COV_SE_ISO])
@pytest.mark.parametrize("model", ["fitc", "vfe"])
def test_sparse_three_way(cov, model):
    X, y, hyp, Xs, U = problem(cov)
    jit = 1e-6
    fo = oracle.fitc_nll if model == "fitc" else oracle.vfe_nll
    fn = np_gp.fitc_nll if model == "fitc" else np_gp.vfe_nll
    nll, g = fo(cov, X, y, U, jit, hyp)
    nll2, g2 = fn(cov, X, y, U, jit, hyp)
    assert abs(nll - nll2) <= 1e-9 * abs(nll2)
    assert rel(g, g2) < 1e-7
    fd = np_gp.finite_diff_grad(lambda h: fo(cov, X, y, U, jit, h, False), hyp, 1e-4)
    # dKuu is taken of the jittered Kuu (FITC.cpp:271 / VFE.cpp:215): the sf entry carries an
    # extra O(jitter) term that finite differences of the NLL do not see
    assert rel(fd, g) < 1e-4


def test_exact_literal_setk_check_and_noise_escalation():
    X, y, hyp, Xs, U = problem(oracle.COV_SE_ARD, n=60)
    st = oracle.ExactState(oracle.COV_SE_ARD, X, y, hyp, literal_spd_check=True)
    assert st.hyp[-2] == hyp[-2]          # well-conditioned: no escalation (GP.cpp:431-440)
    Xd = np.vstack([X, X[:5]])            # duplicate points + tiny noise -> not SPD numerically
    yd = np.concatenate([y, y[:5]])
    h = hyp.copy()
    h[-2] = np.log(1e-12)
    st = oracle.ExactState(oracle.COV_SE_ARD, Xd, yd, h)
    assert st.hyp[-2] > h[-2]
    k = (st.hyp[-2] - h[-2]) / np.log(np.sqrt(10))
    assert abs(k - round(k)) < 1e-9       # raised by whole multiples of log(sqrt(10))


def test_failure_conventions():
    X, y, hyp, Xs, U = problem(oracle.COV_SE_ARD, n=40)
    Xd = np.vstack([X, X])                # exactly singular K, zero noise
    yd = np.concatenate([y, y])
    h = hyp.copy()
    h[-2] = -np.inf
    nll, g = oracle.exact_nll(oracle.COV_SE_ARD, Xd, yd, h)
    assert nll == np.inf and np.all(np.isinf(g))          # GP.cpp:132-134
    Ud = np.vstack([U, U])
    nll, g = oracle.fitc_nll(oracle.COV_SE_ARD, X, y, Ud, 0.0, hyp)
    assert nll == np.inf and np.all(g == 0)               # FITC.cpp:228-233
    nll, g = oracle.vfe_nll(oracle.COV_SE_ARD, X, y, Ud, 0.0, hyp)
    assert nll == np.inf and np.all(g == 0)               # VFE.cpp:190-195


def test_default_hyps_and_bounds():
    X, y, hyp, Xs, U = problem(oracle.COV_SE_ARD)
    d = X.shape[1]
    h = oracle.default_hyps(oracle.COV_SE_ARD, X, y)
    assert np.allclose(h[:d], np.log(X.std(axis=0, ddof=1)), rtol=0, atol=1e-14)
    assert np.isclose(h[d], np.log(y.std(ddof=1)))
    assert np.isclose(h[d + 1], max(np.log(1e-3), np.log(y.std(ddof=1) * 1e-3)))
    assert np.isclose(h[d + 2], y.mean())
    lb, ub = oracle.hyp_range(oracle.COV_SE_ARD, X, y)
    assert np.all(lb < ub) and np.isclose(lb[-2], np.log(1e-3)) and np.isclose(lb[-1], y.min())


def test_golden_fixture():
    g = np.load(GOLD)
    X, y, Xs, U = g["X"], g["y"], g["Xs"], g["U"]
    for cov, c in ((oracle.COV_SE_ARD, "ard"), (oracle.COV_SE_ISO, "iso")):
        assert np.array_equal(oracle.default_hyps(cov, X, y), g[f"{c}_default_hyp"])
        hyp = g[f"{c}_hyp"]
        nll, gr = oracle.exact_nll(cov, X, y, hyp)
        assert abs(nll - g[f"{c}_exact_nll"]) <= 1e-13 * abs(nll) and rel(gr, g[f"{c}_exact_grad"]) < 1e-12
        nll2, gr2 = np_gp.exact_nll(cov, X, y, hyp)
        assert abs(nll2 - g[f"{c}_exact_nll"]) <= 1e-11 * abs(nll) and rel(gr2, g[f"{c}_exact_grad"]) < 1e-9
        mu, s2 = oracle.ExactState(cov, X, y, hyp).predict(Xs)
        assert rel(mu, g[f"{c}_exact_mu"]) < 1e-13 and rel(s2, g[f"{c}_exact_s2"]) < 1e-13
        jit = float(g[f"{c}_jitter"])
        for model, fn in (("fitc", np_gp.fitc_nll), ("vfe", np_gp.vfe_nll)):
            nll2, gr2 = fn(cov, X, y, U, jit, hyp)
            assert abs(nll2 - g[f"{c}_{model}_nll"]) <= 1e-10 * abs(nll2)
            assert rel(gr2, g[f"{c}_{model}_grad"]) < 1e-7
