"""Parity at bench.py's OWN conditions for the sparse models (bench.py run_sparse): synth.eval_hyps,
jitter 1e-8 = (0.1 * noise_lb)^2, inducing points = the first m TRAINING rows, d = 20 / m = 1024 (FITC,
BASELINE configs[3]) and d = 50 / m = 2048 (VFE, configs[4]) — the shapes, conditioning and tile orders
the reported numbers are measured on, at row counts two CPU evaluations finish in a minute or two.

Bars: NLL 1e-9, gradient 1e-7 relative (north_star), fixed, against the independent NumPy/LAPACK
evaluation; the FITC NLL additionally 5e-11 / 1e-10 against the 80-bit evaluation of the same inputs.  Each test PRINTS the measured errors and, where the literal oracle is affordable, the
oracle-vs-NumPy disagreement on the same inputs (the floor any fp64 implementation can be asked to meet
against that oracle); DESIGN.md §1 quotes these numbers."""
import numpy as np
import pytest

import np_gp
import oracle
from gp_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpx():
    from gp_amd import lib
    return lib


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def bench_problem(model, n):
    """exactly bench.py's run_sparse inputs, truncated to n rows"""
    m, d, cfg = (1024, 20, 3) if model == "fitc" else (2048, 50, 4)
    seed = 20240601 + cfg
    hyp = synth.eval_hyps(d)
    X = synth.make_inputs(seed, n, d)
    y = synth.make_targets(seed, X, hyp)
    U = synth.make_inputs(seed, m, d, start_point=0)
    assert np.array_equal(U, X[:m])
    return X, y, U, hyp, 1e-8


def gpu_eval(gpx, model, X, y, U, hyp, jit):
    ctx = gpx.Context()
    ctx.set_data(X, y)
    ctx.set_inducing(U)
    M = gpx.MODEL_FITC if model == "fitc" else gpx.MODEL_VFE
    nll, g = ctx.nll_grad(M, 0, hyp, jitter=jit)
    ctx.close()
    return nll, g


@pytest.mark.parametrize("model", ["fitc", "vfe"])
def test_bench_conditions_three_way_small_n(gpx, model):
    """n = 4096 rows: GPU vs the literal oracle vs NumPy/LAPACK, all three pairings printed"""
    X, y, U, hyp, jit = bench_problem(model, 4096)
    nll, g = gpu_eval(gpx, model, X, y, U, hyp, jit)
    if model == "fitc":   # the literal per-hyper loops: all host cores (same arithmetic, independent columns in parallel)
        nll_o, g_o = oracle.fitc_nll_all_cores(0, X, y, U, jit, hyp)
        nll_n, g_n = np_gp.fitc_nll(0, X, y, U, jit, hyp)
    else:
        nll_o, g_o = oracle.vfe_nll_all_cores(0, X, y, U, jit, hyp)
        nll_n, g_n = np_gp.vfe_nll(0, X, y, U, jit, hyp)
    print(f"\n[bench-conditions {model} n=4096] NLL rel err gpu-oracle {abs(nll - nll_o) / abs(nll_o):.2e} "
          f"gpu-numpy {abs(nll - nll_n) / abs(nll_n):.2e} oracle-numpy {abs(nll_o - nll_n) / abs(nll_n):.2e}; "
          f"grad rel err gpu-oracle {rel(g, g_o):.2e} gpu-numpy {rel(g, g_n):.2e} oracle-numpy {rel(g_o, g_n):.2e}")
    assert abs(nll - nll_n) <= 1e-9 * abs(nll_n)   # fixed north-star bar against the LAPACK evaluation, both models
    if model == "fitc":
        # The FITC value at these conditions (jitter 1e-8, inducing points that ARE training points: pivots of Kuu down
        # to 1e-4) as the reference forms it - A = sn2 Kuu + Kux G^-1 Kxu, cond(A) = 1.2e9, data fit F^T A^-1 F / sn2 -
        # has a rounding floor near 1e-9 of its own: against the 80-bit evaluation of the same inputs
        # (tools/longdouble_fitc.py 4096, repo-generated constant below: parity unpinned) LAPACK is 2.4e-10 away and the
        # literal reference arithmetic 1.7e-9; an A formed in 80 bits and merely ROUNDED to fp64 is 5.7e-10 away
        # (tools/study_fitc_terms.py cpu).  The library therefore evaluates the whitened form M = sn2 I + V^T G^-1 V
        # (cond 2e4; sparse.hip): measured 7.7e-13 from the 80-bit value on MI355X (round 3; 1.0e-9 in round 2).
        nll_ld = 485.69591464012831117
        e_gpu, e_orc, e_np = abs(nll - nll_ld) / nll_ld, abs(nll_o - nll_ld) / nll_ld, abs(nll_n - nll_ld) / nll_ld
        print(f"[bench-conditions fitc n=4096] NLL rel err against the 80-bit evaluation: gpu {e_gpu:.2e} oracle {e_orc:.2e} numpy {e_np:.2e}")
        assert 1e-9 < e_orc < 3e-9 and e_np < 5e-10   # the fixture value belongs to these inputs
        assert e_gpu <= 5e-11
        # and the distance to the literal oracle is the oracle's own error, no more
        assert abs(nll - nll_o) / abs(nll_o) <= e_orc + 1e-10
    assert rel(g, g_n) <= 1e-7
    # against the literal oracle the bar is widened ONLY by the printed oracle-vs-NumPy disagreement on the same
    # inputs (measured on MI355X at these conditions: FITC NLL 1.5e-9, gradient 1.6e-7 — the literal operation
    # order, explicit Kuu^-1 solves with jitter 1e-8 and inducing points that are training points, is itself
    # that far from LAPACK), and the widening is capped
    tol_f = max(1e-9, 2 * abs(nll_o - nll_n) / abs(nll_n))
    tol_g = max(1e-7, 2 * rel(g_o, g_n))
    assert tol_f <= 4e-9 and tol_g <= 4e-7, (tol_f, tol_g)   # measured: 2.9e-9 / 3.2e-7 (FITC), 1e-9 / 1e-7 (VFE)
    assert abs(nll - nll_o) <= tol_f * abs(nll_o)
    assert rel(g, g_o) <= tol_g


@pytest.mark.parametrize("model", ["fitc", "vfe"])
def test_bench_conditions_production_tile_orders(gpx, model):
    """n = 66000 rows (>= 65536: the tile orders, split-K slab counts and K-scale loops of the full-size runs)
    against the independent NumPy / LAPACK evaluation (pinned to the oracle in tests/test_oracle.py)"""
    n = 66000 if model == "fitc" else 34000
    X, y, U, hyp, jit = bench_problem(model, n)
    nll, g = gpu_eval(gpx, model, X, y, U, hyp, jit)
    nll_n, g_n = (np_gp.fitc_nll if model == "fitc" else np_gp.vfe_nll)(0, X, y, U, jit, hyp)
    print(f"\n[bench-conditions {model} n={n}] NLL rel err gpu-numpy {abs(nll - nll_n) / abs(nll_n):.2e}; "
          f"grad rel err gpu-numpy {rel(g, g_n):.2e}")
    assert abs(nll - nll_n) <= 1e-9 * abs(nll_n)   # north-star bar, both models (FITC: LAPACK itself is 7.2e-10 from the 80-bit value)
    assert rel(g, g_n) <= 1e-7
    if model == "fitc":
        # the 80-bit evaluation of exactly these inputs (tools/longdouble_fitc.py 66000, 50 minutes of CPU; repo-generated
        # constant, parity unpinned): LAPACK is 7.2e-10 from it; the GPU 3.2e-12 with the whitened form (round 3; 6.7e-10
        # in round 2)
        nll_ld = -3376.8940968135167182
        e_gpu, e_np = abs(nll - nll_ld) / abs(nll_ld), abs(nll_n - nll_ld) / abs(nll_ld)
        print(f"[bench-conditions fitc n={n}] NLL rel err against the 80-bit evaluation: gpu {e_gpu:.2e} numpy {e_np:.2e}")
        assert 5e-10 < e_np < 1e-9   # the fixture value belongs to these inputs
        assert e_gpu <= 1e-10


def test_vfe_full_size_n4M_directional_derivative(gpx):
    """BASELINE configs[4] at its full size: n = 2^22, m = 2048, d = 50 (138 GB on the device).  No CPU
    evaluation exists at this size; the size-independent property is the analytic gradient against a central
    difference of the NLL along a random direction."""
    n, m, d = 1 << 22, 2048, 50
    seed = 20240601 + 4
    hyp = synth.eval_hyps(d)
    X = synth.make_inputs(seed, n, d)
    y = synth.make_targets(seed, X, hyp)
    ctx = gpx.Context()
    ctx.set_data(X, y)
    ctx.set_inducing(X[:m].copy())
    del X
    rng = np.random.default_rng(3)
    v = rng.normal(size=hyp.size)
    v /= np.linalg.norm(v)
    h = 1e-4
    nll, g = ctx.nll_grad(gpx.MODEL_VFE, 0, hyp, jitter=1e-8)
    fp = ctx.nll_grad(gpx.MODEL_VFE, 0, hyp + h * v, calc_grad=False, jitter=1e-8)
    fm = ctx.nll_grad(gpx.MODEL_VFE, 0, hyp - h * v, calc_grad=False, jitter=1e-8)
    ctx.close()
    fd = (fp - fm) / (2 * h)
    print(f"\n[vfe n=2^22] nll {nll!r} directional derivative analytic {g @ v:.9e} central difference {fd:.9e}")
    assert np.isfinite(nll) and np.all(np.isfinite(g))
    assert abs(fd - g @ v) <= 1e-5 * max(1.0, np.linalg.norm(g))
