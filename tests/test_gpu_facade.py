"""GPU tests of the C++ host facade (gp_amd/host: GP / FITC / VFE / CovSE* / MatrixSolver* with the
reference's class API) and of the three demo programs, against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "gp_amd", "host", "bin")
GOLD = os.path.join(os.path.dirname(__file__), "golden", "refdata_small.npz")
N, M = 256, 32


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope="module")
def workdir(tmp_path_factory):
    g = np.load(GOLD)
    d = tmp_path_factory.mktemp("facade")
    np.savetxt(d / "train_x", np.vstack([g["X"], g["U"]]), fmt="%.17g")   # rows = points (util.cpp:29-62)
    np.savetxt(d / "train_y", np.concatenate([g["y"], np.zeros(M)]), fmt="%.17g")
    np.savetxt(d / "test_x", g["Xs"], fmt="%.17g")
    return d


def run_check(workdir, fixed):
    exe = os.path.join(BIN, "facade_check")
    assert os.path.exists(exe), "build the host facade first: gp_amd/host/build.sh"
    out = subprocess.run([exe, str(N), str(M), str(fixed)], cwd=workdir, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    res = {}
    for line in out.stdout.splitlines():
        tok = line.split()
        if len(tok) >= 2 and tok[0][0].isalpha() and tok[0].replace("_", "").isalnum():
            try:
                res[tok[0]] = np.array([float(t) for t in tok[1:]])
            except ValueError:
                pass
    return res


def test_facade_fixed_hypers_match_oracle(workdir):
    g = np.load(GOLD)
    X, y, Xs, U = g["X"], g["y"], g["Xs"], g["U"]
    r = run_check(workdir, 1)
    hyp_def = oracle.default_hyps(0, X, y)
    assert rel(r["default_hyp"], hyp_def) < 1e-14               # GP.cpp:85-92
    hyp = hyp_def.copy()
    hyp[-2] = np.log(0.1 * np.std(y, ddof=1))
    jit = 1e-8                                                    # (0.1 * noise_lb)^2, FITC.cpp:30
    cases = {"gp": (lambda h: oracle.exact_nll(0, X, y, h), lambda: oracle.ExactState(0, X, y, hyp)),
             "fitc": (lambda h: oracle.fitc_nll(0, X, y, U, jit, h), lambda: oracle.SparseState("fitc", 0, X, y, U, jit, hyp)),
             "vfe": (lambda h: oracle.vfe_nll(0, X, y, U, jit, h), lambda: oracle.SparseState("vfe", 0, X, y, U, jit, hyp))}
    for tag, (fn, st) in cases.items():
        nll_o, g_o = fn(hyp)
        assert abs(r[f"{tag}_nll0"][0] - nll_o) <= 1e-9 * abs(nll_o)
        print(f"\n[facade {tag}] NLL rel err {abs(r[f'{tag}_nll0'][0] - nll_o) / abs(nll_o):.2e} gradient rel err {rel(r[f'{tag}_grad0'], g_o):.2e}")
        assert rel(r[f"{tag}_grad0"], g_o) <= 1e-7
        assert r[f"{tag}_fdrel"][0] < 1e-4                        # analytic vs central differences, GP.cpp:486-507
        assert rel(r[f"{tag}_hyp"], hyp) < 1e-14             # set_fixed(true): train only runs _setK
        assert abs(r[f"{tag}_nll"][0] - nll_o) <= 1e-9 * abs(nll_o)
        mu_o, s2_o = st().predict(Xs)[:2]
        assert rel(r[f"{tag}_mu"], mu_o) <= 1e-9 and rel(r[f"{tag}_s2"], s2_o) <= 1e-8
        assert abs(r[f"{tag}_single"][0] - mu_o[0]) <= 1e-9 * abs(mu_o[0])
        pg = st().predict(Xs[:1], grad=True)                      # predict*_with_grad, GP.h:106-115
        assert rel(r[f"{tag}_gy"], pg[2][0]) <= 1e-8 and abs(r[f"{tag}_yg"][0] - mu_o[0]) <= 1e-9 * abs(mu_o[0])
        if tag == "vfe":
            assert r["vfe_gs2_unsupported"][0] == 1               # VFE.cpp:109-112
        else:
            # FITC: difference of two solves against Kuu + 1e-8 I on values ~1e-17: cond(Kuu)-limited
            print(f"[facade {tag}] gs2 rel err {rel(r[f'{tag}_gs2'], pg[3][0]):.2e}")
            assert rel(r[f"{tag}_gs2"], pg[3][0]) <= (1e-7 if tag == "gp" else 1e-6) and np.array_equal(r[f"{tag}_gy2"], r[f"{tag}_gy"])
            assert abs(r[f"{tag}_sg"][0] - s2_o[0]) <= 1e-8 * abs(s2_o[0])
    # optimiser failure: VFE re-runs the start search (VFE.cpp:94-101), FITC keeps the point reached (FITC.cpp:94-100)
    assert list(r["vfe_failure_fallback"]) == [1, 1] and r["fitc_failure_keeps_point"][0] == 1
    Kiso = oracle.cov_k(1, np.array([0.3, 0.1]), X[:5], X[:7])
    assert rel(r["iso_k"], Kiso.T.ravel()) < 1e-14               # column-major 5 x 7
    dK0 = oracle.cov_dk_dhyp(1, np.array([0.3, 0.1]), 0, X[:5], X[:7], Kiso)
    assert rel(r["iso_dk0"], dK0.T.ravel()) < 1e-14             # Cov::dk_dhyp through the facade (device path)
    assert r["llt_spd"][0] == 1 and r["llt_inv_err"][0] < 1e-9


def test_facade_training_lowers_the_nll(workdir):
    r = run_check(workdir, 0)
    for tag in ("gp", "fitc", "vfe"):
        assert np.isfinite(r[f"{tag}_nll"][0])
        assert r[f"{tag}_nll"][0] <= r[f"{tag}_nll0"][0]          # the best point seen is returned
        if tag != "fitc":   # FITC with 32 far-away inducing points in d=24 starts on a plateau (K_xu ~ 0)
            assert r[f"{tag}_nll"][0] < r[f"{tag}_nll0"][0] - 1.0  # MLE improved on the start point
        assert np.all(np.isfinite(r[f"{tag}_mu"])) and np.all(r[f"{tag}_s2"] >= 0)


@pytest.mark.parametrize("prog,args", [("gp", ["200"]), ("fitc", ["200", "32"]), ("vfe", ["200", "32"])])
def test_demo_programs_write_pred(workdir, prog, args):
    exe = os.path.join(BIN, prog)
    out = subprocess.run([exe] + args, cwd=workdir, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Negative log likelihood:" in out.stdout
    pred = np.loadtxt(workdir / "pred")
    assert pred.shape == (64, 2) and np.all(np.isfinite(pred)) and np.all(pred[:, 1] >= 0)


@pytest.mark.parametrize("prog,args", [("gp", ["200"]), ("fitc", ["200", "32"]), ("vfe", ["200", "32"])])
def test_demo_pred_file_matches_oracle_predictions_at_fixed_hypers(workdir, prog, args):
    """`pred` (mygp.cpp:32-36, test_fitc.cpp:53-58, test_vfe.cpp:57-62: two columns mean, sqrt(s2), 6 significant
    digits) against the oracle's _setK + predict at the demo's own start hypers (GPX_DEMO_FIXED=1 makes train() skip
    the optimiser, GP.cpp:200-206), so the on-disk format either side of the path is pinned, not just its shape."""
    g = np.load(GOLD)
    n = 200
    Xall = np.vstack([g["X"], g["U"]])
    yall = np.concatenate([g["y"], np.zeros(M)])
    X, y, U, Xs = Xall[:n], yall[:n], Xall[-32:], g["Xs"]
    env = dict(os.environ, GPX_DEMO_FIXED="1")
    out = subprocess.run([os.path.join(BIN, prog)] + args, cwd=workdir, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    pred = np.loadtxt(workdir / "pred")
    hyp = oracle.default_hyps(0, X, y)                              # GP.cpp:85-92
    if prog == "gp":
        mu, s2 = oracle.ExactState(0, X, y, hyp).predict(Xs)
    else:
        if prog == "vfe":
            hyp[-2] = np.log(np.std(yall, ddof=1))                  # test_vfe.cpp:38-39: std of the WHOLE train_y file
        mu, s2 = oracle.SparseState(prog, 0, X, y, U, 1e-8, hyp).predict(Xs)[:2]
    assert pred.shape == (64, 2)
    assert np.allclose(pred[:, 0], mu, rtol=1e-5, atol=0)           # 6 significant digits on disk
    assert np.allclose(pred[:, 1], np.sqrt(s2), rtol=1e-5, atol=0)
    txt = (workdir / "pred").read_text().split()
    assert max(len(t.replace("-", "").replace(".", "").lstrip("0")) for t in txt if "e" not in t) <= 6
