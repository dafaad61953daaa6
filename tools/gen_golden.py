"""Generates tests/golden/refdata_small.npz.

Inputs are DATA copied from the reference's own sample files (data/train_x, train_y, test_x,
test_y: first rows only).  The reference holds no expected outputs for them and cannot be built in
this image (no Eigen/NLopt/MVMO), so the expected values stored here come from this repository's
CPU oracle (oracle/gp_oracle.c) and were cross-checked against tests/np_gp.py when generated:
"parity unpinned" in the sense of the task statement.  Run from the repo root:
    python tools/gen_golden.py /root/reference/data
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import np_gp  # noqa: E402
import oracle  # noqa: E402


def main(data_dir):
    n, t, m = 256, 64, 32
    X = np.loadtxt(os.path.join(data_dir, "train_x"), max_rows=n + m)
    y = np.loadtxt(os.path.join(data_dir, "train_y"), max_rows=n + m)
    Xs = np.loadtxt(os.path.join(data_dir, "test_x"), max_rows=t)
    ys = np.loadtxt(os.path.join(data_dir, "test_y"), max_rows=t)
    U = X[n:n + m].copy()          # inducing points: rows the demos would take from the tail
    X, y = X[:n].copy(), y[:n].copy()
    out = dict(X=X, y=y, Xs=Xs, ys=ys, U=U)
    for cov, cname in ((oracle.COV_SE_ARD, "ard"), (oracle.COV_SE_ISO, "iso")):
        hyp = oracle.default_hyps(cov, X, y)           # GP.cpp:85-92
        lb, ub = oracle.hyp_range(cov, X, y)
        hyp_eval = hyp.copy()
        hyp_eval[-2] = np.log(np.std(y, ddof=1) * 0.1)  # a noise level at which K is well conditioned
        out[f"{cname}_default_hyp"] = hyp
        out[f"{cname}_lb"], out[f"{cname}_ub"] = lb, ub
        out[f"{cname}_hyp"] = hyp_eval
        nll, g = oracle.exact_nll(cov, X, y, hyp_eval)
        nll2, g2 = np_gp.exact_nll(cov, X, y, hyp_eval)
        assert abs(nll - nll2) <= 1e-11 * abs(nll2) and np.max(np.abs(g - g2)) <= 1e-9 * np.max(np.abs(g2))
        out[f"{cname}_exact_nll"], out[f"{cname}_exact_grad"] = nll, g
        st = oracle.ExactState(cov, X, y, hyp_eval)
        mu, s2 = st.predict(Xs)
        out[f"{cname}_exact_mu"], out[f"{cname}_exact_s2"] = mu, s2
        jit = 1e-8 * np.exp(2 * hyp_eval[-3]) * 1e4    # jitter scaled to sf^2 so Kuu stays well conditioned
        out[f"{cname}_jitter"] = jit
        for model, fn, fn2 in (("fitc", oracle.fitc_nll, np_gp.fitc_nll), ("vfe", oracle.vfe_nll, np_gp.vfe_nll)):
            nll, g = fn(cov, X, y, U, jit, hyp_eval)
            nll2, g2 = fn2(cov, X, y, U, jit, hyp_eval)
            assert abs(nll - nll2) <= 1e-10 * abs(nll2), (model, nll, nll2)
            assert np.max(np.abs(g - g2)) <= 1e-7 * np.max(np.abs(g2)), (model, g, g2)
            out[f"{cname}_{model}_nll"], out[f"{cname}_{model}_grad"] = nll, g
            st = oracle.SparseState(model, cov, X, y, U, jit, hyp_eval)
            mu, s2 = st.predict(Xs)[:2]
            out[f"{cname}_{model}_mu"], out[f"{cname}_{model}_s2"] = mu, s2
    path = os.path.join(ROOT, "tests", "golden", "refdata_small.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    rmse = np.sqrt(np.mean((out["ard_exact_mu"] - ys) ** 2))
    print("sanity RMSE of exact-GP mean vs data/test_y (never asserted):", rmse, "std(ys)", ys.std())


def full(data_dir):
    """tests/golden/refdata_full.npz: the reference's whole sample data set (BASELINE configs[0]: `gp num_train` on
    data/train_x|train_y, 8000 x 24, predictions on the 2000 rows of data/test_x) as DATA, plus the default hypers of
    GP.cpp:85-92 at the three sizes BASELINE.md section 2 names (from the oracle; the reference holds no outputs)."""
    X = np.loadtxt(os.path.join(data_dir, "train_x"))
    y = np.loadtxt(os.path.join(data_dir, "train_y"))
    Xs = np.loadtxt(os.path.join(data_dir, "test_x"))
    ys = np.loadtxt(os.path.join(data_dir, "test_y"))
    out = dict(X=X, y=y, Xs=Xs, ys=ys)
    for n in (500, 2000, 8000):
        out[f"ard_default_hyp_{n}"] = oracle.default_hyps(oracle.COV_SE_ARD, X[:n], y[:n])
    path = os.path.join(ROOT, "tests", "golden", "refdata_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes", X.shape, Xs.shape)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data")
    full(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data")
