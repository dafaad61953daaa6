"""CPU test of the facade's hyper bookkeeping: the optimiser's box `_hyps_lb/_ub`
(GP.cpp:514-534 over CovSEard.cpp:41-71 / CovSEiso.cpp:47-78), default hypers (GP.cpp:85-92) and the noise-free
packing `vec2hyp / hyp2vec` (GP.cpp:535-564), against the oracle's restatement and the values stored in the
fixture.  The facade makes no GPU call before its first evaluation, so tests/cpp/bounds_check.cpp runs here."""
import os
import subprocess

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "refdata_small.npz")


@pytest.fixture(scope="module")
def res(tmp_path_factory):
    d = tmp_path_factory.mktemp("bounds")
    g = np.load(GOLD)
    np.savetxt(d / "train_x", g["X"], fmt="%.17g")
    np.savetxt(d / "train_y", g["y"], fmt="%.17g")
    exe = str(d / "bounds_check")
    host = os.path.join(ROOT, "gp_amd")
    assert os.path.exists(os.path.join(host, "libgpxhost.so")), "build the host facade first: gp_amd/host/build.sh"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "bounds_check.cpp"),
                    "-L" + host, "-lgpxhost", "-lgpx", "-Wl,-rpath," + host], check=True, timeout=300)
    out = subprocess.run([exe], cwd=d, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    r = {}
    for line in out.stdout.splitlines():
        k, *v = line.split()
        r[k] = np.array([float(x) for x in v])
    return r


@pytest.mark.parametrize("cov,c", [(0, "ard"), (1, "iso")])
def test_bounds_and_defaults_match_oracle_and_fixture(res, cov, c):
    g = np.load(GOLD)
    lb, ub = oracle.hyp_range(cov, g["X"], g["y"])
    assert np.array_equal(res[f"{c}_lb"], lb) and np.array_equal(res[f"{c}_ub"], ub)
    assert np.array_equal(res[f"{c}_lb"], g[f"{c}_lb"]) and np.array_equal(res[f"{c}_ub"], g[f"{c}_ub"])
    assert np.all(res[f"{c}_lb"] < res[f"{c}_ub"])
    assert np.allclose(res[f"{c}_default"], g[f"{c}_default_hyp"], rtol=1e-15, atol=0)
    assert np.allclose(res[f"{c}_default_nlb"], oracle.default_hyps(cov, g["X"], g["y"], noise_lb=1e-2), rtol=1e-15, atol=0)


@pytest.mark.parametrize("c,nk", [("ard", 25), ("iso", 2)])
def test_noise_free_packing(res, c, nk):
    h = res[f"{c}_default"]
    assert np.array_equal(res[f"{c}_pack"], h)                                   # GP.cpp:552-564, noisy: identity
    p = res[f"{c}_pack_noise_free"]
    assert len(p) == nk + 1 and np.array_equal(p[:nk], h[:nk]) and p[-1] == h[-1]  # noise slot dropped, mean kept last
    u = res[f"{c}_unpack_noise_free"]
    assert len(u) == nk + 2 and np.array_equal(u[:nk], h[:nk]) and u[nk] == -np.inf and u[-1] == h[-1]   # GP.cpp:535-551
    assert res[f"{c}_default_noise_free"][nk] == -np.inf                         # GP.cpp:89


def test_cov_overloads_that_take_k(res):
    assert res["diag_dk_dhyp_rows"][0] == 27 and res["diag_overload_diff"][0] == 0   # rows = length of the hyp passed
    assert res["diag_dk_dhyp_sf"][0] == 2 * np.exp(2 * 0.25) and tuple(res["diag_dk_dx1_shape"]) == (24, 1)
