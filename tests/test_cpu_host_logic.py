"""CPU tests of the host logic that needs no GPU (tests/cpp/host_logic_check.cpp, compiled here with
g++): the bounded L-BFGS behind NLopt's callback shape (GP.cpp:186-191, 231-265: clamp the start into
the bounds, honour maxeval, return the best point seen, forced stop) and dense.hpp's Eigen layout."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def res(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("host") / "host_logic_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "host_logic_check.cpp"),
                    os.path.join(ROOT, "gp_amd", "host", "optimizer.cpp")], check=True, timeout=120)
    out = subprocess.run([exe], capture_output=True, text=True, check=True, timeout=60).stdout
    r = {}
    for line in out.splitlines():
        k, *v = line.split()
        r[k] = v
    return r


def f(v):
    return np.array([float(x) for x in v])


def test_bounded_minimum_sits_on_the_active_bound(res):
    x = f(res["quad_x"])
    assert x[0] == 1.0 and abs(x[1] + 2) < 1e-6 and abs(x[2] - 0.5) < 1e-4     # upper bound cuts coordinate 0
    assert abs(float(res["quad_f"][0]) - 4.0) < 1e-9
    assert res["quad_status_ok"] == ["1"] and int(res["quad_evals"][0]) == int(res["quad_calls"][0]) <= 160


def test_start_point_is_clamped_into_the_bounds(res):
    x = f(res["clamp_x"])                                                      # GP.cpp:200-210
    assert x[0] == 1.0 and abs(x[1] + 2) < 1e-6 and abs(x[2] - 0.5) < 1e-4


def test_rosenbrock_inside_the_evaluation_budget(res):
    assert np.allclose(f(res["rosen_x"]), [1, 1], atol=1e-6) and float(res["rosen_f"][0]) < 1e-12
    assert int(res["rosen_evals"][0]) <= 400


def test_maxeval_is_honoured_and_best_point_returned(res):
    assert int(res["budget_evals"][0]) == 7 and res["budget_status"] == ["MAXEVAL_REACHED"]   # GP.cpp:232
    assert int(res["budget_calls"][0]) == 8 and res["budget_best_is_returned"] == ["1"]        # 7 + the re-check


def test_forced_stop_and_infinite_objective(res):
    assert res["forced_status"] == ["FORCED_STOP"] and res["forced_calls"] == ["3"] and res["forced_f_finite"] == ["1"]
    # +inf objective (non-SPD K, GP.cpp:132-134) beyond x0 = 0.5: the search backs off and stays finite
    assert float(res["inf_region_x0"][0]) <= 0.5 and res["inf_region_f_finite"] == ["1"]


def test_dense_stand_ins_are_column_major(res):
    assert f(res["dense_colmajor"]).tolist() == [10, 1, 12]
    assert f(res["dense_transpose"]).tolist() == [3, 2, 12] and f(res["dense_col"]).tolist() == [2, 12]
    assert f(res["dense_blocks"]).tolist() == [11, 12]


def test_synthetic_shards_concatenate_to_the_full_problem():
    """bench.py generates each rank's rows from the stream position of its first row: the shards must be
    exactly the rows of the single-GPU problem (inputs AND targets), or strong scaling compares different data."""
    import numpy as np
    from gp_amd import synth
    from gp_amd.sharding import shard_range
    seed, n, d = 20240604, 1001, 7
    hyp = synth.eval_hyps(d)
    X = synth.make_inputs(seed, n, d)
    y = synth.make_targets(seed, X, hyp)
    for world in (2, 3, 8):
        for rank in range(world):
            r0, r1 = shard_range(n, rank, world)
            Xr = synth.make_inputs(seed, r1 - r0, d, start_point=r0)
            yr = synth.make_targets(seed, Xr, hyp, start_point=r0)
            assert np.array_equal(Xr, X[r0:r1]) and np.allclose(yr, y[r0:r1], rtol=0, atol=1e-12)
