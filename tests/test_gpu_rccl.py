"""The native RCCL hook behind the C++ facade (gpx_set_allreduce_rccl, SparseGP::set_communicator) and the C++ runner
gpx_dist, on the one GPU the test box has: world size 1 through a real ncclComm_t.  The N > 1 phase structure is
covered on the CPU (2-rank gloo test in tests/test_cpu_abi_and_sharding.py); the N > 1 RCCL run itself needs N GPUs."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gp_amd", "host", "bin")


def test_world_size_1_communicator_reproduces_the_hookless_result_bit_for_bit():
    exe = os.path.join(BIN, "rccl_check")
    assert os.path.exists(exe), "build the host programs first: gp_amd/host/build.sh"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    r = dict(line.split(None, 1) for line in out.stdout.splitlines() if line.strip())
    # FITC: #1, the m x m part of #2 on the side stream beside the trace, the trace sums behind it; VFE: #1 and #2
    for tag, per_eval in (("fitc", 3), ("vfe", 2)):
        assert r[f"{tag}_bit_identical"].strip() == "1"
        assert int(r[f"{tag}_collectives"]) == per_eval          # every all-reduce of one evaluation went through RCCL


def test_cpp_runner_single_rank_fitc_matches_the_python_path():
    """gpx_dist --gpus 1 (C++ generator, facade, RCCL world 1) against gp_amd.lib on the Python generator's data"""
    import numpy as np
    from gp_amd import lib, synth
    n, m, d = 20000, 256, 20
    exe = os.path.join(BIN, "gpx_dist")
    out = subprocess.run([exe, "--gpus", "1", "--model", "fitc", "--n", str(n), "--m", str(m), "--d", str(d), "--steps", "2"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    seed = 20240601 + 3
    hyp = synth.eval_hyps(d)
    X = synth.make_inputs(seed, n, d)
    y = synth.make_targets(seed, X, hyp)
    ctx = lib.Context()
    ctx.set_data(X, y)
    ctx.set_inducing(X[:m].copy())
    nll, g = ctx.nll_grad(lib.MODEL_FITC, 0, hyp, jitter=1e-8)
    ctx.close()
    print(f"\n[gpx_dist] {j}\n[python]   nll {nll!r}")
    assert j["world"] == 1 and j["collectives_per_eval"] == 3.0 and j["ms_per_eval"] > 0
    assert abs(j["nll"] - nll) <= 1e-9 * abs(nll)        # same problem from the C++ restatement of the generator
    assert abs(j["grad0"] - g[0]) <= 1e-7 * np.max(np.abs(g))
