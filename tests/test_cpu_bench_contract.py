"""The JSON line of bench.py, as committed beside the rocprofv3 summary of the same command
(profiles/r03_default/bench.json), carries every field of the driver's contract plus the `roofline` and
`cpu_baseline` objects, and its numbers are consistent with one another."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r03_default", "bench.json")) as f:
        lines = [l for l in f.read().splitlines() if l.strip()]
    assert len(lines) == 1            # stdout carries exactly one line
    return json.loads(lines[0])


def test_contract_fields():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and "workload" in d["config"] and "model" not in d["config"]
    assert "n=65536" in d["config"]["workload"] and "65536" in d["metric"]
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-9          # evals/s x s/eval


def test_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.5 < r["frac"] < 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    # the kernel-only rate bounds the end-to-end rate from above; both stay below the bare MFMA loop
    assert d["algorithmic_tflops_per_gpu"] < r["achieved"] < d["measured_bare_mfma_f64_tflops"] <= 78.6 * 1.01
    assert abs(r["algorithmic_flops_per_eval"] / 65536.0 ** 3 - 1.0) < 0.01   # n^3 flops per evaluation
    # traffic (separate --pmc passes) and the live timing are about the SAME launch set: every 128x128-tile GEMM launch
    assert r["traffic_launches_per_eval"] == r["launches_per_eval"] and r["traffic_source"].startswith("profiles/")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "evals/s" and c["value"] > 0 and ("n=4096" in c["sample"] or "n=3072" in c["sample"])
    assert c["host_cores_available"] >= 1 and c["nproc"] >= 1
    a = c["all_cores"]                                   # the all-cores leg with the core count stated
    assert a["cores"] >= 1 and a["value"] > c["value"] and c["fitc_all_cores"]["value"] > 0
    for h in d["hbm_rooflines"] + d["fitc"]["hbm_rooflines"]:   # the HBM-bound kernels, live HIP-event timing
        assert h["bound"] == "hbm" and h["peak"] == 8000.0 and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-12 and 0 < h["frac"] < 1
    assert d["exact8192"]["config"]["n"] == 8192 and d["exact8192"]["ms_per_step"] > 0
    # round 3: every BASELINE config in the one line — VFE n = 2^22 (configs[4]) and config0 (the reference's own data)
    v = d["vfe"]
    assert v["config"]["n"] == 1 << 22 and v["config"]["m"] == 2048 and v["config"]["d"] == 50 and v["ms_per_step"] > 0
    for h in v["hbm_rooflines"]:
        assert h["bound"] == "hbm" and 0 < h["frac"] < 1
    c0 = d["config0"]
    assert set(c0["sizes"]) == {"500", "2000", "8000"} and c0["mle_gp_8000"]["returncode"] == 0 and c0["mle_gp_8000"]["evaluations"] > 0
    for n, e in c0["sizes"].items():
        assert e["gpu_nll_grad_ms"] > 0 and e["gpu_setK_ms"] > 0 and e["gpu_predict_2000_ms"] > 0
        if n != "8000":
            assert e["nll_rel_err_vs_cpu_port"] <= 1e-9 and e["mu_rel_err_vs_cpu_port"] <= 1e-9 and e["cpu_port_nll_grad_s_1_thread"] > 0
    assert any(h["traffic"] for h in d["hbm_rooflines"])     # HBM-side bytes from the committed --pmc passes


def test_bench_gpus_n_starts_n_distinct_ranks_before_any_gpu_call():
    """`python bench.py --gpus N` (the driver's command shape, no outer torchrun) must produce N ranks by itself:
    --launch-dry makes every child report rank/world and exit before gp_amd.lib or torch.cuda is imported."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-dry"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["launch_dry"] and d["n_gpus"] == 3 and d["ranks"] == [0, 1, 2] and d["worlds"] == [3] and d["local_ranks"] == [0, 1, 2]


def test_bench_worker_under_an_outer_launcher_does_not_spawn_again():
    """under torch.distributed.run (WORLD_SIZE set) bench.py is a worker: one dry line per rank, no nested launch"""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="4", RANK="2", LOCAL_RANK="2")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-dry"],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip()) == {"launch_dry": True, "rank": 2, "world": 4, "local_rank": 2}
