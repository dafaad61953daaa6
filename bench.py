#!/usr/bin/env python3
"""bench.py — throughput of the GP training hot path on MI355X.

One "step" = one NLL + hyper-gradient evaluation (one call of the reference's virtual
_calcNegLogProb(hyp, g, true): GP.cpp:120, FITC.cpp:201, VFE.cpp:165), training data already
resident in HBM, fixed evaluation hypers.

  * headline (`value`): exact GP SE-ARD, n=65536 d=20 (BASELINE.json configs[2], the configuration
    the metric is quoted on).  The exact GP does not shard ("replicas only"): with --gpus N every
    rank runs one independent replica and `value` is their aggregate (weak scaling).
  * `fitc` (and `vfe` with --workload all|vfe): the sparse models with the training rows sharded
    over the N ranks and the m x m / m-vector partials summed by RCCL all-reduce (STRONG scaling:
    n is fixed, BASELINE.json configs[3], [4]); reported in the same JSON line.

Launch: `python bench.py` (N=1) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

# Streams share hardware queues round-robin (4 by default): with only 4, the collective's stream can land
# on the queue of the GEMM the library keeps running beside it and wait for that GEMM to finish.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (public spec)
METRIC = "NLL+grad evals/sec: exact GP n=65536 d=20; FITC n=1M m=1024; % fp64 peak"


def cpu_baseline_exact(d, n_sample, n_target):
    """The oracle (a scalar single-thread C port of the reference's literal operation order) timed
    on a bounded sample: one NLL+grad evaluation at n_sample, scaled by n^3 to the target n."""
    import oracle
    from gp_amd import synth
    X, y, hyp, _ = synth.make_problem(2, n_sample, d)
    t0 = time.perf_counter()
    oracle.exact_nll(oracle.COV_SE_ARD, X, y, hyp)
    dt = time.perf_counter() - t0
    scale = (n_target / n_sample) ** 3
    return {"value": 1.0 / (dt * scale), "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"1 NLL+grad eval of the exact GP at n={n_sample} d={d} took {dt:.2f} s on 1 host core "
                      f"({1.0 / dt:.4f} evals/s); value is that scaled by ({n_sample}/{n_target})^3 to n={n_target} "
                      f"(extrapolated; the reference as shipped is single-threaded, CMakeLists.txt:12)"}


class Dist:
    """torch.distributed (backend nccl == RCCL) only when WORLD_SIZE > 1."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.torch = None
        self.dist = None
        if self.world > 1 or os.environ.get("GPX_BENCH_FORCE_DIST") == "1":
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            self.torch, self.dist = torch, dist
            self._views = {}

    def allreduce_on_stream(self, ptr, count, stream):
        """ENQUEUE the in-place sum all-reduce of `count` doubles at a raw device pointer on the
        library's own stream (RCCL over xGMI): torch sees that stream as an ExternalStream, the NCCL
        process group orders its collective after it and makes it wait for the result — the host
        does not block, so the m x m chains that follow are queued ahead of the GPU."""
        torch = self.torch
        key = (ptr, count, stream)
        hit = self._views.get(key)
        if hit is None:
            class _Mem:
                __cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}
            ext = torch.cuda.ExternalStream(stream, device=torch.device("cuda", self.local_rank))
            with torch.cuda.stream(ext):
                t = torch.as_tensor(_Mem(), device=torch.device("cuda", self.local_rank))
            hit = self._views[key] = (ext, t)
        ext, t = hit
        with torch.cuda.stream(ext):
            self.dist.all_reduce(t)

    def sync(self, lib):
        lib.check(lib.lib.gpx_device_synchronize(), "sync")
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def pmc_traffic(n):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE doubled as the gfx950 correction prescribes, + WRITE_SIZE; tools/pmc_bench.sh)."""
    path = os.path.join(ROOT, "profiles", "r01_pmc", f"exact_n{n}_traffic.json")
    if not os.path.exists(path):
        return None
    tot, cnt = 0.0, 0
    for k, v in json.load(open(path)).items():
        if k.startswith("gemm_kernel<4, 4"):
            tot += (v["fetch_bytes_per_launch_corrected"] + v["write_bytes_per_launch"]) * v["launches"]
            cnt += v["launches"]
    return tot / cnt / 1e9 if cnt else None


def timed(D, lib, fn, steps, warmup, prof_min_flops):
    out = None
    for _ in range(warmup):
        out = fn()
    lib.check(lib.lib.gpx_prof_gemm_begin(C.c_double(prof_min_flops), 8192), "prof_begin")
    D.sync(lib)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    D.sync(lib)
    dt = time.perf_counter() - t0
    ms_, fl_, sk_, nl_ = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int64(0)
    lib.check(lib.lib.gpx_prof_gemm_end(C.byref(ms_), C.byref(fl_), C.byref(nl_), C.byref(sk_)), "prof_end")
    prof = {"gemm_ms": ms_.value / steps, "gemm_flops": fl_.value / steps, "launches": int(nl_.value) // max(steps, 1),
            "small_flops": sk_.value / steps}
    return D.max_over_ranks(dt), out, prof


def run_exact(D, lib, synth, n, d, steps, warmup, predict_t):
    X, y, hyp, Xs = synth.make_problem(2, n, d, max(predict_t, 1))
    ctx = lib.Context(D.local_rank)
    ctx.set_data(X, y)
    dt, (nll, g), prof = timed(D, lib, lambda: ctx.nll_grad(lib.MODEL_EXACT, lib.COV_SE_ARD, hyp), steps, warmup, 1e9)
    pred = None
    if predict_t > 0:
        # SURVEY 8(d) secondary metric: test points per second of _predict (mean + variance) after _setK;
        # host Xs in, host mu / s2 out (t*(d+2)*8 bytes over PCIe, included)
        t0 = time.perf_counter()
        ctx.prepare(lib.MODEL_EXACT, lib.COV_SE_ARD, hyp)
        t_prep = time.perf_counter() - t0
        ctx.predict(Xs[:256])
        t0 = time.perf_counter()
        mu, s2 = ctx.predict(Xs)
        t_pred = time.perf_counter() - t0
        fl = 2.0 * predict_t * n * d + float(predict_t) * n * n   # K* build + ONE triangular solve
        pred = {"value": predict_t / t_pred, "unit": "test points/s", "t": predict_t, "ms": 1e3 * t_pred,
                "algorithmic_tflops": fl / t_pred / 1e12, "prepare_ms (_setK: build + factor + alpha)": 1e3 * t_prep,
                "mean_abs_mu": float(abs(mu).mean()), "mean_s2": float(s2.mean())}
    ctx.close()
    return dt, nll, prof, pred


def run_sparse(D, lib, synth, model, n, m, d, steps, warmup, cfg_index):
    seed = 20240601 + cfg_index
    hyp = synth.eval_hyps(d)
    from gp_amd.sharding import shard_range
    r0, r1 = shard_range(n, D.rank, D.world)
    X = synth.make_inputs(seed, r1 - r0, d, start_point=r0)
    y = synth.make_targets(seed, X, hyp, start_point=r0)
    U = synth.make_inputs(seed, m, d, start_point=0)   # inducing points: the first m training points
    ctx = lib.Context(D.local_rank)
    ctx.set_data(X, y)
    ctx.set_inducing(U)
    if D.dist is not None:
        ctx.set_allreduce_stream(D.allreduce_on_stream)
    M = lib.MODEL_FITC if model == "fitc" else lib.MODEL_VFE
    jitter = 1e-8   # (0.1 * noise_lb)^2 with the default noise_lb = 1e-3 (FITC.cpp:30, GP.cpp:28)
    dt, (nll, g), prof = timed(D, lib, lambda: ctx.nll_grad(M, lib.COV_SE_ARD, hyp, jitter=jitter), steps, warmup, 1e8)
    ctx.close()
    return dt, nll, prof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="default", choices=["default", "exact", "fitc", "vfe", "all"])
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--d", type=int, default=20)
    ap.add_argument("--fitc-n", type=int, default=1 << 20)
    ap.add_argument("--fitc-m", type=int, default=1024)
    ap.add_argument("--vfe-n", type=int, default=1 << 22)
    ap.add_argument("--vfe-m", type=int, default=2048)
    ap.add_argument("--vfe-d", type=int, default=50)
    ap.add_argument("--sparse-steps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--predict-t", type=int, default=4096, help="test points of the exact-GP predict timing (0: skip)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: anything a library prints there meanwhile (RCCL's
    # version banner on the first collective) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    D = Dist()
    from gp_amd import lib, synth
    want = {"default": ("exact", "fitc"), "all": ("exact", "fitc", "vfe")}.get(args.workload, (args.workload,))
    out = {"metric": METRIC, "unit": "evals/s", "n_gpus": D.world, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "vs_baseline": None, "dtype": "f64", "data": "synthetic"}

    if "exact" in want:
        n, d = args.n, args.d
        dt, nll, prof, pred = run_exact(D, lib, synth, n, d, args.steps, args.warmup, args.predict_t)
        evals = args.steps * D.world
        gemm_tf = prof["gemm_flops"] / (prof["gemm_ms"] * 1e-3) / 1e12 if prof["gemm_ms"] > 0 else 0.0
        alg_tf = float(n) ** 3 * args.steps / dt / 1e12   # per GPU
        out.update({
            "value": evals / dt, "ms_per_step": 1e3 * dt / args.steps, "scaling": "weak",
            "config": {"workload": f"exact GP SE-ARD NLL+grad, n={n} d={d} (BASELINE configs[2]); "
                                   f"{D.world} independent replica(s), one per GPU", "n": n, "d": d,
                       "parallelism": "replicas" if D.world > 1 else "single"},
            "nll": nll, "algorithmic_tflops_per_gpu": alg_tf, "frac_fp64_peak_whole_eval": alg_tf / FP64_MFMA_PEAK_TFLOPS,
            "roofline": {"bound": "mfma", "achieved": gemm_tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": gemm_tf / FP64_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(n),
                         "traffic_unit": "GB per launch (L2-miss bytes incl. Infinity-Cache hits; separate --pmc passes, profiles/r01_pmc)",
                         "kernel": "gemm_kernel<4,4,*,*> (fp64 MFMA GEMM/SYRK/TRMM tiles; all launches >= 1 GFLOP)",
                         "launches_per_eval": prof["launches"], "kernel_ms_per_eval": prof["gemm_ms"],
                         "algorithmic_flops_per_eval": prof["gemm_flops"],
                         "flops_in_untimed_small_launches_per_eval": prof["small_flops"]},
        })
        if pred is not None:
            out["predict"] = pred
    for model in ("fitc", "vfe"):
        if model not in want:
            continue
        n, m, d = (args.fitc_n, args.fitc_m, args.d) if model == "fitc" else (args.vfe_n, args.vfe_m, args.vfe_d)
        dt, nll, prof = run_sparse(D, lib, synth, model, n, m, d, args.sparse_steps, 1, 3 if model == "fitc" else 4)
        flops = (6.0 if model == "fitc" else 3.0) * n * m * m
        tf = flops * args.sparse_steps / dt / 1e12
        sub = {"value": args.sparse_steps / dt, "unit": "evals/s", "ms_per_step": 1e3 * dt / args.sparse_steps,
               "scaling": "strong", "n_gpus": D.world, "steps": args.sparse_steps,
               "config": {"workload": f"{model.upper()} SE-ARD NLL+grad, n={n} m={m} d={d}, rows sharded over "
                                      f"{D.world} GPU(s), RCCL all-reduce of m x m partials", "n": n, "m": m, "d": d},
               "nll": nll, "algorithmic_tflops_total": tf,
               "frac_fp64_peak_per_gpu": tf / D.world / FP64_MFMA_PEAK_TFLOPS,
               "gemm_ms_per_eval_rank0": prof["gemm_ms"]}
        out[model] = sub
        if "exact" not in want and "value" not in out:
            out.update({"value": sub["value"], "ms_per_step": sub["ms_per_step"], "scaling": "strong",
                        "config": sub["config"], "steps": args.sparse_steps})

    if D.rank == 0:
        best = 0.0
        for _ in range(3):   # the first pass runs while the clocks still ramp
            peak = C.c_double(0)
            lib.check(lib.lib.gpx_bench_mfma_f64_peak(C.byref(peak)), "mfma_peak")
            best = max(best, peak.value)
        out["measured_bare_mfma_f64_tflops"] = best   # register-operand MFMA loop, 16 accumulators, 2 waves/SIMD
        if not args.no_cpu_baseline and D.world == 1 and "exact" in want:
            out["cpu_baseline"] = cpu_baseline_exact(args.d, min(4096, args.n), args.n)  # ~20 s of CPU work
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if D.dist is not None:
        D.dist.destroy_process_group()


if __name__ == "__main__":
    main()
