#!/usr/bin/env python3
"""bench.py — throughput of the GP training hot path on MI355X.

One "step" = one NLL + hyper-gradient evaluation (one call of the reference's virtual
_calcNegLogProb(hyp, g, true)), training data already resident in HBM, fixed evaluation hypers
(SURVEY.md §8d).  Workload at N=1: the exact GP, n=65536 d=20 (BASELINE.json configs[2], the
configuration the metric is quoted on).  The exact GP does not shard ("replicas only"), so with
--gpus N every rank runs one independent replica and `value` is their aggregate.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (public spec)


def cpu_baseline_exact(d, n_sample=2048, n_target=65536):
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
                      f"({1.0 / dt:.4f} evals/s); value is that scaled by (n/{n_target})^3 to n={n_target} (extrapolated)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--d", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from gp_amd import lib, synth
    n, d = args.n, args.d
    X, y, hyp, _ = synth.make_problem(2, n, d)
    ctx = lib.Context(local_rank)
    ctx.set_data(X, y)

    def sync():
        lib.check(lib.lib.gpx_device_synchronize(), "sync")
        if dist is not None:
            dist.barrier()
            import torch
            torch.cuda.synchronize()

    nll = None
    for _ in range(args.warmup):
        nll, g = ctx.nll_grad(lib.MODEL_EXACT, lib.COV_SE_ARD, hyp)
    # HIP-event timing of every large GEMM launch inside the timed region, on the launch stream
    lib.check(lib.lib.gpx_prof_gemm_begin(C.c_double(1e9), 8192), "prof_begin")
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        nll, g = ctx.nll_grad(lib.MODEL_EXACT, lib.COV_SE_ARD, hyp)
    sync()
    dt = time.perf_counter() - t0
    ms_, fl_, sk_ = C.c_double(0), C.c_double(0), C.c_double(0)
    nl_ = C.c_int64(0)
    lib.check(lib.lib.gpx_prof_gemm_end(C.byref(ms_), C.byref(fl_), C.byref(nl_), C.byref(sk_)), "prof_end")
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        peak = C.c_double(0)
        lib.check(lib.lib.gpx_bench_mfma_f64_peak(C.byref(peak)), "mfma_peak")
        evals = args.steps * world
        gemm_tf = fl_.value / (ms_.value * 1e-3) / 1e12 if ms_.value > 0 else 0.0
        out = {
            "metric": "NLL+grad evals/sec: exact GP n=65536 d=20; FITC n=1M m=1024; % fp64 peak",
            "value": evals / dt, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"exact GP SE-ARD NLL+grad, n={n} d={d} (BASELINE configs[2]); "
                                   f"{world} independent replica(s)", "n": n, "d": d,
                       "parallelism": "replicas" if world > 1 else "single"},
            "nll": nll,
            "algorithmic_tflops": (float(n) ** 3) * evals / dt / 1e12 / world,
            "frac_fp64_peak_whole_eval": (float(n) ** 3) * evals / dt / 1e12 / world / FP64_MFMA_PEAK_TFLOPS,
            "roofline": {"bound": "mfma", "achieved": gemm_tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": gemm_tf / FP64_MFMA_PEAK_TFLOPS, "traffic": None,
                         "kernel": "gemm_kernel<4,4,*,*> (fp64 MFMA GEMM/SYRK/TRMM tiles)",
                         "launches_timed": int(nl_.value), "gemm_ms_per_eval": ms_.value / args.steps,
                         "flops_timed_per_eval": fl_.value / args.steps,
                         "flops_in_untimed_small_launches_per_eval": sk_.value / args.steps,
                         "measured_bare_mfma_f64_tflops": peak.value},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_exact(d, 2048, n)
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
