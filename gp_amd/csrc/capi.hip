// capi.hip — the high-level extern "C" entry points declared in include/gpx.h.
#include "ctx.h"
#include <cstdlib>
#include <cstring>
#include <new>

extern "C" {

// restores the caller's current device on every exit path of an entry point that has to switch it
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (dev >= 0 && dev != prev) {
            ok = hipSetDevice(dev) == hipSuccess;
            if (!ok) (void)hipGetLastError();   // the refusal must not surface later as another call's "last error"
        }
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

int gpx_ctx_create(int device, gpx_ctx** out) {
    if (!out) return GPX_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return GPX_ERR_ARG;
    HIP_TRY(hipSetDevice(device));
    gpx_ctx* c = new (std::nothrow) gpx_ctx();
    if (!c) return GPX_ERR_HIP;
    c->dev = device;
    {
        const char* e = getenv("GPX_NO_MFMA_PAIRWISE");
        c->mfma_pairwise = !(e && e[0] == '1');
        e = getenv("GPX_NO_PANELS");
        c->panels = !(e && e[0] == '1');
        e = getenv("GPX_FITC_NO_WHITEN");
        c->fitc_whiten = !(e && e[0] == '1');
        e = getenv("GPX_SPARSE_SERIAL_MM");
        c->sparse_side_mm = !(e && e[0] == '1');
        e = getenv("GPX_FITC_SINGLE_AR2");
        c->fitc_split_ar2 = !(e && e[0] == '1');
    }
    // every failing step releases what the steps before it created (gpx_ctx_destroy copes with a half-built context)
    auto build = [&]() -> int {
        // non-blocking: no implicit ordering against the caller's NULL-stream work (e.g. a collective a
        // framework issues from the all-reduce hook); every entry point drains its own streams
        HIP_TRY(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
        // the side stream carries short dependent chains that run beside a chip-filling GEMM: highest
        // priority, so that their workgroups take the next CU slot that frees up
        int prio_lo = 0, prio_hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        HIP_TRY(hipStreamCreateWithPriority(&c->st2, hipStreamNonBlocking, prio_hi));
        for (auto& e : c->ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->info), 64));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->host_scal), 1024 * sizeof(double)));
        int rc = c->scal.ensure(1024);
        if (!rc) rc = c->trace_ws.ensure(GPX_TRACE_WS);
        return rc;
    };
    const int rc = build();
    if (rc) {
        gpx_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return GPX_OK;
}

int gpx_ctx_destroy(gpx_ctx* c) {
    if (!c) return GPX_OK;
    DeviceGuard dg(c->dev);
    if (c->st) (void)hipStreamSynchronize(c->st);
    if (c->st2) (void)hipStreamSynchronize(c->st2);
    DevBuf* bufs[] = {&c->Tc, &c->X, &c->y, &c->U, &c->Xt, &c->Ut, &c->A, &c->W, &c->T, &c->Wd, &c->r, &c->alpha,
                      &c->nalpha, &c->scal, &c->trace_ws, &c->Kxu, &c->V, &c->Bm, &c->Cm, &c->Kuu, &c->Lu,
                      &c->Am, &c->LA, &c->Winv_u, &c->Winv_A, &c->Kuu_inv, &c->A_inv, &c->Wuu,
                      &c->Wd_u, &c->Wd_A, &c->red, &c->pred_K, &c->pred_xs, &c->pred_xst, &c->pred_out,
                      &c->pred_tmp, &c->pred_K2, &c->pred_g, &c->pLu, &c->pLA, &c->pWd_u, &c->pWd_A, &c->palpha, &c->pUt,
                      &c->Xa, &c->Ua, &c->UaT, &c->cen, &c->cpart, &c->spart, &c->cvec, &c->rowp1, &c->rowp2, &c->gemv_ws};
    for (DevBuf* b : bufs) b->release();
    for (auto& b : c->vn) b.release();
    for (auto& b : c->vm) b.release();
    if (c->info) (void)hipFree(c->info);
    if (c->host_scal) (void)hipHostFree(c->host_scal);
    if (c->st) (void)hipStreamDestroy(c->st);
    if (c->st2) (void)hipStreamDestroy(c->st2);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    gpx_rccl_release(c);
    delete c;
    return GPX_OK;
}

int gpx_set_data(gpx_ctx* c, const double* X, const double* y, int64_t n, int64_t d) {
    if (!c || !X || !y || n <= 0 || d <= 0) return GPX_ERR_ARG;
    if (d > GPX_MAX_DIM) return GPX_ERR_UNSUPPORTED;
    HIP_TRY(hipSetDevice(c->dev));
    c->prepared_model = -1;
    c->n = n;
    c->d = (int)d;
    c->dpad = (int)((d + 7) / 8 * 8);
    c->ldx = gpx_ld(n);
    int rc;
    if ((rc = c->X.ensure(n * d)) || (rc = c->y.ensure(n)) || (rc = c->Xt.ensure((long)c->dpad * c->ldx))) return rc;
    HIP_TRY(hipMemcpyAsync(c->X.p, X, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice, c->st));
    HIP_TRY(hipMemcpyAsync(c->y.p, y, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    return GPX_OK;
}

int gpx_set_inducing(gpx_ctx* c, const double* U, int64_t m) {
    if (!c || !U || m <= 0 || c->d <= 0) return GPX_ERR_ARG;
    HIP_TRY(hipSetDevice(c->dev));
    c->prepared_model = -1;
    c->m = m;
    c->ldu = gpx_ld(m);
    int rc;
    if ((rc = c->U.ensure(m * c->d)) || (rc = c->Ut.ensure((long)c->dpad * c->ldu))) return rc;
    HIP_TRY(hipMemcpyAsync(c->U.p, U, (size_t)m * c->d * sizeof(double), hipMemcpyHostToDevice, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    return GPX_OK;
}

int gpx_set_allreduce(gpx_ctx* c, gpx_allreduce_fn fn, void* user) {
    if (!c) return GPX_ERR_ARG;
    c->allreduce = fn;
    c->allreduce_stream = nullptr;
    c->rccl_comm = nullptr;
    c->allreduce_user = user;
    return GPX_OK;
}

int gpx_set_allreduce_stream(gpx_ctx* c, gpx_allreduce_stream_fn fn, void* user) {
    if (!c) return GPX_ERR_ARG;
    c->allreduce_stream = fn;
    c->allreduce = nullptr;
    c->rccl_comm = nullptr;
    c->allreduce_user = user;
    return GPX_OK;
}

int gpx_set_spd_check(gpx_ctx* c, int sampled_columns) {
    if (!c || sampled_columns < 0) return GPX_ERR_ARG;
    c->spd_check_cols = sampled_columns;
    return GPX_OK;
}

int gpx_last_spd_residual(gpx_ctx* c, double* inv_err) {
    if (!c || !inv_err) return GPX_ERR_ARG;
    *inv_err = c->last_spd_residual;
    return GPX_OK;
}

int gpx_nll_grad(gpx_ctx* c, int model, int cov, const double* hyp, int calc_grad, double jitter, double* nll,
                 double* grad) {
    if (!c || !hyp || !nll || (calc_grad && !grad)) return GPX_ERR_ARG;
    if (cov != GPX_COV_SE_ARD && cov != GPX_COV_SE_ISO) return GPX_ERR_ARG;
    HIP_TRY(hipSetDevice(c->dev));
    if (model == GPX_MODEL_EXACT) return gpx_exact_nll_grad(c, cov, hyp, calc_grad, nll, grad);
    if (model == GPX_MODEL_FITC || model == GPX_MODEL_VFE)
        return gpx_sparse_nll_grad(c, model, cov, hyp, calc_grad, jitter, nll, grad);
    return GPX_ERR_ARG;
}

int gpx_prepare(gpx_ctx* c, int model, int cov, double* hyp, double* jitter) {
    if (!c || !hyp) return GPX_ERR_ARG;
    if (cov != GPX_COV_SE_ARD && cov != GPX_COV_SE_ISO) return GPX_ERR_ARG;
    HIP_TRY(hipSetDevice(c->dev));
    if (model == GPX_MODEL_EXACT) return gpx_exact_prepare(c, cov, hyp);
    if ((model == GPX_MODEL_FITC || model == GPX_MODEL_VFE) && jitter)
        return gpx_sparse_prepare(c, model, cov, hyp, jitter);
    return GPX_ERR_ARG;
}

int gpx_predict_with_grad(gpx_ctx* c, const double* Xs, int64_t t, double* mu, double* s2, double* gmu, double* gs2) {
    if (!c || !Xs || t < 0) return GPX_ERR_ARG;
    if (t == 0) return GPX_OK;
    HIP_TRY(hipSetDevice(c->dev));
    if (c->prepared_model == GPX_MODEL_EXACT) return gpx_exact_predict(c, Xs, t, mu, s2, gmu, gs2);
    if (c->prepared_model == GPX_MODEL_VFE && gs2) return GPX_ERR_UNSUPPORTED;  // VFE.cpp:125-131 has no s2 gradient
    if (c->prepared_model == GPX_MODEL_FITC || c->prepared_model == GPX_MODEL_VFE)
        return gpx_sparse_predict(c, Xs, t, mu, s2, gmu, gs2);
    return GPX_ERR_STATE;
}

int gpx_predict(gpx_ctx* c, const double* Xs, int64_t t, double* mu, double* s2) {
    return gpx_predict_with_grad(c, Xs, t, mu, s2, nullptr, nullptr);
}

int gpx_cov_k(int cov, int64_t d, const double* hyp, const double* x1, int64_t p, const double* x2, int64_t q,
              double* K) {
    if (!hyp || !x1 || !x2 || !K || p <= 0 || q <= 0 || d <= 0) return GPX_ERR_ARG;
    if (d > GPX_MAX_DIM) return GPX_ERR_UNSUPPORTED;
    const int nk = cov == GPX_COV_SE_ARD ? (int)d + 1 : 2;
    std::vector<double> h(hyp, hyp + nk);
    h.push_back(0.0);
    h.push_back(0.0);
    HypView hv;
    int rc = parse_hyp(cov, (int)d, h.data(), hv);
    if (rc) return rc;
    const int dpad = (int)((d + 7) / 8 * 8);
    const long l1 = gpx_ld(p), l2 = gpx_ld(q);
    DevBuf X1, X2, T1, T2, Kd;
    rc = X1.ensure(p * d);
    if (!rc) rc = X2.ensure(q * d);
    if (!rc) rc = T1.ensure(dpad * l1);
    if (!rc) rc = T2.ensure(dpad * l2);
    if (!rc) rc = Kd.ensure(p * q);
    if (!rc) {
        rc = GPX_ERR_HIP;
        if (hipMemcpy(X1.p, x1, (size_t)p * d * 8, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(X2.p, x2, (size_t)q * d * 8, hipMemcpyHostToDevice) == hipSuccess) {
            rc = gpx_scale_inputs(0, X1.p, p, (int)d, dpad, hv.sc, T1.p, l1);
            if (!rc) rc = gpx_scale_inputs(0, X2.p, q, (int)d, dpad, hv.sc, T2.p, l2);
            if (!rc) rc = gpx_cov_build(0, T1.p, l1, p, T2.p, l2, q, dpad, hv.sf2, Kd.p, p, 0, 0.0);
            if (!rc && hipMemcpy(K, Kd.p, (size_t)p * q * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = GPX_ERR_HIP;
        }
    }
    X1.release(); X2.release(); T1.release(); T2.release(); Kd.release();
    return rc;
}

int gpx_cov_dk_dhyp(int cov, int64_t d, const double* hyp, int64_t idx, const double* x1, int64_t p, const double* x2,
                    int64_t q, const double* K, double* dK) {
    if (!hyp || !x1 || !x2 || !dK || p <= 0 || q <= 0 || d <= 0 || idx < 0) return GPX_ERR_ARG;
    if (d > GPX_MAX_DIM) return GPX_ERR_UNSUPPORTED;
    const int nk = cov == GPX_COV_SE_ARD ? (int)d + 1 : 2;
    if (idx >= nk) return GPX_ERR_ARG;   // "Wrong index for dk_dhyp" (CovSEiso.cpp:24-26)
    std::vector<double> h(hyp, hyp + nk);
    h.push_back(0.0);
    h.push_back(0.0);
    HypView hv;
    int rc = parse_hyp(cov, (int)d, h.data(), hv);
    if (rc) return rc;
    const int dpad = (int)((d + 7) / 8 * 8);
    const long l1 = gpx_ld(p), l2 = gpx_ld(q);
    const int mode = idx == nk - 1 ? 2 : (cov == GPX_COV_SE_ARD ? 0 : 1);
    DevBuf X1, X2, T1, T2, Kd, Od;
    rc = X1.ensure(p * d);
    if (!rc) rc = X2.ensure(q * d);
    if (!rc) rc = T1.ensure(dpad * l1);
    if (!rc) rc = T2.ensure(dpad * l2);
    if (!rc) rc = Kd.ensure(p * q);
    if (!rc) rc = Od.ensure(p * q);
    if (!rc) {
        rc = GPX_ERR_HIP;
        if (hipMemcpy(X1.p, x1, (size_t)p * d * 8, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(X2.p, x2, (size_t)q * d * 8, hipMemcpyHostToDevice) == hipSuccess &&
            (!K || hipMemcpy(Kd.p, K, (size_t)p * q * 8, hipMemcpyHostToDevice) == hipSuccess)) {
            rc = gpx_scale_inputs(0, X1.p, p, (int)d, dpad, hv.sc, T1.p, l1);
            if (!rc) rc = gpx_scale_inputs(0, X2.p, q, (int)d, dpad, hv.sc, T2.p, l2);
            if (!rc && !K) rc = gpx_cov_build(0, T1.p, l1, p, T2.p, l2, q, dpad, hv.sf2, Kd.p, p, 0, 0.0);   // the 4-argument overload (CovSEard.cpp:23-27)
            for (long q0 = 0; !rc && q0 < q; q0 += 32768) {   // grid.y limit
                const long qc = q - q0 < 32768 ? q - q0 : 32768;
                rc = gpx_dk_dhyp(0, p, qc, mode, (int)idx, (int)d, T1.p, l1, T2.p + q0, l2, Kd.p + q0 * p, p, Od.p + q0 * p, p);
            }
            if (!rc && hipMemcpy(dK, Od.p, (size_t)p * q * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = GPX_ERR_HIP;
        }
    }
    X1.release(); X2.release(); T1.release(); T2.release(); Kd.release(); Od.release();
    return rc;
}

// ---- MatrixSolverLLT with the factor kept on the device between decomp() and solve() / inverse() ----
// The handle belongs to ONE device (the `device` argument of gpx_chol_factor; -1 = the caller's current one); every
// entry point switches to it and restores the caller's current device on every exit path.  The solve / inverse
// staging buffers live in the handle: no hipMalloc / hipFree per solve() once they have grown to size.
struct gpx_chol {
    int dev = 0;
    long n = 0, ld = 0;
    DevBuf L, Wd;
    mutable DevBuf Bd, W, T, O;
    hipStream_t st = nullptr;
    bool spd = false;
};

static void chol_release(gpx_chol* f) {
    f->L.release(); f->Wd.release(); f->Bd.release(); f->W.release(); f->T.release(); f->O.release();
    if (f->st) (void)hipStreamDestroy(f->st);
    delete f;
}

static bool hip_bad(hipError_t e, int line) {
    if (e == hipSuccess) return false;
    fprintf(stderr, "gpx: HIP error %s at capi.hip:%d\n", hipGetErrorName(e), line);
    return true;
}

int gpx_chol_factor(int device, int64_t n, const double* A, gpx_chol** out, int* spd, double* logdet) {
    if (!A || !out || !spd || n <= 0) return GPX_ERR_ARG;
    *out = nullptr;
    DeviceGuard dg(device);
    if (!dg.ok) return GPX_ERR_ARG;
    gpx_chol* f = new (std::nothrow) gpx_chol();
    if (!f) return GPX_ERR_HIP;
    auto fail = [&](int r) { chol_release(f); return r; };
    if (hip_bad(hipGetDevice(&f->dev), __LINE__)) return fail(GPX_ERR_HIP);
    if (hip_bad(hipStreamCreateWithFlags(&f->st, hipStreamNonBlocking), __LINE__)) return fail(GPX_ERR_HIP);
    f->n = n;
    f->ld = gpx_ld(n);
    int rc = f->L.ensure(f->ld * n);
    if (!rc) rc = f->Wd.ensure(64L * ((n + 63) / 64 * 64) + 16);
    if (rc) return fail(rc);
    int* info = reinterpret_cast<int*>(f->Wd.p + 64L * ((n + 63) / 64 * 64));  // device flag + log-det slot behind Wd
    double* sc = f->Wd.p + 64L * ((n + 63) / 64 * 64) + 8;
    if (hip_bad(hipMemcpy2DAsync(f->L.p, f->ld * 8, A, n * 8, n * 8, n, hipMemcpyHostToDevice, f->st), __LINE__)) return fail(GPX_ERR_HIP);
    if (hip_bad(hipMemsetAsync(info, 0, 8, f->st), __LINE__)) return fail(GPX_ERR_HIP);
    if ((rc = gpx_potrf(f->st, n, f->L.p, f->ld, f->Wd.p, info))) return fail(rc);
    int hinfo = 0;
    if (hip_bad(hipMemcpyAsync(&hinfo, info, 4, hipMemcpyDeviceToHost, f->st), __LINE__)) return fail(GPX_ERR_HIP);
    if (hip_bad(hipStreamSynchronize(f->st), __LINE__)) return fail(GPX_ERR_HIP);
    f->spd = hinfo == 0;
    *spd = f->spd ? 1 : 0;
    if (f->spd && logdet) {
        if ((rc = gpx_logdiag_sum(f->st, n, f->L.p, f->ld, sc))) return fail(rc);
        double v;
        if (hip_bad(hipMemcpyAsync(&v, sc, 8, hipMemcpyDeviceToHost, f->st), __LINE__)) return fail(GPX_ERR_HIP);
        if (hip_bad(hipStreamSynchronize(f->st), __LINE__)) return fail(GPX_ERR_HIP);
        *logdet = 2 * v;
    }
    *out = f;
    return GPX_OK;
}

int gpx_chol_solve_factored(const gpx_chol* f, double* B, int64_t nrhs) {
    if (!f || !B || nrhs <= 0) return GPX_ERR_ARG;
    if (!f->spd) return GPX_ERR_STATE;
    DeviceGuard dg(f->dev);
    if (!dg.ok) return GPX_ERR_HIP;
    const long n = f->n, ld = f->ld;
    int rc = f->Bd.ensure(ld * nrhs);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(f->Bd.p, ld * 8, B, n * 8, n * 8, nrhs, hipMemcpyHostToDevice, f->st));
    if ((rc = gpx_trsm_left(f->st, n, nrhs, f->L.p, ld, f->Wd.p, f->Bd.p, ld, 0))) return rc;
    if ((rc = gpx_trsm_left(f->st, n, nrhs, f->L.p, ld, f->Wd.p, f->Bd.p, ld, 1))) return rc;
    HIP_TRY(hipMemcpy2DAsync(B, n * 8, f->Bd.p, ld * 8, n * 8, nrhs, hipMemcpyDeviceToHost, f->st));
    HIP_TRY(hipStreamSynchronize(f->st));
    return GPX_OK;
}

int gpx_chol_inverse_factored(const gpx_chol* f, double* Inv) {
    if (!f || !Inv) return GPX_ERR_ARG;
    if (!f->spd) return GPX_ERR_STATE;
    DeviceGuard dg(f->dev);
    if (!dg.ok) return GPX_ERR_HIP;
    const long n = f->n, ld = f->ld;
    int rc = f->W.ensure(ld * n);
    if (!rc) rc = f->T.ensure(gpx_trtri_scratch(n));
    if (!rc) rc = f->O.ensure(ld * n);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(f->W.p, 0, (size_t)ld * n * 8, f->st));
    if ((rc = gpx_trtri(f->st, n, f->L.p, ld, f->Wd.p, f->W.p, ld, f->T.p))) return rc;
    if ((rc = gpx_lauum(f->st, n, f->W.p, ld, f->O.p, ld))) return rc;
    if ((rc = gpx_symmetrize_lower(f->st, n, f->O.p, ld))) return rc;   // the device keeps only the lower tiles
    HIP_TRY(hipMemcpy2DAsync(Inv, n * 8, f->O.p, ld * 8, n * 8, n, hipMemcpyDeviceToHost, f->st));
    HIP_TRY(hipStreamSynchronize(f->st));
    return GPX_OK;
}

int gpx_chol_free(gpx_chol* f) {
    if (!f) return GPX_OK;
    DeviceGuard dg(f->dev);
    chol_release(f);
    return GPX_OK;
}

// one-shot flavour: decomp + solve + inverse + free on the caller's current device
int gpx_chol_solve(int64_t n, const double* A, int* spd, double* logdet, double* B, int64_t nrhs, double* Inv) {
    if (!A || !spd || n <= 0 || nrhs < 0 || (nrhs > 0 && !B)) return GPX_ERR_ARG;
    gpx_chol* f = nullptr;
    int rc = gpx_chol_factor(-1, n, A, &f, spd, logdet);
    if (rc) return rc;
    if (*spd && nrhs > 0) rc = gpx_chol_solve_factored(f, B, nrhs);
    if (!rc && *spd && Inv) rc = gpx_chol_inverse_factored(f, Inv);
    gpx_chol_free(f);
    return rc;
}

}  // extern "C"
