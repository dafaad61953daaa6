// exact.hip — exact-GP objective, gradient, _setK and predict on one GPU (GP.cpp:120-176, 273-334,
// 423-444).  The n x n matrix lives in ONE buffer A: K + sn2 I (lower) -> L in place -> after the
// triangular inverse W = L^-1 went to its own buffer, (K + sn2 I)^-1 = W^T W overwrites A; the
// gradient trace terms are then one fused pass over the lower triangle of A that recomputes K_ab.
#include "ctx.h"
#include <cstring>
#include <limits>

namespace {

__global__ void vec_shift_kernel(long n, const double* __restrict__ y, double shift, double scale,
                                 double* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = scale * (y[i] - shift);
}

// out3 = { sum a_i b_i, sum a_i, sum b_i }  (single workgroup, deterministic)
__global__ __launch_bounds__(1024) void reduce3_kernel(long n, const double* __restrict__ a,
                                                       const double* __restrict__ b, double* __restrict__ out3) {
    __shared__ double red[3][1024];
    double s0 = 0, s1 = 0, s2 = 0;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const double x = a[i], y = b[i];
        s0 = fma(x, y, s0);
        s1 += x;
        s2 += y;
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) out3[threadIdx.x] = red[threadIdx.x][0];
}

// out[j] = sum_i V[i + j*ldv]^2, one wave per column
__global__ __launch_bounds__(256) void colsumsq_kernel(long n, long t, const double* __restrict__ V, long ldv,
                                                       double* __restrict__ out) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * 4 + w;
    if (j >= t) return;
    const double* v = V + j * ldv;
    double s = 0.0;
    for (long i = l; i < n; i += 64) s = fma(v[i], v[i], s);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (l == 0) out[j] = s;
}

__global__ void predict_finish_kernel(long t, const double* __restrict__ ksa, const double* __restrict__ vsq,
                                      double mean, double sf2, double sn2, double* __restrict__ mu,
                                      double* __restrict__ s2) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= t) return;
    if (mu) mu[j] = mean + ksa[j];
    if (s2) s2[j] = fmax(sf2 - vsq[j], 0.0) + sn2;  // GP.cpp:283
}

// one workgroup per test point j; for every dimension a block reduction over the p kernel points
__global__ __launch_bounds__(256) void predict_grad_kernel(long p, int d, const double* __restrict__ Xt, long ldx,
                                                           const double* __restrict__ Xst, long ldt, ScaleArgs sc,
                                                           const double* __restrict__ Ks, long ldk,
                                                           const double* __restrict__ alpha, const double* __restrict__ Q1,
                                                           const double* __restrict__ Q2, long ldq, double c2,
                                                           double* __restrict__ gmu, double* __restrict__ gs2) {
    __shared__ double red[2][4];
    const long j = blockIdx.x;
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = 0; i < d; ++i) {
        const double il = sc.divide ? 1.0 / sc.s[i] : sc.s[i];  // 1 / l_i
        const double xs = Xst[(long)i * ldt + j];
        double s1 = 0.0, s2 = 0.0;
        for (long b = threadIdx.x; b < p; b += 256) {
            const double dk = Ks[b + j * ldk] * ((Xt[(long)i * ldx + b] - xs) * il);
            s1 = fma(dk, alpha[b], s1);
            if (gs2) {
                double q = Q1[b + j * ldq];
                if (Q2) q -= c2 * Q2[b + j * ldq];
                s2 = fma(dk, q, s2);
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        __syncthreads();
        if (l == 0) {
            red[0][w] = s1;
            red[1][w] = s2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (gmu) gmu[i + j * d] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            if (gs2) gs2[i + j * d] = 0.0 - 2 * ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));  // diag_dk_dx1 == 0
        }
    }
}

const double INF = std::numeric_limits<double>::infinity();

// ---- optional residual part of _check_SPD (GP.cpp:455-456): inv_err = mean |I - K K^-1| ---------------------------
// unit columns e_col[jj], col[jj] = jj * n / cols
__global__ void unit_columns_kernel(long n, int cols, double* __restrict__ V, long ldv) {
    const int jj = blockIdx.x * blockDim.x + threadIdx.x;
    if (jj < cols) V[(long)jj * n / cols + (long)jj * ldv] = 1.0;
}
// slot[0] = sum over the chunk's rows i and the columns jj of | [i0 + i == col[jj]] - (R(i, jj) + sn2 V(i0 + i, jj)) |
__global__ __launch_bounds__(1024) void spd_residual_kernel(long n, long i0, long tc, int cols, const double* __restrict__ R,
                                                            long ldr, const double* __restrict__ V, long ldv, double sn2,
                                                            double* __restrict__ slot) {
    __shared__ double red[1024];
    double s = 0.0;
    for (long e = threadIdx.x; e < tc * cols; e += 1024) {
        const long i = e % tc, jj = e / tc;
        const double kk = R[i + jj * ldr] + sn2 * V[i0 + i + jj * ldv];
        s += fabs(((i0 + i == jj * n / cols) ? 1.0 : 0.0) - kk);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) slot[0] = red[0];
}

// After a successful factorisation (L in c->A, base-block inverses in c->Wd, scaled inputs in c->Xt): the mean of
// |e_j - (K + sn2 I) ((K + sn2 I)^-1 e_j)| over `cols` evenly spaced columns j (cols >= n: every column = the
// reference's mean over all n^2 entries).  K is rebuilt 2048 rows at a time (it was overwritten by its factor), the
// products run on the matrix cores: 2 triangular solves with `cols` right-hand sides + n x cols x n flops.
int spd_residual(gpx_ctx* c, const HypView& h, int cols, double* inv_err) {
    const long n = c->n, ld = gpx_ld(n), chunk = 2048;
    if (cols > n) cols = (int)n;
    const long nchunk = (n + chunk - 1) / chunk, ldr = gpx_ld(chunk);
    int rc;
    if ((rc = c->pred_K.ensure(ld * cols)) || (rc = c->pred_K2.ensure(ld * chunk)) || (rc = c->pred_g.ensure(ldr * cols)) ||
        (rc = c->pred_out.ensure(nchunk + 8)))
        return rc;
    double* V = c->pred_K.p;
    HIP_TRY(hipMemsetAsync(V, 0, (size_t)ld * cols * sizeof(double), c->st));
    hipLaunchKernelGGL(unit_columns_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, c->st, n, cols, V, ld);
    if ((rc = gpx_trsm_left(c->st, n, cols, c->A.p, c->lda, c->Wd.p, V, ld, 0))) return rc;
    if ((rc = gpx_trsm_left(c->st, n, cols, c->A.p, c->lda, c->Wd.p, V, ld, 1))) return rc;
    for (long q = 0; q < nchunk; ++q) {
        const long i0 = q * chunk, tc = n - i0 < chunk ? n - i0 : chunk;
        // Kc = k(X, X[i0 : i0 + tc]) (n x tc); R = Kc^T V (tc x cols)
        if ((rc = gpx_cov_build(c->st, c->Xt.p, c->ldx, n, c->Xt.p + i0, c->ldx, tc, c->dpad, h.sf2, c->pred_K2.p, ld, 0, 0.0)))
            return rc;
        if ((rc = gpx_gemm(c->st, 1, 1, tc, cols, n, 1.0, c->pred_K2.p, ld, V, ld, 0.0, c->pred_g.p, ldr, 0))) return rc;
        hipLaunchKernelGGL(spd_residual_kernel, dim3(1), dim3(1024), 0, c->st, n, i0, tc, cols, c->pred_g.p, ldr, V, ld, h.sn2,
                           c->pred_out.p + q);
    }
    std::vector<double> part(nchunk);
    HIP_TRY(hipMemcpyAsync(part.data(), c->pred_out.p, nchunk * sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    double s = 0;
    for (double v : part) s += v;
    *inv_err = s / ((double)n * (double)cols);
    return GPX_OK;
}

// K + sn2 I -> c->A (lower triangle) from the scaled inputs in c->Xt (GP.cpp:128-129).  On the matrix cores where the
// dimension allows it (d <= 62; leaves the centred, augmented operands in c->Xa / Ua / UaT for the gradient trace), else by
// direct differences on the vector units; the diagonal is sf2 + sn2 exactly either way.
int build_K(gpx_ctx* c, const HypView& h) {
    const long n = c->n;
    const int d = c->d;
    int rc;
    if (c->mfma_pairwise && gpx_mfma_pair_supported(d)) {
        const long ldxa = gpx_mfma_ldxa(n), ldua = gpx_mfma_ldua(n);
        const int dk = gpx_mfma_dk(d), n2 = gpx_mfma_n2(d);
        if ((rc = c->Xa.ensure(dk * ldxa)) || (rc = c->Ua.ensure(dk * ldua)) || (rc = c->UaT.ensure(ldua * n2)) ||
            (rc = c->cen.ensure(GPX_MAX_DIM)))
            return rc;
        if ((rc = gpx_mfma_prepare(c->st, c->Xt.p, c->ldx, n, c->Xt.p, c->ldx, n, d, c->cen.p, c->Xa.p, c->Ua.p, c->UaT.p))) return rc;
        return gpx_mfma_cov_sym_build(c->st, c->Xa.p, c->Ua.p, n, d, h.sf2, h.sn2, c->A.p, c->lda);
    }
    return gpx_cov_build(c->st, c->Xt.p, c->ldx, n, c->Xt.p, c->ldx, n, c->dpad, h.sf2, c->A.p, c->lda, 1, h.sn2);
}

// K + sn2 I -> A (lower), factor in place, alpha = (K + sn2 I)^-1 (y - mean).  Leaves the device
// scalars scal[0] = sum log L_ii, scal[1..3] = {r.alpha, sum r, sum alpha}.
int factor_and_solve(gpx_ctx* c, int cov, const double* hyp, HypView& h) {
    const long n = c->n;
    int rc = parse_hyp(cov, c->d, hyp, h);
    if (rc) return rc;
    c->lda = gpx_ld(n);
    if ((rc = c->A.ensure(c->lda * n))) return rc;
    if ((rc = c->Wd.ensure(64L * ((n + 63) / 64 * 64)))) return rc;
    if ((rc = c->r.ensure(n)) || (rc = c->alpha.ensure(n)) || (rc = c->nalpha.ensure(n))) return rc;
    if ((rc = gpx_scale_inputs(c->st, c->X.p, n, c->d, c->dpad, h.sc, c->Xt.p, c->ldx))) return rc;
    HIP_TRY(hipMemsetAsync(c->info, 0, sizeof(int), c->st));
    // GP.cpp:128-129: K = k(X, X); decomp(K + sn2 * I)
    if ((rc = build_K(c, h))) return rc;
    if ((rc = gpx_potrf(c->st, n, c->A.p, c->lda, c->Wd.p, c->info))) return rc;
    // GP.cpp:138-141
    if ((rc = gpx_vec_shift(c->st, n, c->y.p, h.mean, 1.0, c->r.p))) return rc;
    HIP_TRY(hipMemcpyAsync(c->alpha.p, c->r.p, n * sizeof(double), hipMemcpyDeviceToDevice, c->st));
    if ((rc = gpx_trsv(c->st, n, c->A.p, c->lda, c->Wd.p, c->alpha.p, 0))) return rc;
    if ((rc = gpx_trsv(c->st, n, c->A.p, c->lda, c->Wd.p, c->alpha.p, 1))) return rc;
    if ((rc = gpx_logdiag_sum(c->st, n, c->A.p, c->lda, c->scal.p))) return rc;
    return gpx_reduce3(c->st, n, c->r.p, c->alpha.p, c->scal.p + 1);
}

}  // namespace

int gpx_vec_shift(hipStream_t st, long n, const double* y, double shift, double scale, double* out) {
    if (n <= 0) return GPX_OK;
    hipLaunchKernelGGL(vec_shift_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, y, shift, scale, out);
    return GPX_OK;
}
int gpx_reduce3(hipStream_t st, long n, const double* a, const double* b, double* out3) {
    hipLaunchKernelGGL(reduce3_kernel, dim3(1), dim3(1024), 0, st, n, a, b, out3);
    return GPX_OK;
}
int gpx_colsumsq(hipStream_t st, long n, long t, const double* V, long ldv, double* out) {
    if (t <= 0) return GPX_OK;
    hipLaunchKernelGGL(colsumsq_kernel, dim3((unsigned)((t + 3) / 4)), dim3(256), 0, st, n, t, V, ldv, out);
    return GPX_OK;
}

int gpx_predict_grad(hipStream_t st, long p, long t, int d, const double* Xt, long ldx, const double* Xst, long ldt,
                     const ScaleArgs& sc, const double* Kstar, long ldk, const double* alpha, const double* Q1,
                     const double* Q2, long ldq, double c2, double* gmu, double* gs2) {
    if (t <= 0) return GPX_OK;
    hipLaunchKernelGGL(predict_grad_kernel, dim3((unsigned)t), dim3(256), 0, st, p, d, Xt, ldx, Xst, ldt, sc, Kstar, ldk,
                       alpha, Q1, Q2, ldq, c2, gmu, gs2);
    return GPX_OK;
}

int gpx_exact_nll_grad(gpx_ctx* c, int cov, const double* hyp, int calc_grad, double* nll, double* grad) {
    const long n = c->n;
    const int d = c->d;
    if (n <= 0) return GPX_ERR_STATE;
    HypView h;
    int rc;
    if (!calc_grad) {
        if ((rc = factor_and_solve(c, cov, hyp, h))) return rc;
    } else {
        // gradient path: factor and invert in ONE recursion (W = L^-1), alpha = W^T (W r)
        if ((rc = parse_hyp(cov, d, hyp, h))) return rc;
        c->lda = gpx_ld(n);
        if ((rc = c->A.ensure(c->lda * n)) || (rc = c->W.ensure(c->lda * n)) || (rc = c->T.ensure(gpx_trtri_scratch(n))) ||
            (rc = c->r.ensure(n)) || (rc = c->alpha.ensure(n)) || (rc = c->nalpha.ensure(n)))
            return rc;
        if ((rc = gpx_scale_inputs(c->st, c->X.p, n, d, c->dpad, h.sc, c->Xt.p, c->ldx))) return rc;
        HIP_TRY(hipMemsetAsync(c->info, 0, sizeof(int), c->st));
        // W = L^-1 is written on and below the diagonal only; the products that read it skip the zero half
        // at tile granularity and so do read the upper part of the diagonal blocks, which must be zero.
        // Nothing ever writes there: one fill per (buffer, n, lda) instead of 34 GB (6 ms) per evaluation.
        if (c->W.p != c->w_clean_p || n != c->w_clean_n || c->lda != c->w_clean_lda) {
            HIP_TRY(hipMemsetAsync(c->W.p, 0, (size_t)c->lda * n * sizeof(double), c->st));
            c->w_clean_p = c->W.p;
            c->w_clean_n = n;
            c->w_clean_lda = c->lda;
        }
        if ((rc = build_K(c, h))) return rc;
        if ((rc = gpx_potrf_inv(c->st, n, c->A.p, c->lda, c->W.p, c->lda, c->T.p, c->info))) return rc;
        if ((rc = gpx_vec_shift(c->st, n, c->y.p, h.mean, 1.0, c->r.p))) return rc;
        if ((rc = gpx_gemv_n(c->st, n, n, 1.0, c->W.p, c->lda, c->r.p, 0.0, c->nalpha.p, 1))) return rc;
        if ((rc = gpx_gemv_t(c->st, n, n, 1.0, c->W.p, c->lda, c->nalpha.p, 0.0, c->alpha.p, 1))) return rc;
        if ((rc = gpx_logdiag_sum(c->st, n, c->A.p, c->lda, c->scal.p))) return rc;
        if ((rc = gpx_reduce3(c->st, n, c->r.p, c->alpha.p, c->scal.p + 1))) return rc;
    }
    const int nh = h.nk + 2;
    // a prepared state survives only an NLL-only evaluation at the very same hypers
    if (!(c->prepared_model == GPX_MODEL_EXACT && !calc_grad && c->prepared_cov == cov &&
          (int)c->prepared_hyp.size() == nh && !memcmp(c->prepared_hyp.data(), hyp, nh * sizeof(double))))
        c->prepared_model = -1;

    HIP_TRY(hipMemcpyAsync(c->host_scal, c->scal.p, 4 * sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipMemcpyAsync(c->host_scal + 8, c->info, sizeof(int), hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    int info;
    memcpy(&info, c->host_scal + 8, sizeof(int));
    *nll = INF;
    if (calc_grad)
        for (int i = 0; i < nh; ++i) grad[i] = INF;  // GP.cpp:133-134
    if (info != 0) return GPX_OK;
    const double data_fit = 0.5 * c->host_scal[1];
    const double model_complexity = c->host_scal[0];  // 0.5 * 2 * sum log L_ii
    const double norm_const = 0.5 * (double)n * log(2 * M_PI);
    double v = data_fit + model_complexity + norm_const;
    if (!std::isfinite(v)) return GPX_OK;  // GP.cpp:147-148
    *nll = v;
    if (!calc_grad) return GPX_OK;

    // Q = (K + sn2 I)^-1 - alpha alpha^T, g_i = 0.5 sum(Q o dK_i)   (GP.cpp:153-163)
    if ((rc = gpx_lauum(c->st, n, c->W.p, c->lda, c->A.p, c->lda))) return rc;
    if ((rc = gpx_vec_shift(c->st, n, c->alpha.p, 0.0, -1.0, c->nalpha.p))) return rc;
    TraceArgs ta = {};
    ta.Xt1 = c->Xt.p; ta.ld1 = c->ldx; ta.p = n;
    ta.Xt2 = c->Xt.p; ta.ld2 = c->ldx; ta.q = n;
    ta.dpad = c->dpad;
    ta.sf2 = h.sf2;
    ta.M1 = c->A.p; ta.ldm1 = c->lda;
    ta.u = c->nalpha.p; ta.v = c->alpha.p;
    ta.sym = 1;
    double* tout = c->scal.p + 16;
    if (c->mfma_pairwise && gpx_mfma_sym_trace_supported(d)) {
        // on the matrix cores: r2 from the norm expansion of the centred inputs, the per-dimension sums as a second
        // contraction (pairwise_mfma.hip); K itself keeps the direct differences (exact sf2 + sn2 diagonal)
        // (the centred, augmented operands were left in c->Xa / Ua / UaT by build_K at these hypers)
        const int n2 = gpx_mfma_n2(d), grid = gpx_mfma_sym_trace_grid(n, d);
        if ((rc = c->spart.ensure((long)grid * (n2 + 2)))) return rc;
        if ((rc = gpx_mfma_sym_trace(c->st, c->Xa.p, c->Ua.p, c->UaT.p, n, d, h.sf2, c->A.p, c->lda, c->nalpha.p, c->alpha.p,
                                     c->spart.p, tout)))
            return rc;
    } else if ((rc = gpx_pair_trace(c->st, ta, d, c->trace_ws.p, c->trace_ws.cap, tout))) {
        return rc;
    }
    HIP_TRY(hipMemcpyAsync(c->host_scal + 16, tout, (d + 2) * sizeof(double), hipMemcpyDeviceToHost, c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    const double* S = c->host_scal + 16;
    if (cov == GPX_COV_SE_ARD) {
        for (int i = 0; i < d; ++i) grad[i] = 0.5 * S[i];
    } else {
        double s = 0;
        for (int i = 0; i < d; ++i) s += S[i];
        grad[0] = 0.5 * s;
    }
    grad[h.nk - 1] = 0.5 * (2 * S[d]);       // dK/dlog sf = 2K
    grad[nh - 2] = h.sn2 * S[d + 1];         // sn2 * trace(Q)
    grad[nh - 1] = -1 * c->host_scal[3];     // -sum(alpha)
    bool fin = true;
    for (int i = 0; i < nh; ++i) fin = fin && std::isfinite(grad[i]);
    if (!fin) {  // GP.cpp:164-171
        *nll = INF;
        for (int i = 0; i < nh; ++i) grad[i] = INF;
    }
    return GPX_OK;
}

// GP::_setK.  SPD policy: the reference's _check_SPD (eigenvalues >= 0, mean|I - K K^-1| < 1e-4,
// factor OK; GP.cpp:445-461) costs ~5 n^3 extra and is infeasible at n = 65536; here a successful
// Cholesky (every pivot > 0) is the SPD test (a symmetric matrix whose Cholesky succeeds has positive
// eigenvalues), optionally followed by the residual test on sampled columns (gpx_set_spd_check), and
// the noise escalation loop is reproduced as is.
int gpx_exact_prepare(gpx_ctx* c, int cov, double* hyp) {
    if (c->n <= 0) return GPX_ERR_STATE;
    c->prepared_model = -1;
    HypView h;
    for (int iter = 0; iter < 200; ++iter) {
        int rc = factor_and_solve(c, cov, hyp, h);
        if (rc) return rc;
        int info = 0;
        HIP_TRY(hipMemcpyAsync(&info, c->info, sizeof(int), hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
        if (info == 0 && c->spd_check_cols > 0) {
            // the residual part of _check_SPD (GP.cpp:455-457: inv_err < 1e-4), on sampled columns
            if ((rc = spd_residual(c, h, c->spd_check_cols, &c->last_spd_residual))) return rc;
            if (!(c->last_spd_residual < 1e-4)) info = 1;
        }
        if (info == 0) {
            c->prepared_model = GPX_MODEL_EXACT;
            c->prepared_cov = cov;
            c->prepared_hyp.assign(hyp, hyp + h.nk + 2);
            return GPX_OK;
        }
        double& lsn = hyp[h.nk];  // GP.cpp:433
        lsn = std::isinf(lsn) ? log(std::numeric_limits<double>::epsilon()) : lsn + log(sqrt(10.0));
    }
    return GPX_ERR_STATE;
}

int gpx_exact_predict(gpx_ctx* c, const double* Xs, long t, double* mu, double* s2, double* gmu, double* gs2) {
    if (c->prepared_model != GPX_MODEL_EXACT) return GPX_ERR_STATE;
    const long n = c->n;
    const int d = c->d;
    HypView h;
    int rc = parse_hyp(c->prepared_cov, d, c->prepared_hyp.data(), h);
    if (rc) return rc;
    const bool grad = gmu || gs2;
    const long chunk = grad ? 512 : 4096;
    const long ldk = gpx_ld(n), ldt = gpx_ld(chunk);
    if ((rc = c->pred_K.ensure(ldk * chunk)) || (rc = c->pred_xs.ensure((long)d * chunk)) ||
        (rc = c->pred_xst.ensure((long)c->dpad * ldt)) || (rc = c->pred_out.ensure(4 * chunk)))
        return rc;
    if (grad && ((rc = c->pred_K2.ensure(ldk * chunk)) || (rc = c->pred_g.ensure(2L * d * chunk)))) return rc;
    double* d_ksa = c->pred_out.p;
    double* d_vsq = c->pred_out.p + chunk;
    double* d_mu = c->pred_out.p + 2 * chunk;
    double* d_s2 = c->pred_out.p + 3 * chunk;
    for (long t0 = 0; t0 < t; t0 += chunk) {
        const long tc = t - t0 < chunk ? t - t0 : chunk;
        HIP_TRY(hipMemcpyAsync(c->pred_xs.p, Xs + t0 * d, (size_t)tc * d * sizeof(double), hipMemcpyHostToDevice, c->st));
        if ((rc = gpx_scale_inputs(c->st, c->pred_xs.p, tc, d, c->dpad, h.sc, c->pred_xst.p, ldt))) return rc;
        // k_test^T = k(X, Xs): n x tc   (GP.cpp:280)
        if ((rc = gpx_cov_build(c->st, c->Xt.p, c->ldx, n, c->pred_xst.p, ldt, tc, c->dpad, h.sf2, c->pred_K.p, ldk, 0, 0.0)))
            return rc;
        if ((rc = gpx_gemv_t(c->st, n, tc, 1.0, c->pred_K.p, ldk, c->alpha.p, 0.0, d_ksa))) return rc;
        // the solves run on a copy when the kernel values are still needed for the gradients
        double* V = grad ? c->pred_K2.p : c->pred_K.p;
        if (grad) HIP_TRY(hipMemcpyAsync(V, c->pred_K.p, (size_t)ldk * tc * sizeof(double), hipMemcpyDeviceToDevice, c->st));
        if (s2 || gs2) {
            // k* K^-1 k*^T = |L^-1 k*^T|^2 per column (GP.cpp:281-283 uses the two-sided solve)
            if ((rc = gpx_trsm_left(c->st, n, tc, c->A.p, c->lda, c->Wd.p, V, ldk, 0))) return rc;
            if ((rc = gpx_colsumsq(c->st, n, tc, V, ldk, d_vsq))) return rc;
        }
        hipLaunchKernelGGL(predict_finish_kernel, dim3((unsigned)((tc + 255) / 256)), dim3(256), 0, c->st, tc, d_ksa,
                           d_vsq, h.mean, h.sf2, h.sn2, mu ? d_mu : nullptr, s2 ? d_s2 : nullptr);
        if (grad) {
            // kks = K^-1 k*^T (GP.cpp:281), then gy = dk * invKys, gs2 = -2 dk * kks (GP.cpp:292-294)
            if (gs2 && (rc = gpx_trsm_left(c->st, n, tc, c->A.p, c->lda, c->Wd.p, V, ldk, 1))) return rc;
            double* d_gmu = c->pred_g.p;
            double* d_gs2 = c->pred_g.p + (long)d * chunk;
            if ((rc = gpx_predict_grad(c->st, n, tc, d, c->Xt.p, c->ldx, c->pred_xst.p, ldt, h.sc, c->pred_K.p, ldk, c->alpha.p,
                                       V, nullptr, ldk, 0.0, gmu ? d_gmu : nullptr, gs2 ? d_gs2 : nullptr)))
                return rc;
            if (gmu) HIP_TRY(hipMemcpyAsync(gmu + t0 * d, d_gmu, (size_t)tc * d * sizeof(double), hipMemcpyDeviceToHost, c->st));
            if (gs2) HIP_TRY(hipMemcpyAsync(gs2 + t0 * d, d_gs2, (size_t)tc * d * sizeof(double), hipMemcpyDeviceToHost, c->st));
        }
        if (mu) HIP_TRY(hipMemcpyAsync(mu + t0, d_mu, tc * sizeof(double), hipMemcpyDeviceToHost, c->st));
        if (s2) HIP_TRY(hipMemcpyAsync(s2 + t0, d_s2, tc * sizeof(double), hipMemcpyDeviceToHost, c->st));
        HIP_TRY(hipStreamSynchronize(c->st));
    }
    return GPX_OK;
}
