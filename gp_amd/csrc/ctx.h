// ctx.h — the opaque context behind the C ABI: device-resident training data and workspaces.
#pragma once
#include "gpx_internal.h"
#include "kernels.h"
#include "../../include/gpx.h"
#include <vector>

struct DevBuf {
    double* p = nullptr;
    long cap = 0;  // doubles
    int ensure(long count) {
        if (count <= cap) return GPX_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&p), (size_t)count * sizeof(double)));
        cap = count;
        return GPX_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct gpx_ctx {
    int dev = 0;
    // exact gradient path: the (buffer, n, lda) for which the upper triangle of W is known to be zero
    const double* w_clean_p = nullptr;
    long w_clean_n = 0, w_clean_lda = 0;
    hipStream_t st = nullptr;
    // side stream + events: the m x m factor chains of the sparse models run here, concurrently
    // with row-local work that does not depend on them (sparse.hip)
    hipStream_t st2 = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    // training data (device): X is d x n as given, y length n
    long n = 0, m = 0;
    int d = 0, dpad = 0;
    DevBuf X, y, U;
    DevBuf Xt, Ut;  // scaled, dimension-major [dpad][ld]
    long ldx = 0, ldu = 0;
    // dense workspaces
    DevBuf Tc;           // scratch of the Lu chain of VFE while c->T holds split-K slabs (sparse.hip)
    DevBuf A, W, T, Wd;  // exact: n x n factor / inverse, trtri scratch, base-block inverses
    long lda = 0;
    DevBuf r, alpha, nalpha;
    DevBuf scal;      // small device result vector
    DevBuf trace_ws;  // pair_trace partials
    int* info = nullptr;
    double* host_scal = nullptr;  // pinned
    // sparse workspaces (sparse.hip)
    DevBuf Kxu, V, Bm, Cm, Kuu, Lu, Am, LA, Winv_u, Winv_A, Kuu_inv, A_inv, Wuu, Wd_u, Wd_A;
    DevBuf vn[8], vm[8], red;
    DevBuf gemv_ws;   // partial sums of the row-split transposed GEMV (F = Kux r / F_v = V^T Gamma^-1 r)
    // matrix-core pairwise path (pairwise_mfma.hip): centred, augmented operands and its partial sums
    DevBuf Xa, Ua, UaT, cen, cpart, spart, cvec;
    DevBuf rowp1, rowp2;  // FITC: per-64-column partial row sums written by the GEMM row-dot epilogue [2 ceil(m/128)][ldn]
    bool panels = true;          // GPX_NO_PANELS=1: the n x m matrices of the sparse models plain column-major (A/B)
    bool fitc_whiten = true;     // GPX_FITC_NO_WHITEN=1: the reference's A = sn2 Kuu + Kux G^-1 Kxu instead of M = sn2 I + V^T G^-1 V (A/B)
    bool mfma_pairwise = true;   // GPX_NO_MFMA_PAIRWISE=1 keeps the vector-unit kernels (A/B runs, fallback tests)
    // GPX_SPARSE_SERIAL_MM=1 (A/B): the m x m products behind the two factor chains (Kuu^-1, W_M W_u, A^-1) on the main
    // stream in phase 3 and the W buffers zero-filled per evaluation, as before round 3's last change
    bool sparse_side_mm = true;
    bool fitc_split_ar2 = true;   // GPX_FITC_SINGLE_AR2=1: one all-reduce #2 behind the trace (A/B); default: m x m part beside the trace
    // Winv_u / Winv_A: the (buffer, m) whose upper triangle is known to be zero (gpx_potrf_inv writes the lower one only)
    const double* winv_clean_p[2] = {nullptr, nullptr};
    long winv_clean_m[2] = {0, 0};
    // all-reduce hook for row-sharded FITC/VFE
    gpx_allreduce_fn allreduce = nullptr;                 // host-synchronous flavour
    gpx_allreduce_stream_fn allreduce_stream = nullptr;   // stream-ordered flavour (preferred)
    void* allreduce_user = nullptr;
    // native RCCL flavour (gpx_set_allreduce_rccl): ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, stream)
    // issued by the library itself on its own stream; optional device timing of every collective
    void* rccl_comm = nullptr;
    bool rccl_timing = false;
    std::vector<hipEvent_t> rccl_ev;   // pairs
    size_t rccl_used = 0;
    // optional residual part of _check_SPD in gpx_prepare (exact model): columns sampled, last value
    int spd_check_cols = 0;
    double last_spd_residual = 0.0;
    // state left by gpx_prepare for gpx_predict
    int prepared_model = -1, prepared_cov = 0;
    std::vector<double> prepared_hyp;
    double prepared_jitter = 0;
    DevBuf pred_K, pred_K2, pred_xs, pred_xst, pred_out, pred_tmp, pred_g;
    DevBuf pLu, pLA, pWd_u, pWd_A, palpha, pUt;  // sparse _setK state, private copies (objective calls keep it valid)
};

struct HypView {
    int nk;            // kernel hypers: d+1 (ARD) or 2 (iso)
    double sf2, sn2, mean;
    ScaleArgs sc;
};
inline int parse_hyp(int cov, int d, const double* hyp, HypView& h) {
    if (d > GPX_MAX_DIM) return GPX_ERR_UNSUPPORTED;
    h.nk = cov == GPX_COV_SE_ARD ? d + 1 : 2;
    h.sf2 = exp(2 * hyp[h.nk - 1]);
    h.sn2 = exp(2 * hyp[h.nk]);
    h.mean = hyp[h.nk + 1];
    if (cov == GPX_COV_SE_ARD) {
        h.sc.divide = 0;
        for (int i = 0; i < d; ++i) h.sc.s[i] = exp(-1 * hyp[i]);  // CovSEard.cpp:9
    } else {
        h.sc.divide = 1;
        const double l = exp(hyp[0]);  // CovSEiso.cpp:8
        for (int i = 0; i < d; ++i) h.sc.s[i] = l;
    }
    return GPX_OK;
}

// rccl.hip: enqueue the in-place fp64 sum all-reduce on st through the communicator registered on c
int gpx_rccl_allreduce(gpx_ctx* c, double* buf, long count, hipStream_t st);
void gpx_rccl_release(gpx_ctx* c);
// exact.hip
int gpx_exact_nll_grad(gpx_ctx* c, int cov, const double* hyp, int calc_grad, double* nll, double* grad);
int gpx_exact_prepare(gpx_ctx* c, int cov, double* hyp);
int gpx_exact_predict(gpx_ctx* c, const double* Xs, long t, double* mu, double* s2, double* gmu, double* gs2);
// sparse.hip
int gpx_sparse_nll_grad(gpx_ctx* c, int model, int cov, const double* hyp, int calc_grad, double jitter,
                        double* nll, double* grad);
int gpx_sparse_prepare(gpx_ctx* c, int model, int cov, const double* hyp, double* jitter);
int gpx_sparse_predict(gpx_ctx* c, const double* Xs, long t, double* mu, double* s2, double* gmu, double* gs2);
int gpx_symmetrize_lower(hipStream_t st, long n, double* A, long lda);   // mirrors the lower triangle into the upper one
// small shared kernels (exact.hip)
int gpx_vec_shift(hipStream_t st, long n, const double* y, double shift, double scale, double* out);
int gpx_reduce3(hipStream_t st, long n, const double* a, const double* b, double* out3);
int gpx_colsumsq(hipStream_t st, long n, long t, const double* V, long ldv, double* out);
// Input gradients of the predictive mean / variance (GP.cpp:284-296; CovSEard.cpp:28-40):
//   gmu[i,j] = sum_b dk_ib alpha_b,  gs2[i,j] = -2 sum_b dk_ib (Q1[b,j] - c2*Q2[b,j]),
//   dk_ib = Kstar[b,j] * (x_ib - xs_ij) / l_i^2; Xt/Xst are the scaled dimension-major inputs.
int gpx_predict_grad(hipStream_t st, long p, long t, int d, const double* Xt, long ldx, const double* Xst, long ldt,
                     const ScaleArgs& sc, const double* Kstar, long ldk, const double* alpha, const double* Q1,
                     const double* Q2, long ldq, double c2, double* gmu, double* gs2);
