// sparse.hip — FITC and VFE objectives, gradients, _setK and predict (FITC.cpp:109-319,
// VFE.cpp:113-249), restructured into the O(n m^2) weight-matrix form of DESIGN.md §"Sparse":
//
//   phase 1 (row-local)  : Kxu, the n x m products, the m x m partial sums      -> all-reduce #1
//   phase 2 (replicated) : A, its Cholesky, a = A^-1 F, log-dets, the NLL
//   phase 3 (row-local)  : the weight matrix W_xu, sum_ab W_xu o dKxu/dtheta_i in one fused pass,
//                          partial B^T diag(w) B                                   -> all-reduce #2
//   phase 4 (replicated) : W_uu and sum W_uu o dKuu/dtheta_i
//
// Training rows may be sharded over ranks (one context per GPU holds its own rows); the two
// all-reduces go through the callback given to gpx_set_allreduce (RCCL over xGMI in bench.py).
// The per-hyper loops of the reference (FITC.cpp:269-308, VFE.cpp:213-237), which materialise
// dKxu and an n x m x m product PER HYPER, become one pass that is independent of d.
#include "ctx.h"
#include <cstdlib>
#include <cstring>
#include <limits>

#include <chrono>

namespace {

// GPX_TRACE=1: host-side stage markers of a sparse evaluation on stderr (ms since the call began): where the HOST waits
struct StageTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    StageTrace() : on(getenv("GPX_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {}
    void mark(const char* what) const {
        if (on)
            fprintf(stderr, "gpx trace %9.3f ms  %s\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
    }
};

const double INF = std::numeric_limits<double>::infinity();
constexpr int PW = 8;  // partial sums per block

// ---- block-partial helper ---------------------------------------------------------------------
__device__ __forceinline__ void block_partials(double (&v)[PW], double* __restrict__ part) {
    __shared__ double red[4][PW];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        double s = v[j];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (l == 0) red[w][j] = s;
    }
    __syncthreads();
    if (threadIdx.x < PW)
        part[(long)blockIdx.x * PW + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void sum_partials_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out, int nout) {
    __shared__ double red[256];
    const int j = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += part[(long)b * PW + j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && j < nout) out[j] = red[0];
}

// ---- FITC row kernels (one thread per training row; consecutive rows -> coalesced columns) ------
// Gamma, 1/Gamma, p = r/Gamma   (FITC.cpp:215-216, 253-254); diag(1/Gamma) Kxu is never formed: the
// Gram products take 1/Gamma as the K-dimension scale of the GEMM and C is scaled per row where it is used
// vv = rowsum(V o V) arrives as per-64-column partial sums from the row-dot epilogue of the GEMM that formed V
// (gemm_f64.hip), so V is not read again
__global__ __launch_bounds__(256) void fitc_gamma_kernel(long n, int nparts, const double* __restrict__ vv, long ldr,
                                                         const double* __restrict__ r, double sn2, double sf2,
                                                         double* __restrict__ g, double* __restrict__ p,
                                                         double* __restrict__ part) {
    const long a = (long)blockIdx.x * 256 + threadIdx.x;
    double v[PW] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (a < n) {
        double s = 0.0;
        for (int q = 0; q < nparts; ++q) s += vv[(long)q * ldr + a];
        const double Gam = (sn2 + sf2 - s) / sn2;
        const double ig = 1.0 / Gam;
        g[a] = ig;
        p[a] = ig * r[a];
        v[0] = log(Gam);
        v[1] = r[a] * (ig * r[a]);
        v[2] = 1.0;
    }
    block_partials(v, part);
}

// With Cu = Kxu A^-1 (C = diag(g) Cu): c1 = rowsum(C o diag(g) Kxu) = g^2 rowsum(Cu o Kxu), q = g (Kxu a), pq = p - q,
// w and the trace coefficients.  rowsum(Cu o Kxu) and Kxu a (FITC.cpp:219-220's Kxu * A^-1 F) arrive as per-64-column
// partial sums from the row-dot epilogue of the GEMM that formed Cu: neither Cu nor Kxu is read again
// (17.2 GB less traffic per evaluation at n = 2^20, m = 1024).
__global__ __launch_bounds__(256) void fitc_rowstats_kernel(long n, int nparts, const double* __restrict__ ck,
                                                            const double* __restrict__ ka, long ldr,
                                                            const double* __restrict__ g,
                                                            const double* __restrict__ p, double sn2,
                                                            double* __restrict__ c2coef, double* __restrict__ ucoef,
                                                            double* __restrict__ wv, double* __restrict__ part) {
    const long a = (long)blockIdx.x * 256 + threadIdx.x;
    double v[PW] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (a < n) {
        double s = 0.0, k = 0.0;
        for (int q = 0; q < nparts; ++q) {
            s += ck[(long)q * ldr + a];
            k += ka[(long)q * ldr + a];
        }
        const double c1 = g[a] * g[a] * s;
        const double q = g[a] * k;
        const double pq = p[a] - q;
        const double w = g[a] - c1 - pq * pq / sn2;
        wv[a] = w;
        c2coef[a] = -w / sn2;
        ucoef[a] = -pq / sn2;
        v[0] = g[a] - c1;
        v[1] = pq * pq;
        v[2] = pq;
        v[3] = w;
    }
    block_partials(v, part);
}

// alpha = (r - Ka)/sn2 (VFE.cpp:183) and its negation; partials: alpha^2, alpha, r^2 not needed here
__global__ __launch_bounds__(256) void vfe_alpha_kernel(long n, const double* __restrict__ r, const double* __restrict__ Ka,
                                                        double sn2, double* __restrict__ nalpha,
                                                        double* __restrict__ part) {
    const long a = (long)blockIdx.x * 256 + threadIdx.x;
    double v[PW] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (a < n) {
        const double al = (r[a] - Ka[a]) / sn2;
        nalpha[a] = -al;
        v[0] = al * al;
        v[1] = al;
    }
    block_partials(v, part);
}

__global__ __launch_bounds__(256) void vfe_rstats_kernel(long n, const double* __restrict__ r, double* __restrict__ part) {
    const long a = (long)blockIdx.x * 256 + threadIdx.x;
    double v[PW] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (a < n) {
        v[0] = r[a] * r[a];
        v[2] = 1.0;
    }
    block_partials(v, part);
}

// ---- m x m kernels ---------------------------------------------------------------------------
// out (lower) = ca*Kuu + cb*S + dadd*I
__global__ void mm_combine_lower_kernel(long m, double ca, const double* __restrict__ Kuu, long ldk, double cb,
                                        const double* __restrict__ S, long lds, double dadd, double* __restrict__ out,
                                        long ldo) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * m) return;
    const long i = idx % m, j = idx / m;
    if (i < j) return;
    double v = ca * Kuu[i + j * ldk] + cb * S[i + j * lds];
    if (i == j) v += dadd;
    out[i + j * ldo] = v;
}
// mirror the lower triangle into the upper one
__global__ void mm_symmetrize_kernel(long m, double* __restrict__ A, long lda) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * m) return;
    const long i = idx % m, j = idx / m;
    if (i > j) A[j + i * lda] = A[i + j * lda];
}
// out = c1*A1 + c2*A2 + c3*a a^T + c4*A4   (full m x m; NULL matrices are skipped)
__global__ void mm_combine_full_kernel(long m, double c1, const double* __restrict__ A1, long ld1, double c2,
                                       const double* __restrict__ A2, long ld2, double c3, const double* __restrict__ a,
                                       double c4, const double* __restrict__ A4, long ld4, double* __restrict__ out,
                                       long ldo) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * m) return;
    const long i = idx % m, j = idx / m;
    double v = 0.0;
    if (A1) v += c1 * A1[i + j * ld1];
    if (A2) v += c2 * A2[i + j * ld2];
    if (a) v += c3 * a[i] * a[j];
    if (A4) v += c4 * A4[i + j * ld4];
    out[i + j * ldo] = v;
}
// partial sums of A o B over the full m x m
__global__ __launch_bounds__(256) void mm_inner_kernel(long m, const double* __restrict__ A, long lda,
                                                       const double* __restrict__ B, long ldb, double* __restrict__ part) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    double v[PW] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (idx < m * m) {
        const long i = idx % m, j = idx / m;
        v[0] = A[i + j * lda] * B[i + j * ldb];
    }
    block_partials(v, part);
}
__global__ void dot_kernel(long m, const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (long i = threadIdx.x; i < m; i += 256) s = fma(x[i], y[i], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}
__global__ void predict_sparse_finish_kernel(long t, const double* __restrict__ ksa, const double* __restrict__ vu,
                                             const double* __restrict__ vA, double mean, double sf2, double sn2,
                                             int fitc, double* __restrict__ mu, double* __restrict__ s2) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= t) return;
    if (mu) mu[j] = ksa[j] + mean;
    if (s2) {
        if (fitc) s2[j] = fmax(sn2 + (sf2 - (vu[j] - sn2 * vA[j])), sn2);  // FITC.cpp:117,153
        else s2[j] = fmax(sf2 - (vu[j] - vA[j]), 0.0);                    // VFE.cpp:130
    }
}

inline unsigned nblk(long n) { return (unsigned)((n + 255) / 256); }

struct Sparse {
    gpx_ctx* c;
    hipStream_t st;
    long n, m, ldn, ldm;
    int d, model, cov;
    HypView h;
    // Row-panel layout of the n x m matrices Kxu, V, B, C (GemmPanels in gpx_internal.h): pan rows per panel = the K range
    // of one split-K slab of the Gram products, so that their K-contiguous operand columns are pan * 8 bytes apart (150 KB)
    // instead of n * 8 (8 MB): probe 20.9 -> 17.0 ms at n = 2^20, m = 1024 (tools/probe/gram_panel_probe.py).  pan = 0 (one
    // panel would do, the vector-unit fallback kernels, GPX_NO_PANELS=1): plain column-major with leading dimension ldn.
    long pan = 0, pst = 0, ldk = 0;   // ldk: leading dimension of the n x m matrices (pan or ldn)
    GemmPanels pn_of(bool a, bool b, bool cc, bool e) const {
        GemmPanels p = {};
        if (a) { p.panA = pan; p.pstA = pst; }
        if (b) { p.panB = pan; p.pstB = pst; }
        if (cc) { p.panC = pan; p.pstC = pst; }
        if (e) { p.panE = pan; p.pstE = pst; }
        return p;
    }

    bool mfma() const { return c->mfma_pairwise && gpx_mfma_pair_supported(d); }
    int build_max_grid = 0;   // > 0: the Kxu build runs on at most that many workgroups (kernels(): the chain beside it)
    long nparts() const { return 2 * ((m + 127) / 128); }   // parts of the GEMM row-dot epilogue (gpx_internal.h)
    // V = Kxu Lu^-T with rowsum(V o V) from the epilogue, then Gamma (FITC.cpp:213-216)
    int v_and_gamma(const double* Winv_u, const double* r, double sn2, double sf2, double* g, double* p) {
        GemmRowdot rd = {};
        rd.R1 = c->rowp1.p; rd.ldr = ldn;
        const GemmPanels pn = pn_of(true, false, true, false);
        int rc = gpx_gemm_rowdot(st, 0, 0, n, m, m, 1.0, c->Kxu.p, ldk, Winv_u, ldm, c->V.p, ldk, GEMM_KHI_N, rd, &pn);
        if (rc) return rc;
        hipLaunchKernelGGL(fitc_gamma_kernel, dim3(nblk(n)), dim3(256), 0, st, n, (int)nparts(), c->rowp1.p, ldn, r, sn2, sf2, g, p,
                           part());
        return GPX_OK;
    }
    int alloc(bool grad) {
        int rc;
        ldn = gpx_ld(n);
        ldm = gpx_ld(m);
        const long tcap = gpx_trtri_scratch(m) > 64 * ldm * m ? gpx_trtri_scratch(m) : 64 * ldm * m;   // trtri scratch / split-K slabs
        pan = pst = 0;
        ldk = ldn;
        if (mfma() && c->panels) {
            const long R = gpx_gemm_splitk_panel_rows(m, m, n, GEMM_LOWER, tcap);
            if (R < n) {
                pan = ldk = R;
                pst = R * m + (R * m) % 2;
            }
        }
        if (mfma()) {
            const long ldxa = gpx_mfma_ldxa(n), ldua = gpx_mfma_ldua(m);
            const int dk = gpx_mfma_dk(d), n2 = gpx_mfma_n2(d), grid = std::max(gpx_mfma_trace_grid(n, d, m, false), gpx_mfma_trace_grid(n, d, m, true));
            if ((rc = c->Xa.ensure(dk * ldxa)) || (rc = c->Ua.ensure(dk * ldua)) || (rc = c->UaT.ensure(ldua * n2)) ||
                (rc = c->cen.ensure(GPX_MAX_DIM)) || (rc = c->cvec.ensure(ldua)))
                return rc;
            if (grad && ((rc = c->cpart.ensure((long)grid * ldua)) || (rc = c->spart.ensure((long)grid * (n2 + 1))))) return rc;
        }
        const long nm = pan ? (n + pan - 1) / pan * pst : ldn * m, mm = ldm * m;
        if ((rc = c->Kxu.ensure(nm))) return rc;
        if (model == GPX_MODEL_FITC) {
            if ((rc = c->V.ensure(nm)) || (rc = c->rowp1.ensure(nparts() * ldn))) return rc;
            if (grad && ((rc = c->Bm.ensure(nm)) || (rc = c->Cm.ensure(nm)) || (rc = c->rowp2.ensure(nparts() * ldn)))) return rc;
        } else if (grad) {
            if ((rc = c->Cm.ensure(nm))) return rc;
        }
        DevBuf* mb[] = {&c->Kuu, &c->Lu, &c->Am, &c->LA, &c->Winv_u, &c->Winv_A, &c->Kuu_inv, &c->A_inv, &c->Wuu};
        for (DevBuf* b : mb)
            if ((rc = b->ensure(mm))) return rc;
        const long wdn = 64L * ((m + 63) / 64 * 64);
        if ((rc = c->Wd_u.ensure(wdn)) || (rc = c->Wd_A.ensure(wdn))) return rc;
        if ((rc = c->T.ensure(tcap))) return rc;
        for (int i = 0; i < 8; ++i)
            if ((rc = c->vn[i].ensure(n)) || (rc = c->vm[i].ensure(m + 80))) return rc;  // + 64: direct-to-LDS loads read whole 32-entry tiles
        if ((rc = c->red.ensure(m * m + m + GPX_MAX_DIM + 96)) || (rc = c->gemv_ws.ensure(gpx_gemv_t_split_ws(m)))) return rc;
        if ((rc = c->r.ensure(n))) return rc;
        if ((rc = c->pred_tmp.ensure((long)PW * (nblk(n) > nblk(m * m) ? nblk(n) : nblk(m * m)) + 64))) return rc;
        return GPX_OK;
    }
    double* part() { return c->pred_tmp.p; }
    int finish_partials(long count, double* out, int nout) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(PW), dim3(256), 0, st, part(), (int)nblk(count), out, nout);
        return GPX_OK;
    }
    // The reduced values are consumed by work queued on `consumer`.  ready: event after which buf is
    // complete (work queued on other streams behind it keeps running during the collective); without
    // one, buf is produced on `consumer` itself.  Stream-ordered hook: the collective is enqueued on
    // `consumer`, the host never waits.  Host-synchronous hook: wait for buf, call, return when reduced.
    int allreduce(double* buf, long count, hipStream_t consumer, hipEvent_t ready = nullptr) {
        if (c->rccl_comm) {   // native RCCL: stream-ordered like the hook below, issued by the library itself
            if (ready) HIP_TRY(hipStreamWaitEvent(consumer, ready, 0));
            return gpx_rccl_allreduce(c, buf, count, consumer);
        }
        if (c->allreduce_stream) {
            if (ready) HIP_TRY(hipStreamWaitEvent(consumer, ready, 0));
            return c->allreduce_stream(buf, count, consumer, c->allreduce_user) ? GPX_ERR_STATE : GPX_OK;
        }
        if (!c->allreduce) return GPX_OK;
        if (ready) HIP_TRY(hipEventSynchronize(ready));
        else HIP_TRY(hipStreamSynchronize(consumer));
        return c->allreduce(buf, count, c->allreduce_user) ? GPX_ERR_STATE : GPX_OK;
    }
    // gpx_potrf_inv writes the lower triangle of W only and the products that read W rely on zeros above it: the
    // zero-fill is needed once per (buffer, m), not per evaluation (39 us each at m = 1024, at the head of both chains)
    int clean_winv(int slot, DevBuf& W, hipStream_t s_) {
        if (c->sparse_side_mm && c->winv_clean_p[slot] == W.p && c->winv_clean_m[slot] == m) return GPX_OK;
        HIP_TRY(hipMemsetAsync(W.p, 0, (size_t)ldm * m * sizeof(double), s_));
        c->winv_clean_p[slot] = W.p;
        c->winv_clean_m[slot] = m;
        return GPX_OK;
    }
    // full symmetric inverse of a factored m x m matrix: Winv = L^-1, Inv = Winv^T Winv
    int inverse(const double* L, const double* Wd, double* Winv, double* Inv, bool need_winv_only = false) {
        int rc;
        HIP_TRY(hipMemsetAsync(Winv, 0, (size_t)ldm * m * sizeof(double), st));
        if ((rc = gpx_trtri(st, m, L, ldm, Wd, Winv, ldm, c->T.p))) return rc;
        if (need_winv_only) return GPX_OK;
        if ((rc = gpx_lauum(st, m, Winv, ldm, Inv, ldm))) return rc;
        hipLaunchKernelGGL(mm_symmetrize_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, Inv, ldm);
        return GPX_OK;
    }
    // Inv (full, symmetric) = Winv^T Winv
    int gram(const double* Winv, double* Inv) {
        int rc;
        if ((rc = gpx_lauum(st, m, Winv, ldm, Inv, ldm))) return rc;
        hipLaunchKernelGGL(mm_symmetrize_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, Inv, ldm);
        return GPX_OK;
    }
    // (F = Kux r of VFE as weighted column sums inside the build kernel - 16-lane DPP sums per tile, LDS accumulator,
    // fixed grid - was measured: the build grows 24.6 -> 34.6 ms and the GEMV it replaces costs 13 ms: break-even, not kept.)
    // Kxu = k(X, U): on the matrix cores (norm expansion of centred inputs) where the dimension allows it,
    // else by direct differences on the vector units
    int build_kxu() {
        int rc;
        if (mfma()) {
            if ((rc = gpx_mfma_prepare(st, c->Xt.p, c->ldx, n, c->Ut.p, c->ldu, m, d, c->cen.p, c->Xa.p, c->Ua.p, c->UaT.p)))
                return rc;
            return gpx_mfma_cov_build(st, c->Xa.p, n, c->Ua.p, m, d, h.sf2, c->Kxu.p, ldk, pan, pst, build_max_grid);
        }
        return gpx_cov_build(st, c->Xt.p, c->ldx, n, c->Ut.p, c->ldu, m, c->dpad, h.sf2, c->Kxu.p, ldn, 0, 0.0);
    }
    // sum_ab W_ab dKxu_ab/dtheta for every kernel hyper in one pass (weights in ta)
    int trace_xu(const TraceArgs& ta, double* out) {
        if (mfma())
            return gpx_mfma_pair_trace(st, c->Xa.p, n, c->Ua.p, c->UaT.p, m, d, h.sf2, ta, c->cpart.p, c->spart.p, c->cvec.p, out, pan, pst);
        return gpx_pair_trace(st, ta, d, c->trace_ws.p, c->trace_ws.cap, out);
    }
    // inputs, Kuu (+jitter on its diagonal) factored into Lu, Kxu
    // overlap: the Lu factor+inverse chain (one workgroup wide for most of its length) runs on the side stream.
    // FITC (!defer_join): V needs W_u right after the build, and a chain launched BESIDE the build stands still until the
    // build has drained (the build's waves hold 500 of the 512 vector registers of every SIMD: the chain's small launches
    // find no room on any CU - 0.45 ms of waiting at n = 2^20, 0.54 ms at the 8-GPU shard size), so the chain is queued
    // FIRST and runs while the host is still launching it; the build follows on the main stream.
    // defer_join (VFE): nothing of phase 1 needs Lu, so the main stream does NOT wait for the chain: it runs beside the
    // HBM-bound GEMV F = Kux r (whose workgroups hold no LDS; the Kxu build's do, and the 150 KB base block of the chain
    // found no CU until the 25 ms build had ended: profiles/r02d_vfe) and phase 2 follows it on the side stream anyway.
    // The chain then takes its own scratch (c->Tc): c->T holds the split-K slabs of the Gram product meanwhile.
    int kernels(double kuu_jitter, bool fused_inverse = false, bool overlap = false, bool defer_join = false,
                bool kuu_inv_behind = false) {
        int rc;
        // the inducing points first: the chain needs nothing of X, and scaling X takes 0.2 ms at n = 2^20 - a head start
        // for the chain before the Kxu build fills every CU (its small launches then find no room until the build ends)
        if ((rc = gpx_scale_inputs(st, c->U.p, m, d, c->dpad, h.sc, c->Ut.p, c->ldu))) return rc;
        if ((rc = gpx_cov_build(st, c->Ut.p, c->ldu, m, c->Ut.p, c->ldu, m, c->dpad, h.sf2, c->Kuu.p, ldm, 1, kuu_jitter)))
            return rc;
        HIP_TRY(hipMemcpyAsync(c->Lu.p, c->Kuu.p, (size_t)ldm * m * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemsetAsync(c->info, 0, 2 * sizeof(int), st));
        if (fused_inverse) {
            hipStream_t sf = st;
            // FITC on a small shard (the build shorter than the chain: the 8-GPU case): the build runs on at most 224
            // workgroups.  At two workgroups per CU they occupy at most 224 of the 256 CUs however the dispatcher places
            // them, so at least 32 CUs stay EMPTY for the chain (a base block needs a whole CU: 150 KB of LDS, every vector
            // register) - the chain and the build then really run side by side, build first in the host's order.
            static const int thr_env = [] { const char* e = getenv("GPX_BUILD_THROTTLE"); return e ? atoi(e) : -1; }();
            const bool throttle = overlap && !defer_join && mfma() && (thr_env >= 0 ? thr_env > 0 : (double)n * (double)m <= 268435456.0);
            if (overlap) {
                sf = c->st2;
                HIP_TRY(hipEventRecord(c->ev[0], st));
                HIP_TRY(hipStreamWaitEvent(sf, c->ev[0], 0));
                if ((rc = gpx_scale_inputs(st, c->X.p, n, d, c->dpad, h.sc, c->Xt.p, c->ldx))) return rc;
                if (defer_join || throttle) {   // VFE: the 25 ms build goes first (the chain's ~70 launches are host-bound)
                    build_max_grid = throttle ? (thr_env > 1 ? thr_env : 224) : 0;
                    rc = build_kxu();
                    build_max_grid = 0;
                    if (rc) return rc;
                    if ((rc = gpx_vec_shift(st, n, c->y.p, h.mean, 1.0, c->r.p))) return rc;
                }
            }
            if ((rc = clean_winv(0, c->Winv_u, sf))) return rc;
            double* tscr = c->T.p;
            if (overlap && defer_join) {
                if ((rc = c->Tc.ensure(gpx_trtri_scratch(m)))) return rc;
                tscr = c->Tc.p;
            }
            if ((rc = gpx_potrf_inv(sf, m, c->Lu.p, ldm, c->Winv_u.p, ldm, tscr, c->info))) return rc;
            if (overlap) {
                if (!defer_join) {
                    HIP_TRY(hipEventRecord(c->ev[1], sf));
                    if (!throttle) {
                        if ((rc = build_kxu())) return rc;
                        if ((rc = gpx_vec_shift(st, n, c->y.p, h.mean, 1.0, c->r.p))) return rc;
                    }
                    HIP_TRY(hipStreamWaitEvent(st, c->ev[1], 0));
                }
                if (kuu_inv_behind) {
                    // Kuu^-1 = W_u^T W_u (FITC gradient) right behind the chain on the side stream, beside the n m^2
                    // products of phase 1, instead of on the main stream at the head of phase 3
                    const hipStream_t keep = st;
                    st = sf;
                    rc = gram(c->Winv_u.p, c->Kuu_inv.p);
                    st = keep;
                    if (rc) return rc;
                }
                return GPX_OK;
            }
        } else if ((rc = gpx_potrf(st, m, c->Lu.p, ldm, c->Wd_u.p, c->info))) {
            return rc;
        }
        if ((rc = gpx_scale_inputs(st, c->X.p, n, d, c->dpad, h.sc, c->Xt.p, c->ldx))) return rc;
        if ((rc = build_kxu())) return rc;
        return gpx_vec_shift(st, n, c->y.p, h.mean, 1.0, c->r.p);
    }
};

// scal layout (device c->scal.p, host c->host_scal)
enum { S_LDU = 0, S_LDA = 1, S_FA = 2, S_TRKS = 3, S_TRAS = 4, S_P0 = 8 /* 8 partials */, S_TR0 = 32, S_TR1 = 32 + GPX_MAX_DIM + 8 };

}  // namespace

int gpx_symmetrize_lower(hipStream_t st, long n, double* A, long lda) {
    if (n <= 0) return GPX_OK;
    hipLaunchKernelGGL(mm_symmetrize_kernel, dim3(nblk(n * n)), dim3(256), 0, st, n, A, lda);
    return GPX_OK;
}

int gpx_sparse_nll_grad(gpx_ctx* c, int model, int cov, const double* hyp, int calc_grad, double jitter,
                        double* nll, double* grad) {
    if (c->n <= 0 || c->m <= 0) return GPX_ERR_STATE;
    if (c->prepared_model == GPX_MODEL_EXACT) c->prepared_model = -1;  // Xt is rescaled below
    const StageTrace tr;
    tr.mark(model == GPX_MODEL_FITC ? "FITC evaluation begins" : "VFE evaluation begins");
    Sparse s;
    s.c = c; s.st = c->st; s.n = c->n; s.m = c->m; s.d = c->d; s.model = model; s.cov = cov;
    int rc = parse_hyp(cov, c->d, hyp, s.h);
    if (rc) return rc;
    if ((rc = s.alloc(calc_grad != 0))) return rc;
    tr.mark("workspaces allocated");
    const HypView& h = s.h;
    const long n = s.n, m = s.m, ldn = s.ldn, ldm = s.ldm, ldk = s.ldk;
    const GemmPanels pnK = s.pn_of(true, true, false, false);   // Gram products: both operands K-contiguous over the n rows
    const int d = s.d, nh = h.nk + 2;
    (void)ldn;
    hipStream_t st = c->st;  // main stream; phase 2 switches to the side stream
    const bool fitc = model == GPX_MODEL_FITC;
    // reduction buffer [F (m, padded to 16) | S (m*m) | tail]: all-reduce #1 takes F, S and the first tail entries,
    // all-reduce #2 of FITC takes S and the tail only (F is not reduced a second time)
    const long mp = (m + 15) / 16 * 16;
    double* redF = c->red.p;
    double* red = redF + mp;         // the m x m slot
    double* redT = red + m * m;
    double* a_vec = c->vm[0].p;
    double* g = c->vn[0].p;   // 1/Gamma
    double* p = c->vn[1].p;   // r/Gamma
    double* Ka = c->vn[2].p;
    double* hs = c->host_scal;
    *nll = INF;
    if (calc_grad) memset(grad, 0, sizeof(double) * nh);  // FITC.cpp:232 / VFE.cpp:194 on failure

    // FITC in WHITENED form.  The reference forms A = sn2 Kuu + Kux Gamma^-1 Kxu (FITC.cpp:217), whose condition number
    // at the benchmark's own conditions is 1.2e9: the data-fit term F^T A^-1 F / sn2 then carries ~3e-9 of rounding from
    // nothing but A's own fp64 entries (an 80-bit A rounded to fp64 and factored by LAPACK is 5.6e-9 off in F^T A^-1 F,
    // tools/study_fitc_terms.py), which is 1e-9 of the NLL.  With V = Kxu Lu^-T (already formed for Gamma):
    //     A = Lu M Lu^T,   M = sn2 I + V^T Gamma^-1 V            (cond 2e4)
    //     F^T A^-1 F = F_v^T M^-1 F_v,  F_v = V^T Gamma^-1 r;    log|A| - log|Kuu| = log|M|  (no cancellation)
    //     a = A^-1 F = Lu^-T M^-1 F_v;  A^-1 = (W_M W_u)^T (W_M W_u)   with W_M = L_M^-1, W_u = Lu^-1
    // Same flops (the Gram product and the GEMV read V instead of Kxu), one extra m x m triangular product for the
    // gradient; the all-reduced partials are [F_v | V^T Gamma^-1 V] instead of [F | S].  GPX_FITC_NO_WHITEN=1 keeps the
    // reference's A (A/B runs).
    const bool wh = fitc && c->fitc_whiten;
    const double* gram_op = wh ? c->V.p : c->Kxu.p;
    // ---- phase 1 ------------------------------------------------------------------------------
    const bool side_mm = c->sparse_side_mm && calc_grad;
    if ((rc = s.kernels(jitter, true, true, !fitc, fitc && side_mm))) return rc;  // Lu diag + Winv_u = Lu^-1
    HIP_TRY(hipMemsetAsync(redT, 0, 16 * sizeof(double), st));
    if (mp > m) HIP_TRY(hipMemsetAsync(redF + m, 0, (mp - m) * sizeof(double), st));   // the padding travels with all-reduce #1
    if (fitc) {
        // V = Kxu Lu^-T (FITC.cpp:213-215 via the factor: diag(Kxu Kuu^-1 Kux) = rowsum(V o V))
        if ((rc = s.v_and_gamma(c->Winv_u.p, c->r.p, h.sn2, h.sf2, g, p))) return rc;
        s.finish_partials(n, redT, 3);  // sum log Gamma, r^T Gamma^-1 r, n_local
        // (this HBM-bound GEMV on the side stream, beside the matrix-pipe-bound Gram product below, was measured: the Gram
        // product loses more than the GEMV's 1.65 ms, 113.4 -> 114.9 ms per evaluation)
        // whitened form (default): F_v = V^T Gamma^-1 r = Lu^-1 F;  else F = Kux Gamma^-1 r (FITC.cpp:218)
        if ((rc = gpx_gemv_t_split(st, n, m, 1.0, gram_op, ldk, p, 0.0, redF, s.pan, s.pst, c->gemv_ws.p))) return rc;
    } else {
        hipLaunchKernelGGL(vfe_rstats_kernel, dim3(nblk(n)), dim3(256), 0, st, n, c->r.p, s.part());
        s.finish_partials(n, redT, 3);  // r^T r, -, n_local
        if ((rc = gpx_gemv_t_split(st, n, m, 1.0, c->Kxu.p, ldk, c->r.p, 0.0, redF, s.pan, s.pst, c->gemv_ws.p))) return rc;  // F = Kux r
    }
    // S = Kux [Gamma^-1] Kxu, lower tiles, split-K over the training rows (FITC.cpp:217, VFE.cpp:173)
    // (F = Kux [Gamma^-1] r as an operand dot of THIS product - from the A fragments of its first tile column, partial sums
    // per split-K slab, as Kxu a leaves the product Kxu P below - was written and measured: the tiles that carry the dot
    // fall behind the other tiles of their slab and the slab loses its shared L2 reads; the Gram product grows by more
    // than the GEMV it replaces, FITC 113.25 -> 113.6 ms, VFE 846 -> 866 ms, same box; not kept.)
    if ((rc = gpx_gemm_splitk(st, 1, 1, m, m, n, 1.0, gram_op, ldk, gram_op, ldk, 0.0, red, m, GEMM_LOWER, c->T.p, c->T.cap,
                              fitc ? g : nullptr, &pnK)))
        return rc;
    HIP_TRY(hipEventRecord(c->ev[0], st));
    if (fitc && calc_grad) {
        // B = Kxu Kuu^-1 = V Lu^-1 (FITC.cpp:214) does not depend on A: it keeps the GPU full while the
        // all-reduce and the (narrow) factor chain of A run beside it
        const GemmPanels pn = s.pn_of(true, false, true, false);
        if ((rc = gpx_gemm(st, 0, 1, n, m, m, 1.0, c->V.p, ldk, c->Winv_u.p, ldm, 0.0, c->Bm.p, ldk, GEMM_KLO_N, &pn))) return rc;
    }
    tr.mark("phase 1 enqueued");
    if ((rc = s.allreduce(redF, mp + m * m + 16, c->st2, c->ev[0]))) return rc;
    tr.mark("all-reduce #1 returned to the host");

    // ---- phase 2 (side stream) ------------------------------------------------------------------
    // A = sn2*Kuu + S (FITC.cpp:217, VFE.cpp:174), factor, a = A^-1 F
    const hipStream_t st_main = st;
    st = c->st2;
    s.st = st;
    HIP_TRY(hipStreamWaitEvent(st, c->ev[0], 0));
    if (wh)   // M = sn2 I + V^T Gamma^-1 V
        hipLaunchKernelGGL(mm_combine_lower_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, 0.0, c->Kuu.p, ldm, 1.0, red, m,
                           h.sn2, c->LA.p, ldm);
    else
        hipLaunchKernelGGL(mm_combine_lower_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, h.sn2, c->Kuu.p, ldm, 1.0, red, m,
                           0.0, c->LA.p, ldm);
    // (straight into the buffer the factorisation works in: the 8 MB copy from a separate A took 146 us of this
    // latency-bound path beside the B product at the 8-GPU shard size)
    if ((rc = s.clean_winv(1, c->Winv_A, st))) return rc;
    if ((rc = gpx_potrf_inv(st, m, c->LA.p, ldm, c->Winv_A.p, ldm, c->T.p, c->info + 1))) return rc;
    // z = W_A F (whitened: W_M F_v);  F^T A^-1 F = z^T z;  a = A^-1 F = W_A^T z (whitened: W_u^T W_M^T z)
    if ((rc = gpx_gemv_n(st, m, m, 1.0, c->Winv_A.p, ldm, redF, 0.0, c->vm[1].p, 1))) return rc;
    if (wh) {
        if ((rc = gpx_gemv_t(st, m, m, 1.0, c->Winv_A.p, ldm, c->vm[1].p, 0.0, c->vm[2].p, 1))) return rc;
        if ((rc = gpx_gemv_t(st, m, m, 1.0, c->Winv_u.p, ldm, c->vm[2].p, 0.0, a_vec, 1))) return rc;
    } else if ((rc = gpx_gemv_t(st, m, m, 1.0, c->Winv_A.p, ldm, c->vm[1].p, 0.0, a_vec, 1))) {
        return rc;
    }
    if ((rc = gpx_logdiag_sum(st, m, c->Lu.p, ldm, c->scal.p + S_LDU))) return rc;
    if ((rc = gpx_logdiag_sum(st, m, c->LA.p, ldm, c->scal.p + S_LDA))) return rc;
    hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(256), 0, st, m, c->vm[1].p, c->vm[1].p, c->scal.p + S_FA);
    if (!fitc) {
        // VFE needs Kuu^-1 for the trace term already in the NLL (VFE.cpp:181,188)
        hipLaunchKernelGGL(mm_symmetrize_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, red, m);
        if ((rc = s.gram(c->Winv_u.p, c->Kuu_inv.p))) return rc;
        hipLaunchKernelGGL(mm_inner_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, c->Kuu_inv.p, ldm, red, m, s.part());
        s.finish_partials(m * m, c->scal.p + S_TRKS, 1);
    }
    HIP_TRY(hipMemcpyAsync(hs, c->scal.p, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hs + 8, redT, 8 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hs + 16, c->info, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    tr.mark("phase 2 enqueued");
    // A^-1 for the gradient: W_A = W_M W_u (whitened; lower x lower: B(n,k) = W_u[k + n ldm] is zero for k < n), A^-1 = W_A^T W_A.
    auto a_inverse = [&]() -> int {
        if (wh) {
            int r2 = gpx_gemm(s.st, 0, 1, m, m, m, 1.0, c->Winv_A.p, ldm, c->Winv_u.p, ldm, 0.0, c->Am.p, ldm, GEMM_KLO_N);
            return r2 ? r2 : s.gram(c->Am.p, c->A_inv.p);
        }
        return s.gram(c->Winv_A.p, c->A_inv.p);
    };
    if (side_mm) {
        // The host only needs the scalars: it waits for an event behind their copies, and the m x m products of A^-1 go on
        // behind that event on the side stream - they run while the host enqueues phase 3 and the main stream finishes B,
        // and the main stream waits for them through a second event (they were ~0.3 ms at the head of phase 3, exposed at
        // the 8-GPU shard size).  If the factorisation failed they work on garbage and nobody reads the result.
        HIP_TRY(hipEventRecord(c->ev[1], st));
        if ((rc = a_inverse())) return rc;
        HIP_TRY(hipEventRecord(c->ev[0], st));
        HIP_TRY(hipEventSynchronize(c->ev[1]));
        tr.mark("phase 2 scalars on the host");
    } else {
        HIP_TRY(hipStreamSynchronize(st));  // the side stream is drained: phase 3 needs no event
        tr.mark("phase 2 done on the device (side stream drained)");
    }
    st = st_main;
    s.st = st;
    int info[2];
    memcpy(info, hs + 16, sizeof(info));
    const double ntot = hs[8 + 2];
    const double ldU = wh ? 0.0 : 2 * hs[S_LDU], ldA = 2 * hs[S_LDA], Fa = hs[S_FA];   // whitened: ldA = log|M| = log|A| - log|Kuu|
    double v, f_tr = 0.0;
    {
        static const bool dbg = getenv("GPX_DEBUG_TERMS") != nullptr;   // the terms of the NLL, for tools/study_fitc_terms.py
        if (dbg)
            fprintf(stderr, "gpx terms: ldA %.17g ldU %.17g slg_or_rr %.17g rGr %.17g Fa %.17g whitened %d\n", 2 * hs[S_LDA],
                    2 * hs[S_LDU], hs[8 + 0], hs[8 + 1], Fa, (int)wh);
    }
    if (fitc) {
        const double data_fit = (hs[8 + 1] - Fa) / h.sn2;                                   // FITC.cpp:221-223
        const double mc = ldA - ldU + hs[8 + 0] + (ntot - (double)m) * log(h.sn2);           // FITC.cpp:226
        v = 0.5 * (data_fit + mc + ntot * log(2 * M_PI));
        if (!(std::isfinite(data_fit) && std::isfinite(mc))) v = INF;
    } else {
        f_tr = 0.5 * (ntot * h.sf2 - hs[S_TRKS]) / h.sn2;                                    // VFE.cpp:188
        v = 0.5 * ntot * log(2 * M_PI) + 0.5 * (ldA - ldU + (ntot - (double)m) * log(h.sn2)) +
            0.5 * (hs[8 + 0] - Fa) / h.sn2 + f_tr;                                           // VFE.cpp:185-189
    }
    if (info[0] != 0 || info[1] != 0 || !std::isfinite(v)) {
        HIP_TRY(hipStreamSynchronize(st));  // B may still be running
        if (side_mm) HIP_TRY(hipStreamSynchronize(c->st2));
        return GPX_OK;                      // NLL = INF, g = 0
    }
    *nll = v;
    if (!calc_grad) return GPX_OK;

    // ---- phase 3 ------------------------------------------------------------------------------
    if (side_mm) HIP_TRY(hipStreamWaitEvent(st, c->ev[0], 0));   // A^-1 (and Kuu^-1 before it) from the side stream
    else if ((rc = a_inverse())) return rc;
    TraceArgs ta = {};
    ta.Xt1 = c->Xt.p; ta.ld1 = c->ldx; ta.p = n;
    ta.Xt2 = c->Ut.p; ta.ld2 = c->ldu; ta.q = m;
    ta.dpad = c->dpad;
    ta.sf2 = h.sf2;
    ta.v = a_vec;
    HIP_TRY(hipMemsetAsync(redT, 0, (GPX_MAX_DIM + 32) * sizeof(double), st));
    if (fitc) {
        if (!side_mm && (rc = s.gram(c->Winv_u.p, c->Kuu_inv.p))) return rc;
        // C = Gamma^-1 Kxu A^-1 (FITC.cpp:262) and B = Kxu Kuu^-1 = V Lu^-1 (FITC.cpp:214)
        double* Cm = c->Cm.p;  // Cu = Kxu A^-1; C = diag(1/Gamma) Cu enters through per-row coefficients
        GemmRowdot rd = {};   // rowsum(Cu o Kxu) and Kxu a leave the GEMM with Cu
        rd.E = c->Kxu.p; rd.lde = ldk; rd.evec = a_vec; rd.R1 = c->rowp1.p; rd.R2 = c->rowp2.p; rd.ldr = s.ldn;
        const GemmPanels pnC = s.pn_of(true, false, true, true);
        if ((rc = gpx_gemm_rowdot(st, 0, 0, n, m, m, 1.0, c->Kxu.p, ldk, c->A_inv.p, ldm, Cm, ldk, 0, rd, &pnC))) return rc;
        double* c2coef = c->vn[3].p;
        double* ucoef = c->vn[4].p;
        double* wv = c->vn[5].p;
        hipLaunchKernelGGL(fitc_rowstats_kernel, dim3(nblk(n)), dim3(256), 0, st, n, (int)s.nparts(), c->rowp1.p, c->rowp2.p, ldn,
                           g, p, h.sn2, c2coef, ucoef, wv, s.part());
        s.finish_partials(n, redT + GPX_MAX_DIM + 8, 4);  // sum(g-c1), sum pq^2, sum pq, sum w
        // B^T diag(w) B: w (signed) is the K-dimension scale of the lower-tile split-K product.  It goes BEFORE the trace:
        // with a stream-ordered collective (RCCL, stream hook) the m x m part of all-reduce #2 then runs on the side stream
        // beside the trace kernel, which needs nothing of it, and only the d + 6 trace sums are reduced behind the trace
        // (#2b, a latency-sized collective) - the 8 MB collective is off the critical path.  The host-synchronous hook and
        // GPX_FITC_SINGLE_AR2=1 keep ONE all-reduce #2 of [S | tail] behind the trace.
        if ((rc = gpx_gemm_splitk(st, 1, 1, m, m, n, 1.0, c->Bm.p, ldk, c->Bm.p, ldk, 0.0, red, m, GEMM_LOWER, c->T.p, c->T.cap,
                                  wv, &pnK)))
            return rc;
        const bool split_ar2 = c->fitc_split_ar2 && (c->rccl_comm || c->allreduce_stream);
        if (split_ar2) {
            HIP_TRY(hipEventRecord(c->ev[1], st));
            if ((rc = s.allreduce(red, m * m, c->st2, c->ev[1]))) return rc;
            HIP_TRY(hipEventRecord(c->ev[0], c->st2));
        }
        // half of W_xu = C - (1/sn2) pq a^T - (1/sn2) diag(w) B, contracted with every dKxu at once
        ta.M1 = Cm; ta.ldm1 = ldk; ta.c1 = g;
        ta.M2 = c->Bm.p; ta.ldm2 = ldk; ta.c2 = c2coef;
        ta.u = ucoef;
        if ((rc = s.trace_xu(ta, redT))) return rc;
        tr.mark("phase 3 enqueued");
        if (split_ar2) {
            if ((rc = s.allreduce(redT, GPX_MAX_DIM + 32, st))) return rc;
            HIP_TRY(hipStreamWaitEvent(st, c->ev[0], 0));   // the reduced m x m part
        } else if ((rc = s.allreduce(red, m * m + GPX_MAX_DIM + 32, st))) {
            return rc;
        }
        tr.mark("all-reduce #2 returned to the host");
        hipLaunchKernelGGL(mm_symmetrize_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, red, m);
        // W_uu = sn2 A^-1 - Kuu^-1 + a a^T + B^T diag(w) B / sn2
        hipLaunchKernelGGL(mm_combine_full_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, h.sn2, c->A_inv.p, ldm, -1.0,
                           c->Kuu_inv.p, ldm, 1.0, a_vec, 1.0 / h.sn2, red, m, c->Wuu.p, ldm);
    } else {
        // P = A^-1 - Kuu^-1/sn2 ; W_xu = Kxu P - alpha a^T
        hipLaunchKernelGGL(mm_combine_full_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, 1.0, c->A_inv.p, ldm,
                           -1.0 / h.sn2, c->Kuu_inv.p, ldm, 0.0, (const double*)nullptr, 0.0, (const double*)nullptr, 0L,
                           c->Wuu.p, ldm);
        // Kxu a (VFE.cpp:176) leaves this product with Kxu P: the operand dot of the GEMM, no pass of its own over Kxu
        // (the GEMV took 13 ms of 841 at n = 2^22, m = 2048)
        GemmRowdot rd = {};
        rd.evec = a_vec; rd.R2 = Ka;
        const GemmPanels pnC = s.pn_of(true, false, true, false);
        if ((rc = gpx_gemm_rowdot(st, 0, 0, n, m, m, 1.0, c->Kxu.p, ldk, c->Wuu.p, ldm, c->Cm.p, ldk, 0, rd, &pnC))) return rc;
        double* nalpha = c->vn[3].p;
        hipLaunchKernelGGL(vfe_alpha_kernel, dim3(nblk(n)), dim3(256), 0, st, n, c->r.p, Ka, h.sn2, nalpha, s.part());
        s.finish_partials(n, redT + GPX_MAX_DIM + 8, 2);  // sum alpha^2, sum alpha
        ta.M1 = c->Cm.p; ta.ldm1 = ldk;
        ta.u = nalpha;
        if ((rc = s.trace_xu(ta, redT))) return rc;
        // tr(A^-1 S) and G = Kuu^-1 S Kuu^-1 use the reduced S (still in red[0..m*m), symmetrised)
        hipLaunchKernelGGL(mm_inner_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, c->A_inv.p, ldm, red, m, s.part());
        s.finish_partials(m * m, c->scal.p + S_TRAS, 1);
        if ((rc = gpx_gemm(st, 0, 0, m, m, m, 1.0, c->Kuu_inv.p, ldm, red, m, 0.0, c->Am.p, ldm, 0))) return rc;
        if ((rc = gpx_gemm(st, 0, 0, m, m, m, 1.0, c->Am.p, ldm, c->Kuu_inv.p, ldm, 0.0, c->LA.p, ldm, 0))) return rc;
        tr.mark("phase 3 enqueued");
        if ((rc = s.allreduce(redT, GPX_MAX_DIM + 32, st))) return rc;
        tr.mark("all-reduce #2 returned to the host");
        // W_uu = 0.5 [a a^T + sn2 A^-1 - Kuu^-1 + G/sn2]
        hipLaunchKernelGGL(mm_combine_full_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, 0.5 * h.sn2, c->A_inv.p, ldm, -0.5,
                           c->Kuu_inv.p, ldm, 0.5, a_vec, 0.5 / h.sn2, c->LA.p, ldm, c->Wuu.p, ldm);
    }
    // ---- phase 4: sum W_uu o dKuu_i (dKuu of the jittered Kuu: FITC.cpp:271, VFE.cpp:215) --------
    TraceArgs tu = {};
    tu.Xt1 = c->Ut.p; tu.ld1 = c->ldu; tu.p = m;
    tu.Xt2 = c->Ut.p; tu.ld2 = c->ldu; tu.q = m;
    tu.dpad = c->dpad;
    tu.sf2 = h.sf2;
    tu.M1 = c->Wuu.p; tu.ldm1 = ldm;
    double* tuu = c->scal.p + S_TR0;
    if ((rc = gpx_pair_trace(st, tu, d, c->trace_ws.p, c->trace_ws.cap, tuu))) return rc;
    HIP_TRY(hipMemcpyAsync(hs + 32, tuu, (d + 2) * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hs + 32 + GPX_MAX_DIM + 8, redT, (GPX_MAX_DIM + 32) * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(hs + 4, c->scal.p + S_TRAS, sizeof(double), hipMemcpyDeviceToHost, st));
    tr.mark("phase 4 enqueued");
    HIP_TRY(hipStreamSynchronize(st));
    tr.mark("evaluation done on the device");
    const double* Suu = hs + 32;                         // [d | sum W K | tr W]
    const double* Sxu = hs + 32 + GPX_MAX_DIM + 8;       // [d | sum W K | -]
    const double* tail = Sxu + GPX_MAX_DIM + 8;
    const double wx = fitc ? 2.0 : 1.0;   // FITC traced half of W_xu
    const double fac = fitc ? 0.5 : 1.0;  // FITC: g = 0.5 * (...)
    auto kern = [&](int i) { return fac * (Suu[i] + wx * Sxu[i]); };
    if (cov == GPX_COV_SE_ARD) {
        for (int i = 0; i < d; ++i) grad[i] = kern(i);
    } else {
        double t = 0;
        for (int i = 0; i < d; ++i) t += kern(i);
        grad[0] = t;
    }
    double gsf = fac * (2 * Suu[d] + 2 * jitter * Suu[d + 1] + wx * 2 * Sxu[d]);
    if (fitc) {
        gsf += 0.5 * (2 * h.sf2 * tail[3] / h.sn2);              // (1/sn2) w . dKnn  (CovSEard.cpp:90-96)
        grad[nh - 2] = tail[0] - tail[1] / h.sn2;                 // FITC.cpp:311-313
        grad[nh - 1] = -1 * tail[2] / h.sn2;                      // FITC.cpp:314
    } else {
        gsf += 0.5 * ntot * 2 * h.sf2 / h.sn2;                    // VFE.cpp:232-234
        grad[nh - 2] = -1 * h.sn2 * tail[0] + (ntot - hs[4]) - 2 * f_tr;  // VFE.cpp:239
        grad[nh - 1] = -1 * tail[1];                               // VFE.cpp:240
    }
    grad[h.nk - 1] = gsf;
    return GPX_OK;
}

// FITC::_setK (FITC.cpp:165-200) and VFE::_setK (VFE.cpp:132-160)
int gpx_sparse_prepare(gpx_ctx* c, int model, int cov, const double* hyp, double* jitter) {
    if (c->n <= 0 || c->m <= 0) return GPX_ERR_STATE;
    c->prepared_model = -1;
    Sparse s;
    s.c = c; s.st = c->st; s.n = c->n; s.m = c->m; s.d = c->d; s.model = model; s.cov = cov;
    int rc = parse_hyp(cov, c->d, hyp, s.h);
    if (rc) return rc;
    if ((rc = s.alloc(false))) return rc;
    const HypView& h = s.h;
    const long n = s.n, m = s.m, ldm = s.ldm, ldk = s.ldk;
    const GemmPanels pnK = s.pn_of(true, true, false, false);
    hipStream_t st = c->st;
    const bool fitc = model == GPX_MODEL_FITC;
    const long mp = (m + 15) / 16 * 16;   // [F | S] as in gpx_sparse_nll_grad
    double* redF = c->red.p;
    double* red = redF + mp;
    double* a_vec = c->vm[0].p;
    double cum = 0.0;  // VFE: jitter accumulated on Kuu's diagonal
    for (int iter = 0; iter < 200; ++iter) {
        const bool first = iter == 0;
        double kuu_add;
        if (fitc) {
            if (!first) *jitter *= 2;                      // FITC.cpp:186
            kuu_add = *jitter;                             // _u_solver->decomp(Kuu + jitter*I)
        } else {
            if (!first) cum += *jitter;                    // VFE.cpp:149 (cumulative)
            kuu_add = cum;
        }
        if ((rc = s.kernels(kuu_add))) return rc;
        if (fitc) {
            if ((rc = s.inverse(c->Lu.p, c->Wd_u.p, c->Winv_u.p, nullptr, true))) return rc;
            if ((rc = s.v_and_gamma(c->Winv_u.p, c->r.p, h.sn2, h.sf2, c->vn[0].p, c->vn[1].p))) return rc;
            if ((rc = gpx_gemv_t(st, n, m, 1.0, c->Kxu.p, ldk, c->vn[1].p, 0.0, redF, 0, s.pan, s.pst))) return rc;
            if ((rc = gpx_gemm_splitk(st, 1, 1, m, m, n, 1.0, c->Kxu.p, ldk, c->Kxu.p, ldk, 0.0, red, m, GEMM_LOWER, c->T.p,
                                      c->T.cap, c->vn[0].p, &pnK)))
                return rc;
        } else {
            if ((rc = gpx_gemv_t(st, n, m, 1.0, c->Kxu.p, ldk, c->r.p, 0.0, redF, 0, s.pan, s.pst))) return rc;
            if ((rc = gpx_gemm_splitk(st, 1, 1, m, m, n, 1.0, c->Kxu.p, ldk, c->Kxu.p, ldk, 0.0, red, m, GEMM_LOWER, c->T.p,
                                      c->T.cap, nullptr, &pnK)))
                return rc;
        }
        if (mp > m) HIP_TRY(hipMemsetAsync(redF + m, 0, (mp - m) * sizeof(double), st));
        if ((rc = s.allreduce(redF, mp + m * m, st))) return rc;
        if (fitc) {
            // A = sn2*Kuu(no jitter) + S, and A + jitter*I inside the retry loop (FITC.cpp:180,195-196)
            const double dadd = -h.sn2 * kuu_add + (first ? 0.0 : *jitter);
            hipLaunchKernelGGL(mm_combine_lower_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, h.sn2, c->Kuu.p, ldm, 1.0,
                               red, m, dadd, c->LA.p, ldm);
        } else {
            // A = Kuu + Kux Kxu / sn2 (VFE.cpp:142,150)
            hipLaunchKernelGGL(mm_combine_lower_kernel, dim3(nblk(m * m)), dim3(256), 0, st, m, 1.0, c->Kuu.p, ldm,
                               1.0 / h.sn2, red, m, 0.0, c->LA.p, ldm);
        }
        if ((rc = gpx_potrf(st, m, c->LA.p, ldm, c->Wd_A.p, c->info + 1))) return rc;
        int info[2];
        HIP_TRY(hipMemcpyAsync(info, c->info, sizeof(info), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (!fitc && !first) *jitter *= 2;                 // VFE.cpp:156
        if (info[0] == 0 && info[1] == 0) {
            // _alpha = A^-1 Kux Gamma^-1 r (FITC.cpp:199) / A^-1 Kux r / sn2 (VFE.cpp:159)
            HIP_TRY(hipMemcpyAsync(a_vec, redF, m * sizeof(double), hipMemcpyDeviceToDevice, st));
            if ((rc = gpx_trsv(st, m, c->LA.p, ldm, c->Wd_A.p, a_vec, 0))) return rc;
            if ((rc = gpx_trsv(st, m, c->LA.p, ldm, c->Wd_A.p, a_vec, 1))) return rc;
            if (!fitc) {
                if ((rc = gpx_vec_shift(st, m, a_vec, 0.0, 1.0 / h.sn2, c->vm[1].p))) return rc;
                HIP_TRY(hipMemcpyAsync(a_vec, c->vm[1].p, m * sizeof(double), hipMemcpyDeviceToDevice, st));
            }
            // private copies: later objective evaluations reuse Lu / LA / Ut as scratch
            const long mm = ldm * m, wdn = 64L * ((m + 63) / 64 * 64), un = (long)c->dpad * c->ldu;
            if ((rc = c->pLu.ensure(mm)) || (rc = c->pLA.ensure(mm)) || (rc = c->pWd_u.ensure(wdn)) ||
                (rc = c->pWd_A.ensure(wdn)) || (rc = c->palpha.ensure(m)) || (rc = c->pUt.ensure(un)))
                return rc;
            HIP_TRY(hipMemcpyAsync(c->pLu.p, c->Lu.p, mm * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(c->pLA.p, c->LA.p, mm * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(c->pWd_u.p, c->Wd_u.p, wdn * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(c->pWd_A.p, c->Wd_A.p, wdn * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(c->palpha.p, a_vec, m * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(c->pUt.p, c->Ut.p, un * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            c->prepared_model = model;
            c->prepared_cov = cov;
            c->prepared_hyp.assign(hyp, hyp + h.nk + 2);
            c->prepared_jitter = *jitter;
            return GPX_OK;
        }
    }
    return GPX_ERR_STATE;
}

int gpx_sparse_predict(gpx_ctx* c, const double* Xs, long t, double* mu, double* s2, double* gmu, double* gs2) {
    const long m = c->m;
    const int d = c->d;
    const bool fitc = c->prepared_model == GPX_MODEL_FITC;
    HypView h;
    int rc = parse_hyp(c->prepared_cov, d, c->prepared_hyp.data(), h);
    if (rc) return rc;
    hipStream_t st = c->st;
    const bool grad = gmu || gs2;
    const long chunk = grad ? 2048 : 16384;
    const long ldm = gpx_ld(m), ldt = gpx_ld(chunk);
    if ((rc = c->pred_K.ensure(3 * ldm * chunk)) || (rc = c->pred_xs.ensure((long)d * chunk)) ||
        (rc = c->pred_xst.ensure((long)c->dpad * ldt)) || (rc = c->pred_out.ensure(5 * chunk)))
        return rc;
    if (grad && (rc = c->pred_g.ensure(2L * d * chunk))) return rc;
    double* K0 = c->pred_K.p;               // K_star^T, kept for the gradients
    double* K1 = c->pred_K.p + ldm * chunk;
    double* K2 = c->pred_K.p + 2 * ldm * chunk;
    double* d_ksa = c->pred_out.p;
    double* d_vu = d_ksa + chunk;
    double* d_vA = d_vu + chunk;
    double* d_mu = d_vA + chunk;
    double* d_s2 = d_mu + chunk;
    for (long t0 = 0; t0 < t; t0 += chunk) {
        const long tc = t - t0 < chunk ? t - t0 : chunk;
        HIP_TRY(hipMemcpyAsync(c->pred_xs.p, Xs + t0 * d, (size_t)tc * d * sizeof(double), hipMemcpyHostToDevice, st));
        if ((rc = gpx_scale_inputs(st, c->pred_xs.p, tc, d, c->dpad, h.sc, c->pred_xst.p, ldt))) return rc;
        // K_star^T = k(U, Xs): m x tc   (FITC.cpp:114, VFE.cpp:116)
        if ((rc = gpx_cov_build(st, c->pUt.p, c->ldu, m, c->pred_xst.p, ldt, tc, c->dpad, h.sf2, K0, ldm, 0, 0.0))) return rc;
        if ((rc = gpx_gemv_t(st, m, tc, 1.0, K0, ldm, c->palpha.p, 0.0, d_ksa))) return rc;
        if (s2 || gs2) {
            // k*(Kuu^-1 - c A^-1)k*^T = |Lu^-1 k*|^2 - c |LA^-1 k*|^2   (FITC.cpp:115-117, VFE.cpp:129-130)
            HIP_TRY(hipMemcpyAsync(K1, K0, (size_t)ldm * tc * sizeof(double), hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(K2, K0, (size_t)ldm * tc * sizeof(double), hipMemcpyDeviceToDevice, st));
            if ((rc = gpx_trsm_left(st, m, tc, c->pLu.p, ldm, c->pWd_u.p, K1, ldm, 0))) return rc;
            if ((rc = gpx_trsm_left(st, m, tc, c->pLA.p, ldm, c->pWd_A.p, K2, ldm, 0))) return rc;
            if ((rc = gpx_colsumsq(st, m, tc, K1, ldm, d_vu))) return rc;
            if ((rc = gpx_colsumsq(st, m, tc, K2, ldm, d_vA))) return rc;
        }
        hipLaunchKernelGGL(predict_sparse_finish_kernel, dim3(nblk(tc)), dim3(256), 0, st, tc, d_ksa, d_vu, d_vA, h.mean,
                           h.sf2, h.sn2, fitc ? 1 : 0, mu ? d_mu : nullptr, s2 ? d_s2 : nullptr);
        if (grad) {
            if (gs2) {  // KinvK = Kuu^-1 k* - sn2 A^-1 k*   (FITC.cpp:115, 127)
                if ((rc = gpx_trsm_left(st, m, tc, c->pLu.p, ldm, c->pWd_u.p, K1, ldm, 1))) return rc;
                if ((rc = gpx_trsm_left(st, m, tc, c->pLA.p, ldm, c->pWd_A.p, K2, ldm, 1))) return rc;
            }
            double* d_gmu = c->pred_g.p;
            double* d_gs2 = c->pred_g.p + (long)d * chunk;
            if ((rc = gpx_predict_grad(st, m, tc, d, c->pUt.p, c->ldu, c->pred_xst.p, ldt, h.sc, K0, ldm, c->palpha.p, K1, K2, ldm,
                                       fitc ? h.sn2 : 1.0, gmu ? d_gmu : nullptr, gs2 ? d_gs2 : nullptr)))
                return rc;
            if (gmu) HIP_TRY(hipMemcpyAsync(gmu + t0 * d, d_gmu, (size_t)tc * d * sizeof(double), hipMemcpyDeviceToHost, st));
            if (gs2) HIP_TRY(hipMemcpyAsync(gs2 + t0 * d, d_gs2, (size_t)tc * d * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        if (mu) HIP_TRY(hipMemcpyAsync(mu + t0, d_mu, tc * sizeof(double), hipMemcpyDeviceToHost, st));
        if (s2) HIP_TRY(hipMemcpyAsync(s2 + t0, d_s2, tc * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return GPX_OK;
}
