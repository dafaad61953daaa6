// kernels.h — internal interfaces between the HIP translation units of libgpx.so.
#pragma once
#include "gpx_internal.h"

// ---- pairwise.hip ---------------------------------------------------------------------------
struct ScaleArgs {
    double s[GPX_MAX_DIM];  // per-dimension factor
    int divide;             // 0: x * s_i (SE-ARD, s_i = exp(-hyp_i)); 1: x / s_i (SE-iso, s = exp(hyp_0))
};
struct TraceArgs {
    const double* Xt1; long ld1; long p;   // scaled inputs, dimension-major [dpad][ld]
    const double* Xt2; long ld2; long q;
    int dpad, dim0;
    double sf2;
    const double* M1; long ldm1; const double* c1;  // w_ab  = c1_a * M1_ab   (c1 NULL -> 1)
    const double* M2; long ldm2; const double* c2;  //       + c2_a * M2_ab   (M2 NULL -> absent)
    const double* u; const double* v;               //       + u_a * v_b      (u NULL -> absent)
    int sym;          // p == q, M1 symmetric and valid in its lower triangle: visit tiles a >= b only
    double* partial;  // filled in by gpx_pair_trace
    int ntiles, tiles_p;
    int vec;          // M1 / M2 allow 16-B vector loads (set by gpx_pair_trace)
};
int gpx_scale_inputs(hipStream_t st, const double* X, long n, int d, int dpad, const ScaleArgs& sc,
                     double* Xt, long ldx);
int gpx_cov_build(hipStream_t st, const double* Xt1, long ld1, long p, const double* Xt2, long ld2, long q,
                  int dpad, double sf2, double* out, long ldo, int lower, double diag_add);
// one materialised dK (Cov::dk_dhyp): mode 0 K o ((x1_dim - x2_dim)/l)^2, 1 K o r2 (iso), 2: 2 K; scaled inputs as for cov_build
int gpx_dk_dhyp(hipStream_t st, long p, long q, int mode, int dim, int d, const double* Xt1, long ld1, const double* Xt2, long ld2,
                const double* K, long ldk, double* out, long ldo);
// out[0..d-1] = S_i, out[d] = sum w K, out[d+1] = tr(w); partial_ws >= 2048*66 doubles
int gpx_pair_trace(hipStream_t st, TraceArgs g, int d, double* partial_ws, long ws_doubles, double* out);
#define GPX_TRACE_WS (2048L * 66)
// HIP-event timing of cov_build (cls 0) / pair_trace (cls 1) launches that move >= min_bytes algorithmic bytes
int gpx_pair_prof_begin(double min_bytes, int max_launches);
int gpx_pair_prof_end(int cls, double* total_ms, double* total_bytes, long* launches);
long gpx_pair_prof_open(hipStream_t st, int cls, double bytes);   // -1: not timed
void gpx_pair_prof_close(hipStream_t st, long slot);

// ---- pairwise_mfma.hip: the n x m kernel matrix and its gradient trace on the matrix cores ------------
bool gpx_mfma_pair_supported(int d);   // d <= 62
long gpx_mfma_ldxa(long n);            // row stride of Xa  [dk][ldxa]
long gpx_mfma_ldua(long m);            // row stride of Ua  [dk][ldua]; UaT is [ldua][n2]
int gpx_mfma_dk(int d);
int gpx_mfma_n2(int d);
int gpx_mfma_trace_grid(long n, int d, long m, bool two_weight_matrices);
// centred, augmented operands from the scaled dimension-major inputs; Xa may be NULL (U side only)
int gpx_mfma_prepare(hipStream_t st, const double* Xt, long ldx, long n, const double* Ut, long ldu, long m, int d,
                     double* cen, double* Xa, double* Ua, double* UaT);
// pan / pst: row-panel layout of out and of the weight matrices M1 / M2 (GemmPanels in gpx_internal.h; 0 = plain)
int gpx_mfma_cov_build(hipStream_t st, const double* Xa, long n, const double* Ua, long m, int d, double sf2, double* out,
                       long ldo, long pan = 0, long pst = 0, int max_grid = 0);   // max_grid > 0: at most that many workgroups
// weights from w (M1/c1, M2/c2, u/v); cpart: grid*ldua doubles, spart: grid*(n2+1), cvec: ldua
int gpx_mfma_pair_trace(hipStream_t st, const double* Xa, long n, const double* Ua, const double* UaT, long m, int d, double sf2,
                        const TraceArgs& w, double* cpart, double* spart, double* cvec, double* out, long pan = 0, long pst = 0);

// k(X, X) + diag_add I on and below the diagonal (whole diagonal tiles) on the matrix cores, exact sf2 + diag_add diagonal
int gpx_mfma_cov_sym_build(hipStream_t st, const double* Xa, const double* Ua, long n, int d, double sf2, double diag_add,
                           double* out, long ldo);
// the exact GP's trace on the matrix cores: symmetric weights w = M1 + u v^T, M1 valid in its lower triangle, both point
// sets = the training set (Ua / UaT prepared from the same points as Xa); d <= 30; spart: grid * (n2 + 2) doubles
bool gpx_mfma_sym_trace_supported(int d);
int gpx_mfma_sym_trace_grid(long n, int d);
int gpx_mfma_sym_trace(hipStream_t st, const double* Xa, const double* Ua, const double* UaT, long n, int d, double sf2,
                       const double* M1, long ldm1, const double* u, const double* v, double* spart, double* out);

// ---- chol.hip -------------------------------------------------------------------------------
// Blocked recursive lower Cholesky in place (lower triangle of A).  Wd (64 x roundup(n,64),
// ld 64) receives the inverse of every 64x64 diagonal base block; *info (device int, must be 0
// on entry) is set to 1 when a pivot is not positive.
int gpx_potrf(hipStream_t st, long n, double* A, long lda, double* Wd, int* info);
// Factor AND invert: diagonal blocks of A <- those of L (off-diagonal blocks of A are scratch: each node keeps
// T = L21 W11 in its A21 block), W <- L^-1.  W must be zero above the diagonal on entry; T is no longer used.
int gpx_potrf_inv(hipStream_t st, long n, double* A, long lda, double* W, long ldw, double* T, int* info);
// W = L^-1 (lower, n x n, ld ldw; W must be zero above the diagonal on entry), T scratch of
// gpx_trtri_scratch(n) doubles.
long gpx_trtri_scratch(long n);
int gpx_trtri(hipStream_t st, long n, const double* L, long ldl, const double* Wd, double* W, long ldw,
              double* T);
// Out (lower tiles) = W^T W
int gpx_lauum(hipStream_t st, long n, const double* W, long ldw, double* Out, long ldo);
// b <- L^-1 b (trans = 0) or L^-T b (trans = 1); b has nrhs columns of leading dimension ldb
int gpx_trsv(hipStream_t st, long n, const double* L, long ldl, const double* Wd, double* b, int trans);
int gpx_trsm_left(hipStream_t st, long n, long nrhs, const double* L, long ldl, const double* Wd, double* B,
                  long ldb, int trans);
// out[0] = sum_i log A[i + i*lda]
int gpx_logdiag_sum(hipStream_t st, long n, const double* A, long lda, double* out);

// ---- vecops.hip -----------------------------------------------------------------------------
// y -= A x (A M x K, M-contiguous) ; y -= A^T x (y has K entries)
// tri = 1: A is lower triangular (entries above the diagonal are skipped)
int gpx_gemv_n(hipStream_t st, long M, long K, double alpha, const double* A, long lda, const double* x,
               double beta, double* y, int tri = 0);
// pan / pst: the M rows of A in row panels (GemmPanels in gpx_internal.h; 0 = plain; not with tri)
int gpx_gemv_t(hipStream_t st, long M, long K, double alpha, const double* A, long lda, const double* x,
               double beta, double* y, int tri = 0, long pan = 0, long pst = 0);
// tall matrices with few columns: the rows cut into parts that run side by side, parts summed in a fixed order (vecops.hip)
long gpx_gemv_t_split_ws(long K);
int gpx_gemv_t_split(hipStream_t st, long M, long K, double alpha, const double* A, long lda, const double* x,
                     double beta, double* y, long pan, long pst, double* ws);
