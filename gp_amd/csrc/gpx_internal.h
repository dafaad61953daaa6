// gpx_internal.h — shared declarations of the HIP side of libgpx.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cmath>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

#define GPX_OK 0
#define GPX_ERR_ARG 1
#define GPX_ERR_HIP 2
#define GPX_ERR_STATE 3
#define GPX_ERR_UNSUPPORTED 4

#define GPX_MAX_DIM 128   // input dimensions held in LDS by the pairwise kernels
#define GPX_NB0 64        // base block of the recursive factorisations

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            fprintf(stderr, "gpx: HIP error %s at %s:%d: %s\n", hipGetErrorName(_e),         \
                    __FILE__, __LINE__, #expr);                                             \
            return GPX_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

// leading dimension for internal column-major matrices: multiple of 16 doubles (128 B), never a
// multiple of 256 doubles so that column strides do not alias on HBM channels
static inline long gpx_ld(long n) {
    long ld = (n + 15) / 16 * 16;
    if (ld % 256 == 0) ld += 16;
    return ld;
}

// ---- GEMM (gemm_f64.hip) -------------------------------------------------------------------
// C[M x N] = alpha * op(A) * op(B)^T + beta * C, all column-major.
//   a_kc == 0: A(m,k) at A[m + k*lda] (M-contiguous);  a_kc == 1: A(m,k) at A[k + m*lda]
//   b_kc == 0: B(n,k) at B[n + k*ldb] (N-contiguous);  b_kc == 1: B(n,k) at B[k + n*ldb]
// flags:
#include "gemm_flags.h"
// Row-panel layout of the tall n x m matrices of the sparse models (Kxu, V, B, C): the n rows are cut into panels of
// `pan` rows (a multiple of 128), element (a, b) lives at (a / pan) * pst + a % pan + b * ld with ld = pan, pst >= pan * m.
// A column of a panel is pan * 8 bytes from the next (150 KB) instead of n * 8 (8 MB): the K-contiguous operand tiles of
// the Gram products Kux G Kxu (128 columns x 16 rows per K step) touch ~10 pages instead of 128.  pan = 0: plain
// column-major.  Per operand: the panelled index is its CONTIGUOUS one (m for A when !a_kc, k when a_kc; C and E: m).
struct GemmPanels {
    long panA, pstA, panB, pstB, panC, pstC, panE, pstE;
};
int gpx_gemm(hipStream_t st, int a_kc, int b_kc, long M, long N, long K, double alpha,
             const double* A, long lda, const double* B, long ldb, double beta, double* C, long ldc,
             int flags, const GemmPanels* pn = nullptr);
// split-K variant for short-and-wide outputs (K >> M,N): partial slabs in `work`
// (>= splitk*M*N doubles), then a deterministic reduction.  work may be NULL -> plain gemm.
// kscale (optional, K-contiguous A only, 16-B aligned): C = alpha * op(A) diag(kscale) op(B)^T + beta*C,
// the weighted Gram products K^T diag(g) K of the sparse models without a scaled copy of K.
// two independent products, one launch where both are plain small ones (gemm_f64.hip gemm_pair_kernel): the first with
// M/N-contiguous operands, the second with a K-contiguous B
int gpx_gemm_pair(hipStream_t st, long M0, long N0, long K0, double alpha0, const double* A0, long lda0, const double* B0, long ldb0,
                  double beta0, double* C0, long ldc0, int flags0, long M1, long N1, long K1, double alpha1, const double* A1,
                  long lda1, const double* B1, long ldb1, double beta1, double* C1, long ldc1, int flags1);
int gpx_gemm_splitk(hipStream_t st, int a_kc, int b_kc, long M, long N, long K, double alpha,
                    const double* A, long lda, const double* B, long ldb, double beta, double* C,
                    long ldc, int flags, double* work, long work_doubles, const double* kscale = nullptr,
                    const GemmPanels* pn = nullptr);
long gpx_gemm_splitk_panel_rows(long M, long N, long K, int flags, long work_doubles);

// C = alpha op(A) op(B)^T (128-tiles, no split-K) with a row-dot epilogue: per part p (64 columns) of the output
//   R1[p*ldr + m] = sum_{n in part p} C(m,n) * E(m,n)     (E == NULL: E = C, the row sums of squares)
//   R2[p*ldr + m] = sum_{n in part p} E(m,n) * evec[n]    (E, evec and R2 come together or not at all)
// (2*ceil(N/128) parts, column n in part 2*(n/128) + (n%64)/32; ldr >= M; every part of every row m < M is written, deterministically)
// Operand dot (E == R1 == nullptr, evec with K entries, R2 with M entries): R2[m] = sum_k A(m,k) evec[k], from the A fragments.
struct GemmRowdot {
    const double* E; long lde;
    const double* evec;
    double* R1; double* R2; long ldr;
};
int gpx_gemm_rowdot(hipStream_t st, int a_kc, int b_kc, long M, long N, long K, double alpha, const double* A, long lda,
                    const double* B, long ldb, double* C, long ldc, int flags, const GemmRowdot& rd, const GemmPanels* pn = nullptr);

// per-launch HIP-event timing of GEMM launches with >= min_flops useful flops (bench.py roofline)
int gpx_gemm_prof_begin(double min_flops, int max_launches);
int gpx_gemm_prof_end(double* total_ms, double* total_flops, long* launches, double* skipped_flops);
int gpx_mfma_peak(double* tflops);
