// chol.hip — blocked recursive fp64 Cholesky, triangular inverse, W^T W and triangular solves.
//
// Replaces MatrixSolverLLT (MatrixSolver.cpp:6-29): decomp -> gpx_potrf, check_SPD -> the info
// flag, log_det -> gpx_logdiag_sum, solve -> gpx_trsv / gpx_trsm_left, inverse -> gpx_trtri +
// gpx_lauum.  Every O(n^3) flop runs in the MFMA GEMM of gemm_f64.hip: the recursion splits the
// matrix in halves (rounded to the 64-wide base block), so the trailing updates are a few large
// SYRK/GEMM launches; only 64x64 diagonal blocks are factored (and inverted) by a scalar
// one-workgroup kernel, and triangular solves against them are products with those inverses.
#include "gpx_internal.h"
#include "kernels.h"

namespace {

constexpr int NB = GPX_NB0;
constexpr int SP = NB + 1;  // padded LDS stride

// Factor one nb x nb (nb <= 64) diagonal block in LDS; write L back and W = L^-1 to Wd (64x64, ld 64).
__global__ __launch_bounds__(256) void potrf_base_kernel(double* __restrict__ A, long lda, int nb,
                                                         double* __restrict__ Wd, int* __restrict__ info) {
    __shared__ double S[NB * SP];  // S[j*SP + i] = element (i, j)
    __shared__ double Wm[NB * SP];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx % NB, j = idx / NB;
        double v = (i == j) ? 1.0 : 0.0;  // identity padding for partial blocks
        if (i < nb && j < nb) v = (i >= j) ? A[i + (long)j * lda] : 0.0;
        S[j * SP + i] = v;
    }
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        if (tid == 0) {
            const double p = S[j * SP + j];
            if (!(p > 0.0)) *info = 1;  // Eigen::LLT reports NumericalIssue on a pivot <= 0
            S[j * SP + j] = sqrt(p);
        }
        __syncthreads();
        const double dinv = 1.0 / S[j * SP + j];
        const double djj = S[j * SP + j];
        if (tid > j && tid < nb) S[j * SP + tid] = S[j * SP + tid] / djj;
        (void)dinv;
        __syncthreads();
        const int t = nb - 1 - j;
        for (int idx = tid; idx < t * t; idx += 256) {
            const int ii = idx % t, kk = idx / t;
            if (ii >= kk) S[(j + 1 + kk) * SP + j + 1 + ii] -= S[j * SP + j + 1 + ii] * S[j * SP + j + 1 + kk];
        }
        __syncthreads();
    }
    for (int idx = tid; idx < nb * nb; idx += 256) {
        const int i = idx % nb, j = idx / nb;
        if (i >= j) A[i + (long)j * lda] = S[j * SP + i];
    }
    // W = L^-1 by forward substitution, one column per thread (identity-padded to 64)
    if (tid < NB) {
        const int c = tid;
        for (int i = 0; i < c; ++i) Wm[c * SP + i] = 0.0;
        Wm[c * SP + c] = 1.0 / S[c * SP + c];
        for (int i = c + 1; i < NB; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s = fma(S[k * SP + i], Wm[c * SP + k], s);
            Wm[c * SP + i] = -s / S[i * SP + i];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx % NB, j = idx / NB;
        Wd[i + j * NB] = Wm[j * SP + i];
    }
}

__device__ __forceinline__ double lane_bcast(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// One WAVE factors a 64x64 (identity-padded) diagonal block held in registers: lane i owns row i.
// Column values are broadcast with v_readlane (no LDS, no barriers).  Writes L over the lower
// triangle of A and W = L^-1 (lower, zeros above) to W with leading dimension ldw.
__global__ __launch_bounds__(64) void potrf_base_reg_kernel(double* __restrict__ A, long lda, int nb,
                                                            double* __restrict__ W, long ldw, int wfull,
                                                            int* __restrict__ info) {
    const int i = threadIdx.x;
    double a[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) a[k] = (i < nb && k < nb && k <= i) ? A[i + (long)k * lda] : ((i == k) ? 1.0 : 0.0);
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double piv = lane_bcast(a[j], j);
        bad = bad || !(piv > 0.0);  // Eigen::LLT: NumericalIssue on a pivot <= 0
        const double rs = sqrt(piv);
        a[j] = (i == j) ? rs : a[j] / rs;
#pragma unroll
        for (int k = j + 1; k < NB; ++k) a[k] = fma(-a[j], lane_bcast(a[j], k), a[k]);
    }
    if (bad && i == 0) *info = 1;
#pragma unroll
    for (int k = 0; k < NB; ++k)
        if (i < nb && k < nb && k <= i) A[i + (long)k * lda] = a[k];
    // column c = lane of W = L^-1 by forward substitution; L_ik comes from lane i, register k
    const int c = i;
    double w[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        double s = (r == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k) s = fma(-lane_bcast(a[k], r), w[k], s);
        w[r] = s / lane_bcast(a[r], r);
    }
    const int lim = wfull ? NB : nb;
#pragma unroll
    for (int r = 0; r < NB; ++r)
        if (r < lim && c < lim) W[r + (long)c * ldw] = (r >= c) ? w[r] : 0.0;
}

// B (M x nb) <- B * W^T with W = inverse of a base block (lower): the TRSM X L^T = B
__global__ __launch_bounds__(256) void trsm_right_base_kernel(double* __restrict__ B, long ldb, long M, int nb,
                                                              const double* __restrict__ Wd) {
    __shared__ double sB[NB * SP];  // sB[k*SP + i]
    __shared__ double sW[NB * SP];  // sW[k*SP + j] = W[j][k]
    const int tid = threadIdx.x;
    const long r0 = (long)blockIdx.x * NB;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int i = idx % NB, k = idx / NB;
        sB[k * SP + i] = (r0 + i < M && k < nb) ? B[r0 + i + (long)k * ldb] : 0.0;
        sW[k * SP + i] = Wd[i + k * NB];  // W[i][k], i = row j of W
    }
    __syncthreads();
    const int i = tid & 63, jq = tid >> 6;
    double out[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) out[jj] = 0.0;
    for (int k = 0; k < nb; ++k) {
        const double b = sB[k * SP + i];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) out[jj] = fma(b, sW[k * SP + jq * 16 + jj], out[jj]);
    }
    if (r0 + i < M) {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = jq * 16 + jj;
            if (j < nb) B[r0 + i + (long)j * ldb] = out[jj];
        }
    }
}

// B_blk (nb x up to 64 columns per workgroup) <- W * B_blk  or  W^T * B_blk
__global__ __launch_bounds__(256) void trsm_left_base_kernel(double* __restrict__ B, long ldb, long nrhs, int nb,
                                                             const double* __restrict__ Wd, int trans) {
    __shared__ double sB[NB * SP];  // sB[c*SP + k]
    __shared__ double sW[NB * SP];  // sW[k*SP + i] = op(W)[i][k]
    const int tid = threadIdx.x;
    const long c0 = (long)blockIdx.x * NB;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int k = idx % NB, c = idx / NB;
        sB[c * SP + k] = (c0 + c < nrhs && k < nb) ? B[k + (c0 + c) * ldb] : 0.0;
        // element (i=k', k=c') of W is Wd[k' + c'*64]
        sW[c * SP + k] = trans ? Wd[c + k * NB] : Wd[k + c * NB];
    }
    __syncthreads();
    const int i = tid & 63, cq = tid >> 6;
    double out[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) out[cc] = 0.0;
    for (int k = 0; k < nb; ++k) {
        const double w = sW[k * SP + i];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) out[cc] = fma(w, sB[(cq * 16 + cc) * SP + k], out[cc]);
    }
    if (i < nb) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const long c = c0 + cq * 16 + cc;
            if (c < nrhs) B[i + c * ldb] = out[cc];
        }
    }
}

__global__ void copy_base_blocks_kernel(const double* __restrict__ Wd, long n, double* __restrict__ W, long ldw) {
    const long o = (long)blockIdx.x * NB;
    for (int idx = threadIdx.x; idx < NB * NB; idx += blockDim.x) {
        const int i = idx % NB, j = idx / NB;
        if (o + i < n && o + j < n) W[o + i + (o + j) * ldw] = Wd[o * NB + i + j * NB];
    }
}

__global__ void logdiag_sum_kernel(const double* __restrict__ A, long lda, long n, double* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) s += log(A[i + i * lda]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

inline long split(long n) { return ((n / 2 + NB - 1) / NB) * NB; }

int trsm_right_rec(hipStream_t st, long M, long n1, const double* L, long ldl, const double* Wd, double* B,
                   long ldb) {
    if (n1 <= NB) {
        hipLaunchKernelGGL(trsm_right_base_kernel, dim3((unsigned)((M + NB - 1) / NB)), dim3(256), 0, st, B, ldb, M,
                           (int)n1, Wd);
        return GPX_OK;
    }
    const long a = split(n1), b = n1 - a;
    int rc = trsm_right_rec(st, M, a, L, ldl, Wd, B, ldb);
    if (rc) return rc;
    // B[:, a:] -= B[:, :a] * L[a:, :a]^T
    rc = gpx_gemm(st, 0, 0, M, b, a, -1.0, B, ldb, L + a, ldl, 1.0, B + a * ldb, ldb, 0);
    if (rc) return rc;
    return trsm_right_rec(st, M, b, L + a + a * ldl, ldl, Wd + a * NB, B + a * ldb, ldb);
}

int potrf_rec(hipStream_t st, long n, double* A, long lda, double* Wd, int* info) {
    if (n <= NB) {
        hipLaunchKernelGGL(potrf_base_reg_kernel, dim3(1), dim3(64), 0, st, A, lda, (int)n, Wd, (long)NB, 1, info);
        return GPX_OK;
    }
    const long n1 = split(n), n2 = n - n1;
    int rc = potrf_rec(st, n1, A, lda, Wd, info);
    if (rc) return rc;
    rc = trsm_right_rec(st, n2, n1, A, lda, Wd, A + n1, lda);
    if (rc) return rc;
    rc = gpx_gemm(st, 0, 0, n2, n2, n1, -1.0, A + n1, lda, A + n1, lda, 1.0, A + n1 + n1 * lda, lda, GEMM_LOWER);
    if (rc) return rc;
    return potrf_rec(st, n2, A + n1 + n1 * lda, lda, Wd + n1 * NB, info);
}

int trtri_rec(hipStream_t st, long n, const double* L, long ldl, double* W, long ldw, double* T) {
    if (n <= NB) return GPX_OK;  // diagonal base blocks were copied from Wd
    const long n1 = split(n), n2 = n - n1;
    int rc = trtri_rec(st, n1, L, ldl, W, ldw, T);
    if (rc) return rc;
    rc = trtri_rec(st, n2, L + n1 + n1 * ldl, ldl, W + n1 + n1 * ldw, ldw, T);
    if (rc) return rc;
    const long ldt = gpx_ld(n2);
    // T = L21 * W11      (W11 lower: B(n,k) = W11[k + n*ldw] is zero for k < n)
    rc = gpx_gemm(st, 0, 1, n2, n1, n1, 1.0, L + n1, ldl, W, ldw, 0.0, T, ldt, GEMM_KLO_N);
    if (rc) return rc;
    // W21 = -W22 * T     (W22 lower: A(m,k) zero for k > m)
    return gpx_gemm(st, 0, 1, n2, n1, n2, -1.0, W + n1 + n1 * ldw, ldw, T, ldt, 0.0, W + n1, ldw,
                    GEMM_KHI_M | GEMM_HEAVY_LAST);
}

// Factor and invert in one recursion: on return the diagonal blocks of A hold those of L (the
// off-diagonal blocks of A are NOT L) and W = L^-1.  L21 lives in W's (2,1) block until that block
// is overwritten by W21, so there is no triangular solve and no small kernel besides the 64x64 base.
int potrf_inv_rec(hipStream_t st, long n, double* A, long lda, double* W, long ldw, double* T, int* info) {
    if (n <= NB) {
        hipLaunchKernelGGL(potrf_base_reg_kernel, dim3(1), dim3(64), 0, st, A, lda, (int)n, W, ldw, 0, info);
        return GPX_OK;
    }
    const long n1 = split(n), n2 = n - n1;
    double* W21 = W + n1;
    double* W22 = W + n1 + n1 * ldw;
    int rc = potrf_inv_rec(st, n1, A, lda, W, ldw, T, info);
    if (rc) return rc;
    // L21 = A21 * W11^T   (B(n,k) = W11[n + k*ldw], zero for k > n)
    rc = gpx_gemm(st, 0, 0, n2, n1, n1, 1.0, A + n1, lda, W, ldw, 0.0, W21, ldw, GEMM_KHI_N);
    if (rc) return rc;
    rc = gpx_gemm(st, 0, 0, n2, n2, n1, -1.0, W21, ldw, W21, ldw, 1.0, A + n1 + n1 * lda, lda, GEMM_LOWER);
    if (rc) return rc;
    rc = potrf_inv_rec(st, n2, A + n1 + n1 * lda, lda, W22, ldw, T, info);
    if (rc) return rc;
    const long ldt = gpx_ld(n2);
    rc = gpx_gemm(st, 0, 1, n2, n1, n1, 1.0, W21, ldw, W, ldw, 0.0, T, ldt, GEMM_KLO_N);  // T = L21 W11
    if (rc) return rc;
    return gpx_gemm(st, 0, 1, n2, n1, n2, -1.0, W22, ldw, T, ldt, 0.0, W21, ldw, GEMM_KHI_M | GEMM_HEAVY_LAST);
}

int trsm_left_rec(hipStream_t st, long n, long nrhs, const double* L, long ldl, const double* Wd, double* B,
                  long ldb, int trans) {
    if (n <= NB) {
        hipLaunchKernelGGL(trsm_left_base_kernel, dim3((unsigned)((nrhs + NB - 1) / NB)), dim3(256), 0, st, B, ldb,
                           nrhs, (int)n, Wd, trans);
        return GPX_OK;
    }
    const long n1 = split(n), n2 = n - n1;
    int rc;
    if (!trans) {
        rc = trsm_left_rec(st, n1, nrhs, L, ldl, Wd, B, ldb, 0);
        if (rc) return rc;
        if (nrhs == 1) rc = gpx_gemv_n(st, n2, n1, -1.0, L + n1, ldl, B, 1.0, B + n1);
        else rc = gpx_gemm(st, 0, 1, n2, nrhs, n1, -1.0, L + n1, ldl, B, ldb, 1.0, B + n1, ldb, 0);
        if (rc) return rc;
        return trsm_left_rec(st, n2, nrhs, L + n1 + n1 * ldl, ldl, Wd + n1 * NB, B + n1, ldb, 0);
    }
    rc = trsm_left_rec(st, n2, nrhs, L + n1 + n1 * ldl, ldl, Wd + n1 * NB, B + n1, ldb, 1);
    if (rc) return rc;
    if (nrhs == 1) rc = gpx_gemv_t(st, n2, n1, -1.0, L + n1, ldl, B + n1, 1.0, B);
    else rc = gpx_gemm(st, 1, 1, n1, nrhs, n2, -1.0, L + n1, ldl, B + n1, ldb, 1.0, B, ldb, 0);
    if (rc) return rc;
    return trsm_left_rec(st, n1, nrhs, L, ldl, Wd, B, ldb, 1);
}

}  // namespace

int gpx_potrf(hipStream_t st, long n, double* A, long lda, double* Wd, int* info) {
    if (n <= 0) return GPX_OK;
    const int rc = potrf_rec(st, n, A, lda, Wd, info);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

int gpx_potrf_inv(hipStream_t st, long n, double* A, long lda, double* W, long ldw, double* T, int* info) {
    if (n <= 0) return GPX_OK;
    const int rc = potrf_inv_rec(st, n, A, lda, W, ldw, T, info);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

long gpx_trtri_scratch(long n) {
    if (n <= NB) return 16;
    const long n1 = split(n), n2 = n - n1;
    return gpx_ld(n2) * n1 + 16;
}

int gpx_trtri(hipStream_t st, long n, const double* L, long ldl, const double* Wd, double* W, long ldw,
              double* T) {
    if (n <= 0) return GPX_OK;
    hipLaunchKernelGGL(copy_base_blocks_kernel, dim3((unsigned)((n + NB - 1) / NB)), dim3(256), 0, st, Wd, n, W,
                       ldw);
    const int rc = trtri_rec(st, n, L, ldl, W, ldw, T);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

int gpx_lauum(hipStream_t st, long n, const double* W, long ldw, double* Out, long ldo) {
    // Out_mn = sum_k W[k][m] W[k][n], k >= max(m, n) = m on the lower tiles
    return gpx_gemm(st, 1, 1, n, n, n, 1.0, W, ldw, W, ldw, 0.0, Out, ldo, GEMM_LOWER | GEMM_KLO_M);
}

int gpx_trsm_left(hipStream_t st, long n, long nrhs, const double* L, long ldl, const double* Wd, double* B,
                  long ldb, int trans) {
    if (n <= 0 || nrhs <= 0) return GPX_OK;
    const int rc = trsm_left_rec(st, n, nrhs, L, ldl, Wd, B, ldb, trans);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

int gpx_trsv(hipStream_t st, long n, const double* L, long ldl, const double* Wd, double* b, int trans) {
    return gpx_trsm_left(st, n, 1, L, ldl, Wd, b, n, trans);
}

int gpx_logdiag_sum(hipStream_t st, long n, const double* A, long lda, double* out) {
    hipLaunchKernelGGL(logdiag_sum_kernel, dim3(1), dim3(256), 0, st, A, lda, n, out);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}
