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

// Factor AND invert one nb x nb (nb <= 64, identity-padded to 64) diagonal block in ONE workgroup.
// Right-looking elimination on the register-resident pair [S | E]: thread (tr, tc) owns the 4x4 tile
// rows 4tr.., columns 4tc.. of both the Schur complement S and of E (initially I).  Step j needs only
// column j of S and row j of E; their owners publish them to a double-buffered LDS vector during
// step j-1 (look-ahead), so a step costs ONE barrier:  S_ik -= S_ij S_kj / d_j,  E_i: -= (S_ij/d_j) E_j:.
// At the end L_ij = S_ij / sqrt(d_j) and W = L^-1 = diag(1/sqrt(d)) E.  Writes L over the lower
// triangle of A and W (lower, zeros above) with leading dimension ldw.
__global__ __launch_bounds__(256) void potrf_base_tile_kernel(double* __restrict__ A, long lda, int nb,
                                                              double* __restrict__ W, long ldw, int wfull,
                                                              int* __restrict__ info) {
    __shared__ __align__(16) double colbuf[2][NB];
    __shared__ __align__(16) double rowbuf[2][NB];
    __shared__ double dvec[NB];
    const int tid = threadIdx.x, tr = tid & 15, tc = tid >> 4;
    const int r0 = 4 * tr, c0 = 4 * tc;
    double s[4][4], e[4][4];  // [a][b] = element (r0 + a, c0 + b)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = r0 + a, k = c0 + b;
            e[a][b] = (i == k) ? 1.0 : 0.0;
            s[a][b] = (i < nb && k < nb && k <= i) ? A[i + (long)k * lda] : e[a][b];
        }
    if (tc == 0)
#pragma unroll
        for (int a = 0; a < 4; ++a) colbuf[0][r0 + a] = s[a][0];
    if (tr == 0)
#pragma unroll
        for (int b = 0; b < 4; ++b) rowbuf[0][c0 + b] = e[0][b];
    __syncthreads();
    bool bad = false;
    for (int jt = 0; jt < NB / 4; ++jt) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * jt + jj;
            const double* cb = colbuf[jj & 1];
            const double* rb = rowbuf[jj & 1];
            const double piv = cb[j];
            bad = bad || !(piv > 0.0);  // Eigen::LLT: NumericalIssue on a pivot <= 0
            const double inv = 1.0 / piv;
            double li[4], lk[4], mi[4], rj[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                li[a] = (r0 + a > j) ? cb[r0 + a] : 0.0;
                mi[a] = li[a] * inv;
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                lk[b] = (c0 + b > j) ? cb[c0 + b] * inv : 0.0;
                rj[b] = rb[c0 + b];
            }
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    s[a][b] = fma(-li[a], lk[b], s[a][b]);
                    e[a][b] = fma(-mi[a], rj[b], e[a][b]);
                }
            if (tid == 0) dvec[j] = piv;
            // look-ahead: publish column j+1 of S and row j+1 of E for the next step
            const int jn = j + 1, bn = (jj + 1) & 3, tn = jn >> 2;
            if (jn < NB) {
                double* cbn = colbuf[(jj + 1) & 1];
                double* rbn = rowbuf[(jj + 1) & 1];
                if (tc == tn)
#pragma unroll
                    for (int a = 0; a < 4; ++a) cbn[r0 + a] = s[a][bn];
                if (tr == tn)
#pragma unroll
                    for (int b = 0; b < 4; ++b) rbn[c0 + b] = e[bn][b];
            }
            __syncthreads();
        }
    }
    if (bad && tid == 0) *info = 1;
    double sc[4], sr[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) sc[b] = sqrt(dvec[c0 + b]);
#pragma unroll
    for (int a = 0; a < 4; ++a) sr[a] = 1.0 / sqrt(dvec[r0 + a]);
    const int lim = wfull ? NB : nb;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const double ic = 1.0 / sc[b];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = r0 + a, k = c0 + b;
            if (i < nb && k < nb && k <= i) A[i + (long)k * lda] = (i == k) ? sc[b] : s[a][b] * ic;
            if (i < lim && k < lim) W[i + (long)k * ldw] = (i >= k) ? e[a][b] * sr[a] : 0.0;
        }
    }
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
        hipLaunchKernelGGL(potrf_base_tile_kernel, dim3(1), dim3(256), 0, st, A, lda, (int)n, Wd, (long)NB, 1, info);
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
        hipLaunchKernelGGL(potrf_base_tile_kernel, dim3(1), dim3(256), 0, st, A, lda, (int)n, W, ldw, 0, info);
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
