// vecops.hip — memory-bound vector / matrix-vector helpers (gfx950).
#include "gpx_internal.h"
#include "kernels.h"
#include <algorithm>
#include <cstdlib>

namespace {

// y[i] = beta*y[i] + alpha * sum_k A[i + k*lda] x[k]; 32 rows per workgroup, 16 waves x 2 half-waves split K (the last row
// blocks of a triangular factor are chains of dependent load batches and ONE CU pulls all their bytes: 278 us at n = 8192
// with 64 rows and 4 waves per block, 165 us with 16 waves)
constexpr int GEMV_NW = 16;           // waves per block
constexpr int GEMV_R = 32;            // rows per block: lanes 0-31 take column k, lanes 32-63 column k + 1
constexpr int GEMV_KS = 2 * GEMV_NW;  // columns per step of the whole block
__global__ __launch_bounds__(64 * GEMV_NW) void gemv_n_kernel(long M, long K, double alpha, const double* __restrict__ A,
                                                              long lda, const double* __restrict__ x, double beta,
                                                              double* __restrict__ y, int tri) {
    constexpr int KS = GEMV_KS, R = GEMV_R;
    __shared__ double red[KS][R + 1];
    // lower triangular A: full columns are those left of this row block; the diagonal block is masked
    const long Kf = tri ? min(K, (long)blockIdx.x * R) : K;
    if (tri) K = min(K, ((long)blockIdx.x + 1) * R);
    const int l = threadIdx.x & (R - 1), w = threadIdx.x >> 5;   // w: column slot 0 .. KS-1
    const long i = (long)blockIdx.x * R + l;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (i < M) {
        const double* a = A + i;
        long k = w;
        // eight loads in flight per lane first: the long rows of a triangular factor are bound by load latency
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        for (; k + 7 * KS < Kf; k += 8 * KS) {
            s0 = fma(a[k * lda], x[k], s0);
            s1 = fma(a[(k + KS) * lda], x[k + KS], s1);
            s2 = fma(a[(k + 2 * KS) * lda], x[k + 2 * KS], s2);
            s3 = fma(a[(k + 3 * KS) * lda], x[k + 3 * KS], s3);
            t0 = fma(a[(k + 4 * KS) * lda], x[k + 4 * KS], t0);
            t1 = fma(a[(k + 5 * KS) * lda], x[k + 5 * KS], t1);
            t2 = fma(a[(k + 6 * KS) * lda], x[k + 6 * KS], t2);
            t3 = fma(a[(k + 7 * KS) * lda], x[k + 7 * KS], t3);
        }
        s0 += t0;
        s1 += t1;
        s2 += t2;
        s3 += t3;
        for (; k + 3 * KS < Kf; k += 4 * KS) {
            s0 = fma(a[k * lda], x[k], s0);
            s1 = fma(a[(k + KS) * lda], x[k + KS], s1);
            s2 = fma(a[(k + 2 * KS) * lda], x[k + 2 * KS], s2);
            s3 = fma(a[(k + 3 * KS) * lda], x[k + 3 * KS], s3);
        }
        for (; k < Kf; k += KS) s0 = fma(a[k * lda], x[k], s0);
        if (tri)
            for (; k < K; k += KS)
                if (k <= i) s0 = fma(a[k * lda], x[k], s0);
    }
    red[w][l] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && i < M) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < KS; q += 4) t += (red[q][l] + red[q + 1][l]) + (red[q + 2][l] + red[q + 3][l]);
        y[i] = beta == 0.0 ? alpha * t : beta * y[i] + alpha * t;
    }
}

// y[j] = beta*y[j] + alpha * sum_i A[i + j*lda] x[i]; one wave per column.  pan > 0: the M rows live in row panels of pan
// rows (element (i, j) at (i / pan) * pst + i % pan + j * lda): the column is walked panel by panel.
// nsplit > 1 (blockIdx.y = part): the rows are cut into nsplit parts (whole panels, or multiples of 512 rows) and every
// part writes its sum to ws[part * K + j]; gemv_t_finish_kernel adds the parts in a fixed order.  One wave per column is
// 1024 waves for the n x 1024 matrices of FITC - four per CU: the 1 GB shard of the 8-GPU case streamed at 3.5 TB/s.
__global__ __launch_bounds__(256) void gemv_t_kernel(long M, long K, double alpha, const double* __restrict__ A,
                                                     long lda, const double* __restrict__ x, double beta,
                                                     double* __restrict__ y, int tri, long pan, long pst, int nsplit,
                                                     double* __restrict__ ws) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long j = (long)blockIdx.x * 4 + w;
    if (j >= K) return;
    double s0 = 0.0, s1 = 0.0;
    const long step = pan ? pan : M;
    long rb = 0, re = M;
    if (nsplit > 1) {
        const long unit = pan ? pan : 512, nu = (M + unit - 1) / unit;
        rb = (nu * blockIdx.y / nsplit) * unit;
        re = (nu * (blockIdx.y + 1) / nsplit) * unit;
        if (re > M) re = M;
    }
    for (long r0 = rb; r0 < re; r0 += step) {
        const double* a = A + j * lda + (pan ? (r0 / pan) * pst - r0 : 0);   // a[i] = element (i, j) for r0 <= i < r1
        const long r1 = r0 + step < re ? r0 + step : re;
        long i0 = tri ? j : r0;  // lower triangular: rows >= column
        if (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(x)) & 15) == 0) {
            // long columns (the n x m products of the sparse models have only m of them): 16-B loads, eight
            // of them in flight per lane, so that one wave per column still covers the HBM latency
            if (i0 & 1) {
                if (l == 0) s0 = fma(a[i0], x[i0], s0);
                ++i0;
            }
            double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
            const long nchunk = r1 > i0 ? (r1 - i0) / 512 : 0;  // 512 rows per wave and step
            for (long c = 0; c < nchunk; ++c) {
                const long i = i0 + 512 * c + 2 * l;
                const v2d a0 = *reinterpret_cast<const v2d*>(a + i), a1 = *reinterpret_cast<const v2d*>(a + i + 128);
                const v2d a2 = *reinterpret_cast<const v2d*>(a + i + 256), a3 = *reinterpret_cast<const v2d*>(a + i + 384);
                const v2d x0 = *reinterpret_cast<const v2d*>(x + i), x1 = *reinterpret_cast<const v2d*>(x + i + 128);
                const v2d x2 = *reinterpret_cast<const v2d*>(x + i + 256), x3 = *reinterpret_cast<const v2d*>(x + i + 384);
                t0 = fma(a0.x, x0.x, fma(a0.y, x0.y, t0));
                t1 = fma(a1.x, x1.x, fma(a1.y, x1.y, t1));
                t2 = fma(a2.x, x2.x, fma(a2.y, x2.y, t2));
                t3 = fma(a3.x, x3.x, fma(a3.y, x3.y, t3));
            }
            s0 += (t0 + t1) + (t2 + t3);
            for (long r = i0 + 512 * nchunk + l; r < r1; r += 64) s1 = fma(a[r], x[r], s1);
        } else {
            long i = l + i0;
            for (; i + 64 < r1; i += 128) {
                s0 = fma(a[i], x[i], s0);
                s1 = fma(a[i + 64], x[i + 64], s1);
            }
            for (; i < r1; i += 64) s0 = fma(a[i], x[i], s0);
        }
    }
    double s = s0 + s1;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (l == 0) {
        if (nsplit > 1) ws[(long)blockIdx.y * K + j] = s;
        else y[j] = beta == 0.0 ? alpha * s : beta * y[j] + alpha * s;
    }
}
__global__ __launch_bounds__(256) void gemv_t_finish_kernel(long K, int nsplit, const double* __restrict__ ws, double alpha,
                                                            double beta, double* __restrict__ y) {
    const long j = (long)blockIdx.x * 256 + threadIdx.x;
    if (j >= K) return;
    double s = 0.0;
    for (int q = 0; q < nsplit; ++q) s += ws[(long)q * K + j];
    y[j] = beta == 0.0 ? alpha * s : beta * y[j] + alpha * s;
}

}  // namespace

int gpx_gemv_n(hipStream_t st, long M, long K, double alpha, const double* A, long lda, const double* x,
               double beta, double* y, int tri) {
    if (M <= 0) return GPX_OK;
    hipLaunchKernelGGL(gemv_n_kernel, dim3((unsigned)((M + GEMV_R - 1) / GEMV_R)), dim3(64 * GEMV_NW), 0, st, M, K, alpha, A, lda, x,
                       beta, y, tri);
    return GPX_OK;
}

int gpx_gemv_t(hipStream_t st, long M, long K, double alpha, const double* A, long lda, const double* x,
               double beta, double* y, int tri, long pan, long pst) {
    if (K <= 0) return GPX_OK;
    if (pan && (tri || pan % 2 || pst % 2)) return GPX_ERR_ARG;
    hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)((K + 3) / 4)), dim3(256), 0, st, M, K, alpha, A, lda, x, beta,
                       y, tri, pan, pst, 1, (double*)nullptr);
    return GPX_OK;
}

// The same product with the rows cut into parts that run side by side (tall matrices with few columns); ws holds
// gpx_gemv_t_split_ws(K) doubles.  Deterministic: fixed parts, fixed order of the final sum.
long gpx_gemv_t_split_ws(long K) { return 8 * K; }
int gpx_gemv_t_split(hipStream_t st, long M, long K, double alpha, const double* A, long lda, const double* x,
                     double beta, double* y, long pan, long pst, double* ws) {
    if (K <= 0) return GPX_OK;
    if (pan && (pan % 2 || pst % 2)) return GPX_ERR_ARG;
    // enough waves for ~16 per CU, parts of at least 8192 rows
    static const int forced = [] { const char* e = getenv("GPX_GEMV_SPLIT"); return e ? atoi(e) : 0; }();
    int nsplit = forced > 0 ? forced : (int)std::min<long>(8, std::max<long>(1, std::min<long>(4096 / std::max<long>(K, 1), M / 8192)));
    if (nsplit > 8) nsplit = 8;
    if (nsplit <= 1 || !ws) return gpx_gemv_t(st, M, K, alpha, A, lda, x, beta, y, 0, pan, pst);
    hipLaunchKernelGGL(gemv_t_kernel, dim3((unsigned)((K + 3) / 4), (unsigned)nsplit), dim3(256), 0, st, M, K, alpha, A, lda, x,
                       beta, y, 0, pan, pst, nsplit, ws);
    hipLaunchKernelGGL(gemv_t_finish_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st, K, nsplit, ws, alpha, beta, y);
    return GPX_OK;
}
