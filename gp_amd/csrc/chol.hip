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
#include <algorithm>
#include <cstdlib>

namespace {

constexpr int NB = GPX_NB0;
constexpr int SP = NB + 1;  // padded LDS stride

// Factor AND invert one nb x nb (nb <= 64, identity-padded to 64) diagonal block in ONE workgroup.
// Right-looking LDL^T elimination on the register-resident pair [S | E] (E starts as I): thread
// (tr, tc) owns the 4x4 patch rows 4tr.., columns 4tc.. of both.  The 16 panels of 4 columns cost two
// barriers each:
//   A  the owner of the diagonal patch eliminates it in registers (the only serial divisions) and
//      publishes its unscaled lower entries u_kj, the multipliers m_kj = u_kj / d_j and 1/d_j;
//   B  the patches below it finish their 4 columns (publishing u_ij, m_ij), the E patches of the panel
//      rows apply the same 4x4 unit-lower elimination (publishing the 4 finished rows of E);
//   C  every patch below the panel takes ONE rank-4 update: S_ik -= sum_j u_ij m_kj right of the
//      panel, E_i: -= sum_j m_ij E_j: left of it (a patch is on one side only).
// At the end L_ij = u_ij / sqrt(d_j) and W = L^-1 = diag(1/sqrt(d)) E.  Writes L over the lower
// triangle of A and W (lower, zeros above) with leading dimension ldw.
__global__ __launch_bounds__(256) void potrf_base_tile_kernel(double* __restrict__ A, long lda, int nb,
                                                              double* __restrict__ W, long ldw, int wfull,
                                                              int* __restrict__ info) {
    __shared__ __align__(16) double U11[4][4], M11[4][4], invd[4];
    __shared__ __align__(16) double Ucol[NB][4], Mcol[NB][4], Erow[4][NB];
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
    bool bad = false;
    for (int p = 0; p < NB / 4; ++p) {
        // ---- A: diagonal patch
        if (tr == p && tc == p) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double d = s[j][j];
                bad = bad || !(d > 0.0);  // Eigen::LLT: NumericalIssue on a pivot <= 0
                // 1/d on the critical path of the whole block: hardware reciprocal + two Newton steps (full
                // precision to the last bit or two) instead of the IEEE division sequence
                double inv = __builtin_amdgcn_rcp(d);
                inv = fma(fma(-d, inv, 1.0), inv, inv);
                inv = fma(fma(-d, inv, 1.0), inv, inv);
                dvec[4 * p + j] = d;
                invd[j] = inv;
                double m[4];
#pragma unroll
                for (int a = j + 1; a < 4; ++a) {
                    m[a] = s[a][j] * inv;
                    U11[a][j] = s[a][j];
                    M11[a][j] = m[a];
                }
#pragma unroll
                for (int b = j + 1; b < 4; ++b)
#pragma unroll
                    for (int a = b; a < 4; ++a) s[a][b] = fma(-m[a], s[b][j], s[a][b]);
            }
        }
        __syncthreads();
        // ---- B: rest of the column panel, and the panel's rows of E
        if (tc == p && tr > p) {
            double u[4][4], mm[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double inv = invd[j];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    u[a][j] = s[a][j];
                    mm[a][j] = u[a][j] * inv;
                }
#pragma unroll
                for (int k = j + 1; k < 4; ++k) {
                    const double ukj = U11[k][j];
#pragma unroll
                    for (int a = 0; a < 4; ++a) s[a][k] = fma(-mm[a][j], ukj, s[a][k]);
                }
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    Ucol[r0 + a][j] = u[a][j];
                    Mcol[r0 + a][j] = mm[a][j];
                }
        }
        if (tr == p && tc <= p) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int a = j + 1; a < 4; ++a) {
                    const double m = M11[a][j];
#pragma unroll
                    for (int b = 0; b < 4; ++b) e[a][b] = fma(-m, e[j][b], e[a][b]);
                }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) Erow[a][c0 + b] = e[a][b];
        }
        __syncthreads();
        // ---- C: one rank-4 update per patch below the panel
        if (tr > p) {
            if (tc > p) {
                if (tc <= tr) {
                    double u[4][4], mk[4][4];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            u[a][j] = Ucol[r0 + a][j];
                            mk[a][j] = Mcol[c0 + a][j];
                        }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int b = 0; b < 4; ++b)
#pragma unroll
                            for (int a = 0; a < 4; ++a) s[a][b] = fma(-u[a][j], mk[b][j], s[a][b]);
                }
            } else {
                double mi[4][4], er[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        mi[a][j] = Mcol[r0 + a][j];
                        er[a][j] = Erow[a][c0 + j];  // er[j'][b]: row j' of the panel, column c0 + b
                    }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int a = 0; a < 4; ++a) e[a][b] = fma(-mi[a][j], er[j][b], e[a][b]);
            }
        }
    }
    __syncthreads();
    if (bad) *info = 1;
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

// ---- factor AND invert one nb x nb block, nb <= 128, in ONE workgroup on the matrix cores ----------------
// The base case of potrf_inv_rec.  The block lives in LDS (column-major, leading dimension 132) for the whole
// kernel; 16 x 16 sub-blocks are the unit of work (v_mfma_f64_16x16x4_f64):
//   for k = 0 .. nb/16 - 1:   wave 0 factors the diagonal sub-block in registers (one row per lane, pivots and
//                             columns broadcast with v_readlane);  the panel below it is a forward substitution
//                             X L_kk^T = S_:k with one row per thread (L_kk read as LDS broadcasts), so that no
//                             inverse sits on the critical path;  the trailing update  S_ij -= L_ik L_jk^T  is one
//                             4-MFMA chain per sub-block, with a look-ahead: wave 0 updates and factors diagonal
//                             sub-block k + 1 WHILE waves 1..7 update block column k + 1 and then the rest
//                             (factor phase 29 us -> 25 us at nb = 128);  2 barriers per k.  Per k (base_probe):
//                             panel 0.8 us (1.24 before its LDS reads were batched), wave 0's update + factorisation 1.9 - 2.4 us (270 cycles per pivot: the
//                             v_readlane broadcasts of a column), the other waves wait 0.08 us for it.
//                             Waves 1..7 store block column k of L to memory while they wait for wave 0 (no store phase).
//   The inverses of the diagonal sub-blocks follow, all at once (one per wave, one column per lane).
//   W = L^-1 below the diagonal sub-blocks: block column w on wave w < 4, 11 - w on wave w >= 4 (waves w, w + 4 share a SIMD),
//   right-looking inside the column (W_qj known -> every D_i += L_iq W_qj, independent 4-MFMA chains whose operands are all
//   requested before the first MFMA; then W_(q+1)j = -W_(q+1)(q+1) D_(q+1)), no barrier at all: W_ij is parked in the (unused)
//   upper block (j, i) so that L stays intact, and stored to memory by the same wave one step later, beside the next step's
//   MFMAs.  (Left-looking with every operand read right before its MFMA: 122 cycles per MFMA, 8.3 us for block column 0.)
// (The block inverse by recursive doubling over the sub-blocks - T = L21 W11, W21 = -W22 T per level, at most 12 dependent
// block products instead of 35 - was written and measured: 7.8 us against 8.2 us at the time; not kept.)
// Replaces, per 128 rows, two launches of the 64 x 64 kernel above and four single-tile GEMM launches (74 us ->
// 43 us at the end of round 2, 37 us now; phases: tools/probe/base_probe.hip): the serial chain of small launches is what bounds exact n <= 8192 and the replicated m x m
// phase of the sparse models.  Contract as potrf_base_tile_kernel(..., wfull = 0): A's lower triangle <- L,
// W (nb x nb) <- L^-1 with zeros above the diagonal, *info = 1 on a pivot <= 0 (Eigen::LLT NumericalIssue).
constexpr int NBI = 128;       // rows of the base block
constexpr int LDI = NBI + 4;   // LDS leading dimension (16-B aligned columns)
constexpr int LDW = 17;        // leading dimension of the 16 x 16 diagonal inverses
#ifdef GPX_BASE_TIMING             // tools/probe/base_probe.hip: 100 MHz time stamps of the kernel's phases
__device__ long long g_base_t[64];
#define BASE_STAMP(i) do { if (threadIdx.x == 0) g_base_t[i] = wall_clock64(); } while (0)
#define BASE_STAMP_WAVE(i) do { if ((threadIdx.x & 63) == 0) g_base_t[i] = wall_clock64(); } while (0)
#else
#define BASE_STAMP(i) do { } while (0)
#define BASE_STAMP_WAVE(i) do { } while (0)
#endif

__device__ __forceinline__ double readlane_f64(double v, int lane) {   // lane: wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ v4d mfma16(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// XP = true (GPX_EXACT_PIVOT=1, A/B): correctly rounded pivots, l_jj = sqrt(d), column divided by l_jj, 1 / l_jj by division
template <bool XP>
__global__ __launch_bounds__(512) void potrf_inv_base_kernel(double* __restrict__ A, long lda, int nb, double* __restrict__ W,
                                                             long ldw, int* __restrict__ info) {
    __shared__ __align__(16) double S[NBI * LDI];
    __shared__ __align__(16) double Wd[NBI / 16][16 * LDW];
    __shared__ double invd[NBI];   // 1 / l_jj
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, l4 = lane >> 4;
    const int nblk = (nb + 15) / 16, np = 16 * nblk;
    BASE_STAMP(0);
    {   // lower triangle, identity padding, zeros above: all loads of a thread in flight together (clamped addresses,
        // no divergent branches; a guarded loop pays one HBM round trip per element: 32 in a row)
        double v[NBI * NBI / 512];
#pragma unroll
        for (int q = 0; q < NBI * NBI / 512; ++q) {
            const int idx = tid + 512 * q, r = idx & (NBI - 1), c = idx >> 7;
            v[q] = (c <= r) ? A[min(r, nb - 1) + (long)min(c, nb - 1) * lda] : 0.0;   // lower triangle only (masked, no wait between)
        }
#pragma unroll
        for (int q = 0; q < NBI * NBI / 512; ++q) {
            const int idx = tid + 512 * q, r = idx & (NBI - 1), c = idx >> 7;
            S[r + c * LDI] = (r < nb && c <= r) ? v[q] : (r == c ? 1.0 : 0.0);
        }
    }
    bool bad = false;
    const bool lvec = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 2 == 0);   // 16-byte stores of L / W are aligned
    const bool wvec = ((reinterpret_cast<uintptr_t>(W) & 15) == 0) && (ldw % 2 == 0);
    auto chol = [&](int o) {   // wave 0 only
        // ---- Cholesky of the 16 x 16 diagonal sub-block: lane i (and its three copies) holds row i
        double r[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) r[c] = S[(o + l15) + (o + c) * LDI];
        double tl = 0.0;   // 1 / l_jj of this lane's row
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double d = readlane_f64(r[j], j);
            bad = bad || !(d > 0.0);
            // 1/sqrt(d) on the critical path of the whole block: hardware estimate t0 (>= 26 bits), then ONE third-order
            // step t0 (1 + e/2 + 3 e^2/8), e = 1 - d t0^2 (error ~ e^3: full precision, two dependent operations less
            // than two Newton steps)
            double t, lij;
            if (XP) {
                const double l = sqrt(d);
                t = 1.0 / l;
                lij = (l15 == j) ? l : r[j] / l;
            } else {
                t = __builtin_amdgcn_rsq(d);
                const double er = fma(-d * t, t, 1.0);
                t = fma(t * er, fma(0.375, er, 0.5), t);
                lij = r[j] * t;                             // column j of L (lane j: sqrt(d))
            }
            if (l15 == j) tl = t;
            r[j] = lij;
#pragma unroll
            for (int c = j + 1; c < 16; ++c) r[c] = fma(-lij, readlane_f64(lij, c), r[c]);
        }
        if (lane < 16) {
#pragma unroll
            for (int c = 0; c < 16; ++c) S[(o + l15) + (o + c) * LDI] = (c <= l15) ? r[c] : 0.0;
            invd[o + l15] = tl;
        }
    };
    auto trail = [&](int o, int bi, int bj) {   // S_ij -= L_ik L_jk^T
        v4d D;
#pragma unroll
        for (int r = 0; r < 4; ++r) D[r] = S[(16 * bi + l4 + 4 * r) + (16 * bj + l15) * LDI];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            D = mfma16(-S[(16 * bi + l15) + (o + 4 * kk + l4) * LDI], S[(16 * bj + l15) + (o + 4 * kk + l4) * LDI], D);
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(16 * bi + l4 + 4 * r) + (16 * bj + l15) * LDI] = D[r];
    };
    // x <- the forward substitution with the 16 x 16 diagonal sub-block at o:  x[j] = (x[j] - sum_{q<j} x[q] L_jq) / l_jj  (a row of
    // the panel times L_kk^-T, or a column of L_kk^-1 from a unit vector).  L_kk is read in two batches (columns 0-7 of rows
    // 1-7; then of rows 8-15 together with columns 8-15), each requested in full BEFORE the arithmetic that uses it: left to
    // itself the compiler reads every entry right before its multiply-add, one exposed LDS round trip per entry (panel 1.24 ->
    // 0.8 us per sub-block column; same operations in the same order)
    auto fwd16 = [&](int o, double (&x)[16], bool divide) {
        double dg[8], l1[8][8], l2[8][8], l3[8][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dg[j] = divide ? S[(o + j) + (o + j) * LDI] : invd[o + j];
#pragma unroll
            for (int c = j + 1; c < 8; ++c) l1[j][c] = S[(o + c) + (o + j) * LDI];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = divide ? x[j] / dg[j] : x[j] * dg[j];
#pragma unroll
            for (int c = j + 1; c < 8; ++c) x[c] = fma(-x[j], l1[j][c], x[c]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c) l2[j][c] = S[(o + 8 + c) + (o + j) * LDI];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            dg[j] = divide ? S[(o + 8 + j) + (o + 8 + j) * LDI] : invd[o + 8 + j];
#pragma unroll
            for (int c = j + 1; c < 8; ++c) l3[j][c] = S[(o + 8 + c) + (o + 8 + j) * LDI];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int c = 0; c < 8; ++c) x[8 + c] = fma(-x[j], l2[j][c], x[8 + c]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[8 + j] = divide ? x[8 + j] / dg[j] : x[8 + j] * dg[j];
#pragma unroll
            for (int c = j + 1; c < 8; ++c) x[8 + c] = fma(-x[8 + j], l3[j][c], x[8 + c]);
        }
    };
    __syncthreads();
    BASE_STAMP(1);
    if (wv == 0) chol(0);
    for (int k = 0; k < nblk; ++k) {
        const int o = 16 * k;
        __syncthreads();   // L_kk and 1 / l_jj are there, and so is every trailing update of the step before
        BASE_STAMP(8 + 3 * k);
        // ---- panel: X L_kk^T = S_:k by forward substitution, one row of the panel per thread (L_kk as LDS broadcasts):
        // no inverse of the diagonal sub-block on the critical path of the factorisation
        {
            const int row = o + 16 + tid;
            if (row < np) {
                double x[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = S[row + (o + c) * LDI];
                fwd16(o, x, XP);
#pragma unroll
                for (int c = 0; c < 16; ++c) S[row + (o + c) * LDI] = x[c];
            }
        }
        __syncthreads();
        BASE_STAMP(9 + 3 * k);
        // ---- trailing update with a look-ahead: first the block column k + 1 (all the next diagonal factorisation
        // and the next panel need), then wave 0 factors the next diagonal sub-block WHILE waves 1..7 update the rest
        const int t = nblk - k - 1;
        if (wv == 0) {
            // the next diagonal sub-block: its own update, then its factorisation, beside the other waves' updates (no
            // barrier between the block column k + 1 and the rest: nothing below reads what another wave writes here)
            if (t > 0) {
                trail(o, k + 1, k + 1);
                chol(o + 16);
            }
            BASE_STAMP(10 + 3 * k);
        } else {
            const int ncol = t - 1;   // block column k + 1 below its diagonal sub-block first (the next panel), then the rest
            for (int p = wv - 1; p < ncol + t * (t - 1) / 2; p += 7) {
                if (p < ncol) {
                    trail(o, k + 2 + p, k + 1);
                } else {
                    int i = 0, rem = p - ncol;   // -> (i, rem), row-major over the lower triangle of the (t - 1) x (t - 1) blocks right of column k + 1
                    while (rem > i) rem -= ++i;
                    trail(o, k + 2 + i, k + 2 + rem);
                }
            }
            // block column k of L is final since the panel: waves 1-7 store it (lower triangle, rows < nb; two adjacent rows
            // per lane, 16-byte stores) in the time they would otherwise wait for wave 0's factorisation
            v2d v[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int e = min(tid - 64 + 448 * m, 1023);
                v[m] = *reinterpret_cast<const v2d*>(&S[2 * (e & 63) + (o + (e >> 6)) * LDI]);
            }
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int e = tid - 64 + 448 * m, r = 2 * (e & 63), c = o + (e >> 6);
                if (e < 1024) {
                    double* ap = A + r + (long)c * lda;
                    if (lvec && c <= r && r + 1 < nb) {
                        *reinterpret_cast<v2d*>(ap) = v[m];
                    } else {
                        if (c <= r && r < nb) ap[0] = v[m].x;
                        if (c <= r + 1 && r + 1 < nb) ap[1] = v[m].y;
                    }
                }
            }
        }
    }
    __syncthreads();
    BASE_STAMP(2);
    if (bad) *info = 1;
    // ---- inverses of the diagonal sub-blocks, all at once (wave k: sub-block k): lane c solves L_kk w = e_c
    for (int k = wv; k < nblk; k += 8) {
        double w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = (i == l15) ? 1.0 : 0.0;
        {
            const int o = 16 * k;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                w[q] *= invd[o + q];
#pragma unroll
                for (int i = q + 1; i < 16; ++i) w[i] = fma(-S[(o + i) + (o + q) * LDI], w[q], w[i]);
            }
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) Wd[k][i + l15 * LDW] = (i >= l15) ? w[i] : 0.0;
        }
    }
    __syncthreads();
    BASE_STAMP(3);
    // ---- the diagonal sub-blocks of W (wave k: sub-block k; zero above the diagonal is the caller's contract)
    if (wv < nblk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 16 * wv + l15, col = 16 * wv + l4 + 4 * q;
            if (row < nb && col <= row) W[row + (long)col * ldw] = Wd[wv][l15 + (l4 + 4 * q) * LDW];
        }
    }
    // ---- W = L^-1 below the diagonal sub-blocks: block column w on wave w < 4, block column 11 - w on wave w >= 4 (waves w and
    // w + 4 share a SIMD: 168 + 0, 132 + 12, 100 + 28, 72 + 48 MFMAs per SIMD); W_ij (i > j) is parked in the upper block (j, i) of
    // the LDS image, L stays intact.  Right-looking within a block column: as soon as W_qj is known every row block below
    // it takes D_i += L_iq W_qj (one 4-MFMA chain per row block, independent of one another, all operands of the step
    // requested before its first MFMA), then W_(q+1)j = -W_(q+1)(q+1) D_(q+1).  The left-looking form (all of D_i, then W_ij,
    // one row block after the other, every operand read right before its MFMA) took 122 cycles per MFMA: 8.3 us for block
    // column 0.  Same sums in the same order.  Each wave stores its own blocks from the parked copy (rows along the lanes) one
    // step after it formed them, beside the next step's MFMAs: no store phase of its own.
    // block (j + 1 + ii, j) of W from its parked copy: element (16 i + a, 16 j + b) sits at S[(16 j + a) + (16 i + b) LDI]
    auto store_w = [&](int j, int ii) {
        const int a = 2 * (lane & 7), b0 = lane >> 3, i = j + 1 + ii;   // rows a, a + 1 of the block; columns b0, b0 + 8
        v2d v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) v[h] = *reinterpret_cast<const v2d*>(&S[(16 * j + a) + (16 * i + b0 + 8 * h) * LDI]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 16 * i + a, col = 16 * j + b0 + 8 * h;
            double* wp = W + row + (long)col * ldw;
            if (wvec && row + 1 < nb) {
                *reinterpret_cast<v2d*>(wp) = v[h];
            } else {
                if (row < nb) wp[0] = v[h].x;
                if (row + 1 < nb) wp[1] = v[h].y;
            }
        }
    };
    {
        const int j = wv < 4 ? wv : 11 - wv;
        if (j < nblk - 1) {
            v4d D[7];
#pragma unroll
            for (int ii = 0; ii < 7; ++ii) D[ii] = (v4d){0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 7; ++st) {   // step st: q = j + st, row blocks i = j + 1 + ii with ii >= st
                const int q = j + st;
                if (q < nblk - 1) {
                    double bq[4];
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        bq[kk] = (st == 0) ? Wd[j][(4 * kk + l4) + l15 * LDW] : S[(16 * j + 4 * kk + l4) + (16 * q + l15) * LDI];
                    double aq[7][4];
#pragma unroll
                    for (int ii = st; ii < 7; ++ii)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            aq[ii][kk] = S[(16 * min(j + 1 + ii, nblk - 1) + l15) + (16 * q + 4 * kk + l4) * LDI];
                    if (st > 0) store_w(j, st - 1);   // the block parked one step ago goes to memory beside this step's MFMAs
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int ii = st; ii < 7; ++ii)
                            if (j + 1 + ii < nblk) D[ii] = mfma16(aq[ii][kk], bq[kk], D[ii]);
                    // -W_ii * D: D's rows (l4 + 4 r) are the contraction index of the second product, taken in that order
                    const int i = q + 1;
                    v4d R = {0, 0, 0, 0};
#pragma unroll
                    for (int r = 0; r < 4; ++r) R = mfma16(-Wd[i][l15 + (l4 + 4 * r) * LDW], D[st][r], R);
#pragma unroll
                    for (int r = 0; r < 4; ++r) S[(16 * j + l4 + 4 * r) + (16 * i + l15) * LDI] = R[r];
                }
            }
            store_w(j, nblk - 2 - j);   // the last block of the column
        }
    }
    BASE_STAMP_WAVE(40 + wv);   // end of this wave's share of the block inverse
#ifdef GPX_BASE_TIMING
    if (lane == 0) g_base_t[50 + wv] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID: SIMD of this wave in bits 5:4
#endif
    BASE_STAMP(4);
    BASE_STAMP(5);
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

__global__ void __launch_bounds__(1024) logdiag_sum_kernel(const double* __restrict__ A, long lda, long n, double* __restrict__ out) {
    // one workgroup (fixed summation order); 1024 threads: every diagonal entry is a cache line of its own
    __shared__ double red[1024];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) s += log(A[i + i * lda]);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
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

inline long inv_split(long n) { return ((n / 2 + NBI - 1) / NBI) * NBI; }

// Factor and invert in one recursion: on return the diagonal blocks of A hold those of L (the
// off-diagonal blocks of A are NOT L) and W = L^-1.  L21 lives in W's (2,1) block until that block
// is overwritten by W21, so there is no triangular solve and no small kernel besides the base block.
// (Running T = L21 W11 on a side stream beside the trailing recursion was measured and dropped: the compute-bound
// side product holds every CU until it ends, equal or low stream priority alike, so the chain of small launches
// it was meant to hide waits behind it: exact n = 8192 17.1 -> 23.2 ms.)
bool g_base64 = false;   // GPX_BASE64=1: the 64-row scalar base block and its split (A/B measurements)
bool g_exact_pivot = false;   // GPX_EXACT_PIVOT=1: correctly rounded pivots in the 128-row base block (A/B measurements)

int potrf_inv_rec(hipStream_t st, long n, double* A, long lda, double* W, long ldw, double* T, int* info) {
    if (g_base64 && n <= NB) {
        hipLaunchKernelGGL(potrf_base_tile_kernel, dim3(1), dim3(256), 0, st, A, lda, (int)n, W, ldw, 0, info);
        return GPX_OK;
    }
    if (!g_base64 && n <= NBI) {
        if (g_exact_pivot) hipLaunchKernelGGL(potrf_inv_base_kernel<true>, dim3(1), dim3(512), 0, st, A, lda, (int)n, W, ldw, info);
        else hipLaunchKernelGGL(potrf_inv_base_kernel<false>, dim3(1), dim3(512), 0, st, A, lda, (int)n, W, ldw, info);
        return GPX_OK;
    }
    const long n1 = g_base64 ? split(n) : inv_split(n), n2 = n - n1;
    double* W21 = W + n1;
    double* W22 = W + n1 + n1 * ldw;
    int rc = potrf_inv_rec(st, n1, A, lda, W, ldw, T, info);
    if (rc) return rc;
    // L21 = A21 * W11^T   (B(n,k) = W11[n + k*ldw], zero for k > n)
    rc = gpx_gemm(st, 0, 0, n2, n1, n1, 1.0, A + n1, lda, W, ldw, 0.0, W21, ldw, GEMM_KHI_N);
    if (rc) return rc;
    // A22 -= L21 L21^T and T = L21 W11 both wait for L21 only: ONE launch below 4096 rows (gpx_gemm_pair), where a product
    // is a few workgroups bound by the launch-to-launch step.  T lives in the A21 block (off-diagonal blocks of A are
    // scratch by contract) until the recursion on A22 has delivered W22.
    double* T21 = A + n1;
    rc = gpx_gemm_pair(st, n2, n2, n1, -1.0, W21, ldw, W21, ldw, 1.0, A + n1 + n1 * lda, lda, GEMM_LOWER,
                       n2, n1, n1, 1.0, W21, ldw, W, ldw, 0.0, T21, lda, GEMM_KLO_N);
    if (rc) return rc;
    rc = potrf_inv_rec(st, n2, A + n1 + n1 * lda, lda, W22, ldw, T, info);
    if (rc) return rc;
    return gpx_gemm(st, 0, 1, n2, n1, n2, -1.0, W22, ldw, T21, lda, 0.0, W21, ldw, GEMM_KHI_M | GEMM_HEAVY_LAST);
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

static void chol_env_once() {
    static const bool init = [] {
        const char* e = getenv("GPX_BASE64");
        g_base64 = e && e[0] == '1';
        e = getenv("GPX_EXACT_PIVOT");
        g_exact_pivot = e && e[0] == '1';
        return true;
    }();
    (void)init;
}

int gpx_potrf_inv(hipStream_t st, long n, double* A, long lda, double* W, long ldw, double* T, int* info) {
    if (n <= 0) return GPX_OK;
    chol_env_once();
    const int rc = potrf_inv_rec(st, n, A, lda, W, ldw, T, info);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

// (Measured and dropped, round 3: Out11 = W11^T W11 of the final W^T W formed on the side stream BESIDE the recursion on A22
// at n = 8192, as block products of at most 128 workgroups so that half of the CUs stay empty for the chain, the rest
// of W^T W as three products afterwards.  The base blocks keep their 36 us, but the chain's small products - bound by
// the latency of their dependent global loads - run 2-3 times slower while the side product streams through memory
// (128-row level 5.2 -> ~14 us), and the three remaining 4096-row pieces take as long as the single 8192-row launch
// they replace (2.75 ms: 60 / 69 / 38 TFLOP/s against 66.6): 12.6 -> 14.0 ms, same box.)

long gpx_trtri_scratch(long n) {   // covers gpx_trtri (64-row base) and gpx_potrf_inv (128-row base)
    if (n <= NB) return 16;
    const long n1 = split(n), m1 = inv_split(n);
    const long a = gpx_ld(n - n1) * n1 + 16, b = n > NBI ? gpx_ld(n - m1) * m1 + 16 : 16;
    return a > b ? a : b;
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
    // Out = W^T W on the lower triangle, W lower triangular (stored W(k,m) at W[k + m*ldw], k >= m).
    // One launch with k >= max(m, n) has trip counts that depend on BOTH tile indices, which the GEMM's
    // square tile order cannot keep in step; above 16384 rows split instead:
    //   Out11 = W11^T W11 + W21^T W21,  Out21 = W22^T W21,  Out22 = W22^T W22
    // which puts 3/4 of the flops of each level into a full-K LOWER product and a product whose trip
    // count depends on the M tile only.
    if (n <= 16384) return gpx_gemm(st, 1, 1, n, n, n, 1.0, W, ldw, W, ldw, 0.0, Out, ldo, GEMM_LOWER | GEMM_KLO_M);
    const long n1 = (n / 2 + 1023) / 1024 * 1024, n2 = n - n1;
    const double* W21 = W + n1;
    const double* W22 = W + n1 + n1 * ldw;
    int rc = gpx_lauum(st, n1, W, ldw, Out, ldo);
    if (rc) return rc;
    rc = gpx_gemm(st, 1, 1, n1, n1, n2, 1.0, W21, ldw, W21, ldw, 1.0, Out, ldo, GEMM_LOWER);
    if (rc) return rc;
    rc = gpx_gemm(st, 1, 1, n2, n1, n2, 1.0, W22, ldw, W21, ldw, 0.0, Out + n1, ldo, GEMM_KLO_M);
    if (rc) return rc;
    return gpx_lauum(st, n2, W22, ldw, Out + n1 + n1 * ldo, ldo);
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
    hipLaunchKernelGGL(logdiag_sum_kernel, dim3(1), dim3(1024), 0, st, A, lda, n, out);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}
