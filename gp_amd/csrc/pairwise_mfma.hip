// pairwise_mfma.hip — the n x m kernel matrix Kxu and the fused gradient trace over it ON THE MATRIX CORES
// (v_mfma_f64_16x16x4_f64, gfx950), for the sparse models (FITC.cpp:204, 269-308; VFE.cpp:168, 213-237;
// Cov/CovSEard.cpp:7-22).
//
//  * squared distances by the norm expansion the reference keeps as a commented variant (util.cpp:67-69):
//      r2_ab = |x_a|^2 + |u_b|^2 - 2 x_a.u_b  =  sum_c Xa[c][a] * Ua[c][b]
//    with AUGMENTED operands  Xa = [x~ ; |x~|^2 ; 1],  Ua = [-2 u~ ; 1 ; |u~|^2]  (x~, u~ = lengthscale-scaled
//    inputs, centred on the mean inducing point: distances are translation invariant and the centring keeps
//    |x~|^2 of the order of r2, so the cancellation costs a few ulps of r2, not of |x|^2).  One contraction of
//    depth dk = roundup(d + 2, 4) per pair, no vector-unit work for r2 at all; r2 is clamped at 0.
//  * gradient sums: with Q = W o K per tile (one exp per pair stays on the vector unit)
//      sum_ab Q_ab (x~_ia - u~_ib)^2 = sum_a x~_ia^2 (Q 1)_a + sum_a x~_ia (Q [-2 U~])_ai + sum_b u~_ib^2 (Q^T 1)_b
//    The MFMA accumulator layout of the r2 tile (lane holds [b = 4r + lane/16][a = lane%16]) IS the operand
//    layout of a second MFMA that contracts over b: Q goes from the first product into the second without
//    leaving its registers, and [Q(-2U~) | Q1] (d + 1 columns) accumulates over all inducing points of a row
//    block.  The column sums Q^T 1 are a 16-lane reduction per tile.  3 d vector operations per pair become
//    2 (dk + 16 ceil((d+1)/16)) matrix flops per pair.
//
// Two trace kernels: trace_mfma_kernel (below) and, for d > 30 with one weight matrix (the VFE trace),
// trace_mfma8_kernel: one 8-wave workgroup per CU, direct-to-LDS staging one tile ahead, one barrier per tile.
// Workgroup = 4 waves = 128 training rows (32 per wave: two interleaved 16-row MFMA tiles, so that a lane's two
// rows are adjacent in memory and every weight load is 16 B); inducing points go through LDS 64 at a time in two
// images (row-major for the first product, transposed for the second), both conflict-free for their fragment
// reads.  k(X,X) of the exact path and Kuu keep the direct-difference kernels of pairwise.hip (exact sf2
// diagonal, as util.cpp:64-75 computes it).
#include "gpx_internal.h"
#include "kernels.h"
#include <cstdlib>

namespace {

struct MfmaPairArgs {
    const double* Xa; long ldxa; long n;    // [dk][ldxa], ldxa = roundup(n, 128), zero beyond n
    const double* Ua; long ldua; long m;    // [dk][ldua], ldua = roundup(m, 64), zero beyond m
    const double* UaT;                      // [ldua][N2]: rows 0..d of Ua, transposed, zero padded
    int d;
    double sf2;
    // cov build
    double* out; long ldo;
    // trace: w_ab = c1_a * M1_ab + c2_a * M2_ab + u_a * v_b
    const double* M1; long ldm1; const double* c1;
    const double* M2; long ldm2; const double* c2;
    const double* u; const double* v;
    double* cpart;   // [gridDim][ldua] column sums of Q per workgroup (zero on entry)
    double* spart;   // [gridDim][N2 + 1]
    int nblk;        // row blocks of the trace kernel (128 or 64 rows)
    // row-panel layout of out / M1 / M2 (GemmPanels in gpx_internal.h): rows in panels of pan (a multiple of 128, 0 = plain),
    // element (a, b) at (a / pan) * pst + a % pan + b * ld; a row block never straddles two panels
    long pan, pst;
};
// base shift of the row block that starts at row a0 (see MfmaPairArgs::pan)
__device__ __forceinline__ long panel_shift(const MfmaPairArgs& g, long a0) { return g.pan ? (a0 / g.pan) * (g.pst - g.pan) : 0; }

template <int KK>
struct Cfg {
    static constexpr int DK = 4 * KK;                      // depth of the r2 contraction (>= d + 2)
    static constexpr int JT = (4 * KK - 1 + 15) / 16;      // 16-column tiles of [Q(-2U) | Q1] (>= d + 1 columns)
    static constexpr int N2 = 16 * JT;
    // inducing points per LDS tile: 64, or 32 for the deep contractions, whose two images of a 64-point tile
    // (79 KB) would leave room for ONE workgroup per CU only
    static constexpr int TB = KK <= 8 ? 64 : 32;
    static constexpr int S1 = TB + 16;                     // row stride of image 1 [DK][TB]: = 128 B mod 256 B
    static constexpr int S2 = (N2 % 32 == 16) ? N2 : N2 + 16;  // row stride of image 2 [TB][N2]: = 128 B mod 256 B
};

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// exp(x) for x <= 0 (the kernel's -r2/2): k = rint(x log2 e), r = x - k ln 2 in two pieces, degree-12 Taylor polynomial on
// |r| <= 0.347 (truncation 1.7e-16 relative), ldexp.  No overflow / NaN branches: the argument is clamped at -800, where the
// result is 0 as it is for the library exp (the library version spends 6 compare / select operations per value on them).
__device__ __forceinline__ double exp_nonpos(double x) {
    x = fmax(x, -800.0);
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;           // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);       // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);       // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);       // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);       // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);       // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);       // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);       // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);       // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);       // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// sum over the 16 lanes of a DPP row, every lane ends with the same bits (ds_bpermute shuffles would put
// four dependent LDS round trips per tile on the critical path of a wave that has one partner to hide them)
template <int CTRL>
__device__ __forceinline__ double dpp_add(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return x + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double x) {
    x = dpp_add<0xB1>(x);    // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);    // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);   // row_half_mirror: the other quad of the half
    return dpp_add<0x140>(x);  // row_mirror: the other half
}

// stage one tile of 64 inducing points: image 1 sU[c][b] (first product: A operand rows = b, k = c),
// image 2 sUT[b][c] (second product: B operand k = b, columns = c)
template <int KK, bool WITH_T>
__device__ __forceinline__ void stage_u(const MfmaPairArgs& g, long b0, double* __restrict__ sU, double* __restrict__ sUT) {
    using C = Cfg<KK>;
    for (int idx = threadIdx.x; idx < C::DK * C::TB; idx += 256) {
        const int c = idx / C::TB, b = idx % C::TB;
        sU[c * C::S1 + b] = g.Ua[(long)c * g.ldua + b0 + b];
    }
    if (WITH_T) {
        for (int idx = threadIdx.x; idx < C::TB * C::N2; idx += 256) {
            const int b = idx / C::N2, c = idx % C::N2;
            sUT[b * C::S2 + c] = g.UaT[(b0 + b) * C::N2 + c];
        }
    }
}

// K_ab = sf2 exp(-r2/2) for one row block of 128 x all inducing points
template <int KK>
__global__ __launch_bounds__(256, 2) void cov_mfma_kernel(const MfmaPairArgs g) {
    using C = Cfg<KK>;
    __shared__ __attribute__((aligned(16))) double sU[C::DK * C::S1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(g.out) & 15) == 0) && (g.ldo % 2 == 0) && (g.pst % 2 == 0);
    // one 128-row block per workgroup, or (grid < nblk: a launch that must leave CUs empty for a chain of small kernels
    // running beside it, sparse.hip) a stride over the row blocks
    for (int blk = blockIdx.x; blk < g.nblk; blk += gridDim.x) {
    const long a0 = (long)blk * 128 + 32 * wv + 2 * l15;   // this lane's two rows: a0, a0 + 1
    double xf[2][KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
        const v2d x = *reinterpret_cast<const v2d*>(g.Xa + (long)(4 * kk + l4) * g.ldxa + a0);
        xf[0][kk] = x.x;
        xf[1][kk] = x.y;
    }
    const bool rows_in = (long)blk * 128 + 128 <= g.n;
    double* const outb = g.out + panel_shift(g, (long)blk * 128);
    for (long b0 = 0; b0 < g.m; b0 += C::TB) {
        __syncthreads();
        stage_u<KK, false>(g, b0, sU, nullptr);
        __syncthreads();
        const bool full = rows_in && vec && b0 + C::TB <= g.m;
#pragma unroll
        for (int bt = 0; bt < C::TB / 16; ++bt) {
            v4d D0 = {0, 0, 0, 0}, D1 = {0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const double ua = sU[(4 * kk + l4) * C::S1 + 16 * bt + l15];
                D0 = mfma(ua, xf[0][kk], D0);
                D1 = mfma(ua, xf[1][kk], D1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long b = b0 + 16 * bt + 4 * r + l4;
                const double k0 = g.sf2 * exp_nonpos(-0.5 * fmax(D0[r], 0.0));
                const double k1 = g.sf2 * exp_nonpos(-0.5 * fmax(D1[r], 0.0));
                double* o = outb + a0 + b * g.ldo;
                if (full) {
                    *reinterpret_cast<v2d*>(o) = (v2d){k0, k1};
                } else if (b < g.m) {
                    if (a0 < g.n) o[0] = k0;
                    if (a0 + 1 < g.n) o[1] = k1;
                }
            }
        }
    }
    }
}

// the fused gradient trace: per workgroup partial sums of
//   S_c = sum_a x~_ca (x~_ca (Q1)_a + (Q[-2U~])_ac)   c < d,      S_K = sum_ab Q_ab,      colsum_b = sum_a Q_ab
// NS = 16-row MFMA tiles per wave (rows per workgroup = 64 NS): 2 while the X fragments and the [Q(-2U) | Q1]
// accumulators of two tiles fit the register file beside the exp, 1 for the deep contractions (d > 30)
template <int KK, bool HAS_M2, int NS>
__global__ __launch_bounds__(256, 2) void trace_mfma_kernel(const MfmaPairArgs g) {
    using C = Cfg<KK>;
    constexpr int JT = C::JT;
    constexpr int RB = 64 * NS;   // rows per workgroup
    __shared__ __attribute__((aligned(16))) double sU[C::DK * C::S1];
    constexpr int TB = C::TB;
    __shared__ __attribute__((aligned(16))) double sUT[TB * C::S2];
    __shared__ double sv[TB];
    __shared__ double csum[4][TB];
    __shared__ double red[4][C::N2 + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int d = g.d;
    double accS[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) accS[jt] = 0.0;
    double accK = 0.0;
    double* cp = g.cpart + (long)blockIdx.x * g.ldua;
    const bool vec = NS == 1 || (((reinterpret_cast<uintptr_t>(g.M1) & 15) == 0) && (g.ldm1 % 2 == 0) && (g.pst % 2 == 0) &&
                                 (!HAS_M2 || (((reinterpret_cast<uintptr_t>(g.M2) & 15) == 0) && (g.ldm2 % 2 == 0))));

    for (int blk = blockIdx.x; blk < g.nblk; blk += gridDim.x) {
        const long a0 = (long)blk * RB + 16 * NS * wv + NS * l15;   // this lane's NS adjacent rows in the r2 tiles
        const long psh = panel_shift(g, (long)blk * RB);
        const double* const M1b = g.M1 + psh;
        const double* const M2b = HAS_M2 ? g.M2 + psh : nullptr;
        double xf[NS][KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const double* px = g.Xa + (long)(4 * kk + l4) * g.ldxa + a0;
            if (NS == 2) {
                const v2d x = *reinterpret_cast<const v2d*>(px);
                xf[0][kk] = x.x;
                xf[NS - 1][kk] = x.y;
            } else {
                xf[0][kk] = px[0];
            }
        }
        double c1a[NS], c2a[NS], ua[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bool in = a0 + s < g.n;
            c1a[s] = (in && g.c1) ? g.c1[a0 + s] : 1.0;
            c2a[s] = (HAS_M2 && in && g.c2) ? g.c2[a0 + s] : 1.0;
            ua[s] = (in && g.u) ? g.u[a0 + s] : 0.0;
        }
        v4d acc2[NS][JT];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) acc2[s][jt] = (v4d){0, 0, 0, 0};
        const bool rows_in = (long)blk * RB + RB <= g.n;

        for (long b0 = 0; b0 < g.m; b0 += TB) {
            const bool full = rows_in && vec && b0 + TB <= g.m;
            // single weight matrix, one row per lane: all 16 weights of the tile are requested BEFORE the
            // inducing points are staged, so that the HBM latency hides behind the staging and the first products
            constexpr bool PF = !HAS_M2 && NS == 1;
            double wpf[PF ? TB / 4 : 1];
            if (PF) {
#pragma unroll
                for (int i = 0; i < TB / 4; ++i) {
                    const long b = b0 + 4 * i + l4;   // i = 4 bt + r
                    wpf[i] = (full || (b < g.m && a0 < g.n)) ? M1b[a0 + b * g.ldm1] : 0.0;
                }
            }
            __syncthreads();   // the previous tile's fragments are consumed, its column sums are in csum
            if (b0 > 0 && threadIdx.x < TB)
                cp[b0 - TB + threadIdx.x] += (csum[0][threadIdx.x] + csum[1][threadIdx.x]) + (csum[2][threadIdx.x] + csum[3][threadIdx.x]);
            stage_u<KK, true>(g, b0, sU, sUT);
            if (threadIdx.x < TB) sv[threadIdx.x] = (g.v && b0 + threadIdx.x < g.m) ? g.v[b0 + threadIdx.x] : 0.0;
            __syncthreads();
#pragma unroll
            for (int bt = 0; bt < TB / 16; ++bt) {
                // weights of this 16-column group first: their latency hides behind the r2 product
                double w1[4][NS], w2[4][NS];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (PF) {
                        w1[r][0] = wpf[4 * bt + r];
                        continue;
                    }
                    const long b = b0 + 16 * bt + 4 * r + l4;
                    const double* p1 = M1b + a0 + b * g.ldm1;
                    const double* p2 = HAS_M2 ? M2b + a0 + b * g.ldm2 : nullptr;
                    if (full) {
                        if (NS == 2) {
                            const v2d t = *reinterpret_cast<const v2d*>(p1);
                            w1[r][0] = t.x;
                            w1[r][NS - 1] = t.y;
                            if (HAS_M2) {
                                const v2d t2 = *reinterpret_cast<const v2d*>(p2);
                                w2[r][0] = t2.x;
                                w2[r][NS - 1] = t2.y;
                            }
                        } else {
                            w1[r][0] = p1[0];
                            if (HAS_M2) w2[r][0] = p2[0];
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < NS; ++s) {
                            const bool in = b < g.m && a0 + s < g.n;
                            w1[r][s] = in ? p1[s] : 0.0;
                            if (HAS_M2) w2[r][s] = in ? p2[s] : 0.0;
                        }
                    }
                }
                v4d D[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) D[s] = (v4d){0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const double uf = sU[(4 * kk + l4) * C::S1 + 16 * bt + l15];
#pragma unroll
                    for (int s = 0; s < NS; ++s) D[s] = mfma(uf, xf[s][kk], D[s]);
                }
                double q[NS][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int bl = 16 * bt + 4 * r + l4;
                    const double vb = sv[bl];
                    // beyond n / m the operands are zero, so K = sf2 there: the weights must vanish instead
                    const bool bin = b0 + bl < g.m;
                    double cs = 0.0;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        double w = c1a[s] * w1[r][s];
                        if (HAS_M2) w = fma(c2a[s], w2[r][s], w);
                        w = fma(ua[s], vb, w);
                        if (!bin) w = 0.0;
                        q[s][r] = w * (g.sf2 * exp_nonpos(-0.5 * fmax(D[s][r], 0.0)));
                        cs += q[s][r];
                    }
                    // column sum over this wave's rows
                    cs = row16_sum(cs);
                    if (l15 == 0) csum[wv][bl] = cs;
                }
                // second product: [Q(-2U~) | Q1] += Q * UaT, contraction over the 16 inducing points of this group
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        const double ub = sUT[(16 * bt + 4 * r + l4) * C::S2 + 16 * jt + l15];
#pragma unroll
                        for (int s = 0; s < NS; ++s) acc2[s][jt] = mfma(q[s][r], ub, acc2[s][jt]);
                    }
            }
        }
        __syncthreads();
        if (threadIdx.x < TB) {
            const long bl = (g.m - 1) / TB * TB;
            cp[bl + threadIdx.x] += (csum[0][threadIdx.x] + csum[1][threadIdx.x]) + (csum[2][threadIdx.x] + csum[3][threadIdx.x]);
        }
        // end of the row block: acc2[s][jt][r] = row a = base + NS (l4 + 4 r) + s, column c = 16 jt + l15
        const int jtd = d >> 4, ld = d & 15;   // column d holds (Q 1)_a
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double q1a = 0.0;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
                    if (jt == jtd) q1a = acc2[s][jt][r];
                q1a = __shfl(q1a, (lane & 48) | ld, 64);
                const long a = (long)blk * RB + 16 * NS * wv + NS * (l4 + 4 * r) + s;
                if (l15 == 0) accK += q1a;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const int c = 16 * jt + l15;
                    if (c < d) {
                        const double x = g.Xa[(long)c * g.ldxa + a];   // zero beyond n
                        accS[jt] = fma(x, fma(x, q1a, acc2[s][jt][r]), accS[jt]);
                    }
                }
            }
    }
    // workgroup partials: column c = 16 jt + l15 summed over l4 and the four waves
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        double s = accS[jt];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (l4 == 0) red[wv][16 * jt + l15] = s;
    }
    {
        double s = accK;
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lane == 0) red[wv][C::N2] = s;
    }
    __syncthreads();
    if (threadIdx.x <= C::N2)
        g.spart[(long)blockIdx.x * (C::N2 + 1) + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// The same trace for the deep contractions (d > 30) as ONE 8-wave workgroup per CU: 128 rows per workgroup
// (16 per wave), inducing points 32 at a time through a TWO-stage LDS ring that is filled by direct-to-LDS
// loads (global_load_lds: no staging registers, which this kernel does not have) one tile ahead, one barrier
// per tile, the next tile's weights requested right after that barrier so that nothing the tile computes with
// is still in flight (hipcc drains every outstanding load at the first use of an ordinary load's result while a
// direct-to-LDS load is pending).  The column sums stay in LDS for the whole launch.  The 4-wave kernel above
// stages synchronously between two barriers and spends 60 % of its wave cycles waiting at d = 50.
constexpr int CACC_MAX = 4096;   // inducing points (padded) whose column sums fit the LDS accumulator

template <int KK>
__device__ __forceinline__ void dma_tile(const MfmaPairArgs& g, int b0, double* sU_s, double* sUT_s, double* sv_s, int wv, int lane) {
    using C = Cfg<KK>;
    constexpr int TB = 32, S1 = TB + 16;
    constexpr int CH = (2 * C::N2 + 63) / 64;            // 256-B pieces per transposed row
    constexpr int NQ = C::DK + CH * TB + 1;              // wave-loads per tile
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int ldua = (int)g.ldua;                        // <= CACC_MAX: every offset below fits 32 bits
    asm volatile("" : "+s"(wv));                         // the 15 row offsets are recomputed per tile (scalar unit), not kept in SGPRs that spill
    const char* pU = reinterpret_cast<const char*>(g.Ua + b0) + lane * 4;
    const char* pT = reinterpret_cast<const char*>(g.UaT + b0 * C::N2) + lane * 4;
#pragma unroll
    for (int i = 0; i < (NQ + 7) / 8; ++i) {
        const int q = wv + 8 * i;   // wave-uniform
        if (q < C::DK) {            // one row of image 1: 32 inducing points of augmented dimension q
            __builtin_amdgcn_global_load_lds((gptr_t)(pU + (q * ldua) * 8), (lptr_t)(sU_s + q * S1), 4, 0, 0);
        } else if (q < C::DK + CH * TB) {
            const int r = q - C::DK, b = r / CH, h = r % CH;
            if (CH * 64 == 2 * C::N2 || lane + 64 * h < 2 * C::N2)
                __builtin_amdgcn_global_load_lds((gptr_t)(pT + (b * C::N2 + 32 * h) * 8), (lptr_t)(sUT_s + b * C::S2 + 32 * h), 4, 0, 0);
        } else if (q == C::DK + CH * TB && g.v) {
            __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char*>(g.v + b0) + lane * 4), (lptr_t)sv_s, 4, 0, 0);
        }
    }
}

template <int KK, bool HAS_M2>
__global__ __launch_bounds__(512) void trace_mfma8_kernel(const MfmaPairArgs g) {
    using C = Cfg<KK>;
    constexpr int JT = C::JT, TB = 32, S1 = TB + 16, RB = 128, NWV = 8;
    __shared__ __attribute__((aligned(16))) double sU[2][C::DK * S1];
    __shared__ __attribute__((aligned(16))) double sUT[2][TB * C::S2];
    __shared__ __attribute__((aligned(16))) double sv[2][TB];
    __shared__ double csum[2][NWV][TB];
    __shared__ double cacc[CACC_MAX];
    __shared__ double red[NWV][C::N2 + 1];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, l4 = lane >> 4;
    const int d = g.d;
    double accS[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) accS[jt] = 0.0;
    double accK = 0.0;
    for (int i = threadIdx.x; i < CACC_MAX; i += 512) cacc[i] = 0.0;
    if (threadIdx.x < 2 * TB) sv[threadIdx.x / TB][threadIdx.x % TB] = 0.0;   // stays zero when there is no u v^T term

    const int ntile = (int)((g.m + TB - 1) / TB);
    const int nmine = (int)blockIdx.x < g.nblk ? (g.nblk - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int m = (int)g.m;
    // weights of tile tl of row block blk: this lane's 8 (b = b0 + 4 i + l4, a)
    // uniform 64-bit base (scalar registers, advanced per tile by the scalar unit) + ONE loop-invariant 32-bit lane offset
    const unsigned wo1 = (unsigned)(16 * wv + l15) + (unsigned)l4 * (unsigned)g.ldm1;   // 3 ldm1 + 127 < 2^32 for any matrix that fits the device
    const unsigned wo2 = HAS_M2 ? (unsigned)(16 * wv + l15) + (unsigned)l4 * (unsigned)g.ldm2 : 0u;
    auto load_w = [&](int blk, int tl, double (&w1)[TB / 4], double (&w2)[TB / 4]) {
        const int b0 = tl * TB;
        const long a = (long)blk * RB + 16 * wv + l15;
        const long psh = panel_shift(g, (long)blk * RB);
        const double* s1 = g.M1 + psh + (long)blk * RB + (long)b0 * g.ldm1;   // wave-uniform
        const double* s2 = HAS_M2 ? g.M2 + psh + (long)blk * RB + (long)b0 * g.ldm2 : nullptr;
        if ((long)blk * RB + RB <= g.n && b0 + TB <= m) {   // wave-uniform: the interior takes no per-load guard
#pragma unroll
            for (int i = 0; i < TB / 4; ++i) {
                w1[i] = (s1 + (long)(4 * i) * g.ldm1)[wo1];
                w2[i] = HAS_M2 ? (s2 + (long)(4 * i) * g.ldm2)[wo2] : 0.0;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TB / 4; ++i) {
                const bool in = b0 + 4 * i + l4 < m && a < g.n;
                w1[i] = in ? (s1 + (long)(4 * i) * g.ldm1)[wo1] : 0.0;
                w2[i] = (HAS_M2 && in) ? (s2 + (long)(4 * i) * g.ldm2)[wo2] : 0.0;
            }
        }
    };
    // (Three weight register sets in rotation, i.e. a tile's weights requested TWO tiles ahead with the tile body unrolled
    // three times, were measured: slower - FITC 5.13 -> 5.39 ms, VFE 51.5 -> 58.4 ms at 255 VGPRs with spills; not kept.)
    double wc1[TB / 4], wc2[TB / 4], wn1[TB / 4], wn2[TB / 4];
    double xf[KK];
    double c1a = 1.0, c2a = 1.0, ua = 0.0;
    v4d acc2[JT];
    __syncthreads();   // cacc / sv initialised before the first direct-to-LDS load may land in sv
    int blk = blockIdx.x, tl = 0, st = 0;   // row block, tile and LDS stage of the current iteration
    if (nmine > 0) {
        dma_tile<KK>(g, 0, sU[0], sUT[0], sv[0], wv, lane);
        load_w(blk, 0, wc1, wc2);
    }
    for (int k = 0; k < nmine;) {
        const int b0 = tl * TB;
        if (tl == 0) {   // new row block: X fragments, row coefficients, fresh accumulators
            const long a0 = (long)blk * RB + 16 * wv + l15;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) xf[kk] = g.Xa[(long)(4 * kk + l4) * g.ldxa + a0];
            const bool in = a0 < g.n;
            c1a = (in && g.c1) ? g.c1[a0] : 1.0;
            c2a = (HAS_M2 && in && g.c2) ? g.c2[a0] : 1.0;
            ua = (in && g.u) ? g.u[a0] : 0.0;
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) acc2[jt] = (v4d){0, 0, 0, 0};
        }
        __syncthreads();   // this tile has landed (the barrier drains this wave's loads); stage st ^ 1 and csum[st] are free
        if ((k > 0 || tl > 0) && threadIdx.x < TB) {   // column sums of the previous tile -> LDS accumulator
            const int pt = tl > 0 ? tl - 1 : ntile - 1;
            double sum = 0.0;
#pragma unroll
            for (int w = 0; w < NWV; ++w) sum += csum[st ^ 1][w][threadIdx.x];
            cacc[pt * TB + threadIdx.x] += sum;
        }
        const bool last_tile = tl == ntile - 1;
        const int ntl = last_tile ? 0 : tl + 1, nblkidx = last_tile ? blk + (int)gridDim.x : blk, nk = last_tile ? k + 1 : k;
        if (nk < nmine) {
            dma_tile<KK>(g, ntl * TB, sU[st ^ 1], sUT[st ^ 1], sv[st ^ 1], wv, lane);
            load_w(nblkidx, ntl, wn1, wn2);
        }
        const double* U1 = sU[st];
        const double* U2 = sUT[st];
        // Both r2 chains first, then per 16-point group Q = W o K on the vector unit and the second product.  (An
        // explicit MFMA / VALU interleave of the groups (sched_group_barrier pipeline) was measured: no gain, 52.5 vs
        // 53.4 ms at d = 50.  PMC: MFMA busy 48 %, ~560 vector instructions per wave and tile of which ~210 fp64: the
        // fp64 MFMA and the fp64 vector operations of a SIMD share its fp64 ALUs (same 78.6 TFLOP/s peak for both on
        // gfx950), so their times add: 3712 + ~3000 cycles per wave and tile, two waves per SIMD = the measured 6.3 us.)
        auto mfma1 = [&](int bt, v4d& Dg) {
            double u1[KK];   // the operands of one chain up front: their LDS latency is paid once per chain, not per MFMA
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) u1[kk] = U1[(4 * kk + l4) * S1 + 16 * bt + l15];
            Dg = (v4d){0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) Dg = mfma(u1[kk], xf[kk], Dg);
        };
        auto expq = [&](int bt, const v4d& Dg, double (&q)[4]) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int bl = 16 * bt + 4 * r + l4;
                double w = c1a * wc1[4 * bt + r];
                if (HAS_M2) w = fma(c2a, wc2[4 * bt + r], w);
                w = fma(ua, sv[st][bl], w);
                if (b0 + bl >= m) w = 0.0;   // beyond m the operands are zero, K = sf2 there: the weight must vanish
                q[r] = w * (g.sf2 * exp_nonpos(-0.5 * fmax(Dg[r], 0.0)));
                csum[st][wv][bl] = row16_sum(q[r]);   // the 16 lanes of a row hold the same sum: same value, same address
            }
        };
        auto mfma2 = [&](int bt, const double (&q)[4]) {
            double u2[4][JT];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) u2[r][jt] = U2[(16 * bt + 4 * r + l4) * C::S2 + 16 * jt + l15];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) acc2[jt] = mfma(q[r], u2[r][jt], acc2[jt]);
        };
        static_assert(TB == 32, "two groups per tile");
        {
            v4d D0, D1;
            double q0[4], q1[4];
            mfma1(0, D0);
            mfma1(1, D1);
            expq(0, D0, q0);
            mfma2(0, q0);
            expq(1, D1, q1);
            mfma2(1, q1);
        }
        if (last_tile) {   // end of the row block: acc2[jt][r] = row base + l4 + 4 r, column 16 jt + l15
            const int jtd = d >> 4, ld = d & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double q1a = 0.0;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
                    if (jt == jtd) q1a = acc2[jt][r];
                q1a = __shfl(q1a, (lane & 48) | ld, 64);
                const long a = (long)blk * RB + 16 * wv + l4 + 4 * r;
                if (l15 == 0) accK += q1a;
                double x[JT];   // all loads of the row first: one wait, not JT
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const int c = 16 * jt + l15;
                    x[jt] = g.Xa[(long)(c < d ? c : 0) * g.ldxa + a];
                }
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const double t = fma(x[jt], fma(x[jt], q1a, acc2[jt][r]), accS[jt]);
                    accS[jt] = (16 * jt + l15 < d) ? t : accS[jt];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TB / 4; ++i) {
            wc1[i] = wn1[i];
            wc2[i] = wn2[i];
        }
        tl = ntl;
        blk = nblkidx;
        k = nk;
        st ^= 1;
    }
    __syncthreads();
    if (nmine > 0 && threadIdx.x < TB) {   // the last tile's column sums (st was flipped after it)
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) sum += csum[st ^ 1][w][threadIdx.x];
        cacc[(ntile - 1) * TB + threadIdx.x] += sum;
    }
    __syncthreads();
    for (long i = threadIdx.x; i < g.ldua; i += 512) g.cpart[(long)blockIdx.x * g.ldua + i] = cacc[i];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        double sum = accS[jt];
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (l4 == 0) red[wv][16 * jt + l15] = sum;
    }
    {
        double sum = accK;
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        if (lane == 0) red[wv][C::N2] = sum;
    }
    __syncthreads();
    if (threadIdx.x <= C::N2) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < NWV; ++w) sum += red[w][threadIdx.x];
        g.spart[(long)blockIdx.x * (C::N2 + 1) + threadIdx.x] = sum;
    }
}


// ---- the exact GP's gradient trace on the matrix cores: k(X, X) against a SYMMETRIC weight matrix ----------------------
// w_ab = M1_ab + u_a v_b with M1 = (K + sn2 I)^-1 valid in its lower triangle only (GP.cpp:154-161; exact.hip), both point
// sets are the training set.  Tiles on and below the diagonal are visited once: an element below the diagonal counts twice
// (w, K and (x_ia - x_ib)^2 are symmetric), the diagonal once with K_aa = sf2 EXACTLY (r2 forced to 0, as the direct
// differences of util.cpp:64-75 give it) and elements above it not at all.  Same two contractions as trace_mfma_kernel;
// the column term  sum_b x~_ib^2 (Q^T 1)_b  is contracted per tile and wave on the vector unit (d multiply-adds per lane and
// 64-column tile beside 32 exp: 3 %) instead of being written out as n column sums per workgroup.
// Work unit = (row block of 64 NS rows, chunk of SYM_CT column tiles): the triangle's row blocks differ in length by a
// factor of n / 128, fixed-size chunks keep the workgroups within a few per cent of one another.
//   out slots per workgroup (stride N2 + 2): [c < d] S_c, [N2] sum Q, [N2 + 1] tr(w)
// gr = row blocks per chunk width (a chunk = gr * RB columns): 16 (2048 columns) for large n, less where that would leave
// fewer units than resident workgroups (n = 8192: 160 units of 32 tiles took 327 us, 2080 units of 2 tiles take a third)
inline long sym_units(long nblk, long gr) {
    const long a = nblk / gr, b = nblk % gr;
    return gr * a * (a + 1) / 2 + b * (a + 1);
}
inline int sym_gr(long nblk) {
    int gr = 16;
    while (gr > 1 && sym_units(nblk, gr) < 2048) gr /= 2;
    return gr;
}

template <int KK, int NS>
__global__ __launch_bounds__(256, 2) void trace_sym_mfma_kernel(const MfmaPairArgs g, int nunit, int GR) {
    using C = Cfg<KK>;
    constexpr int JT = C::JT, TB = C::TB, RB = 64 * NS;
    const int SYM_CT = GR * RB / TB;       // column tiles per work unit
    constexpr int DC = 4 * KK - 2;         // dimensions this instantiation can hold (d <= DC)
    __shared__ __attribute__((aligned(16))) double sU[C::DK * C::S1];
    __shared__ __attribute__((aligned(16))) double sUT[TB * C::S2];
    __shared__ double sv[TB];
    __shared__ double csum[4][TB];
    __shared__ double red[4][C::N2 + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const int d = g.d;
    double accS[JT], accC[DC];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) accS[jt] = 0.0;
#pragma unroll
    for (int c = 0; c < DC; ++c) accC[c] = 0.0;
    double accK = 0.0, accT = 0.0;
    const bool vec = NS == 1 || (((reinterpret_cast<uintptr_t>(g.M1) & 15) == 0) && (g.ldm1 % 2 == 0));

    for (int unit = blockIdx.x; unit < nunit; unit += gridDim.x) {
        // unit -> (row block blk, chunk ch): row blocks GR a .. GR a + GR - 1 have a + 1 chunks each
        int a_ = 0;
        while (GR * (a_ + 1) * (a_ + 2) / 2 <= unit) ++a_;
        const int o_ = unit - GR * a_ * (a_ + 1) / 2;
        const int blk = GR * a_ + o_ / (a_ + 1), ch = o_ % (a_ + 1);
        const long row0 = (long)blk * RB;                       // first row of the block
        const long a0 = row0 + 16 * NS * wv + NS * l15;         // this lane's NS adjacent rows
        const long bend = min((long)g.n, row0 + RB);            // columns < bend can be on or below the diagonal
        const long bfirst = (long)ch * SYM_CT * TB, blast = min(bend, bfirst + (long)SYM_CT * TB);
        double xf[NS][KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const double* px = g.Xa + (long)(4 * kk + l4) * g.ldxa + a0;
            if (NS == 2) {
                const v2d x = *reinterpret_cast<const v2d*>(px);
                xf[0][kk] = x.x;
                xf[NS - 1][kk] = x.y;
            } else {
                xf[0][kk] = px[0];
            }
        }
        double ua[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) ua[s] = (a0 + s < g.n && g.u) ? g.u[a0 + s] : 0.0;
        v4d acc2[NS][JT];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) acc2[s][jt] = (v4d){0, 0, 0, 0};
        const bool rows_in = row0 + RB <= g.n;

        // the weights of a 16-column group are requested one group ahead (the first group of the next tile across the
        // tile's barriers): a wave has 4 KB of weights in flight beside the group it works on
        auto load_w = [&](long tb0, int bt, double (&w)[4][NS]) {
            const bool tfull = tb0 + TB <= row0 && rows_in && vec;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long b = tb0 + 16 * bt + 4 * r + l4;
                const double* p1 = g.M1 + a0 + b * g.ldm1;
                if (tfull) {
                    if (NS == 2) {
                        const v2d t = *reinterpret_cast<const v2d*>(p1);
                        w[r][0] = t.x;
                        w[r][NS - 1] = t.y;
                    } else {
                        w[r][0] = p1[0];
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < NS; ++s) w[r][s] = (b <= a0 + s && a0 + s < g.n) ? p1[s] : 0.0;   // lower triangle only
                }
            }
        };
        double wn[4][NS];
        if (bfirst < blast) load_w(bfirst, 0, wn);
        for (long b0 = bfirst; b0 < blast; b0 += TB) {
            const bool below = b0 + TB <= row0;                  // the whole tile lies strictly below the diagonal
            __syncthreads();   // the previous tile's fragments are consumed
            stage_u<KK, true>(g, b0, sU, sUT);
            if (threadIdx.x < TB) sv[threadIdx.x] = (g.v && b0 + threadIdx.x < g.n) ? g.v[b0 + threadIdx.x] : 0.0;
            __syncthreads();
#pragma unroll
            for (int bt = 0; bt < TB / 16; ++bt) {
                double w1[4][NS];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int s = 0; s < NS; ++s) w1[r][s] = wn[r][s];
                if (bt + 1 < TB / 16) load_w(b0, bt + 1, wn);
                else if (b0 + TB < blast) load_w(b0 + TB, 0, wn);
                v4d D[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) D[s] = (v4d){0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const double uf = sU[(4 * kk + l4) * C::S1 + 16 * bt + l15];
#pragma unroll
                    for (int s = 0; s < NS; ++s) D[s] = mfma(uf, xf[s][kk], D[s]);
                }
                double q[NS][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int bl = 16 * bt + 4 * r + l4;
                    const long b = b0 + bl;
                    const double vb = sv[bl];
                    double cs = 0.0;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const long a = a0 + s;
                        double w = fma(ua[s], vb, w1[r][s]);
                        double r2 = fmax(D[s][r], 0.0);
                        if (below) {
                            w *= 2.0;
                        } else {
                            const bool diag = a == b;
                            if (diag) {
                                accT += w;
                                r2 = 0.0;
                            }
                            w = (b < a) ? 2.0 * w : (diag ? w : 0.0);
                            if (a >= g.n) w = 0.0;
                        }
                        q[s][r] = w * (g.sf2 * exp_nonpos(-0.5 * r2));
                        cs += q[s][r];
                    }
                    cs = row16_sum(cs);
                    if (l15 == 0) csum[wv][bl] = cs;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt) {
                        const double ub = sUT[(16 * bt + 4 * r + l4) * C::S2 + 16 * jt + l15];
#pragma unroll
                        for (int s = 0; s < NS; ++s) acc2[s][jt] = mfma(q[s][r], ub, acc2[s][jt]);
                    }
            }
            // column term of this wave's rows: lane = column b0 + lane (TB = 64) or b0 + (lane & 31) with the dimensions split
            // over the two half-waves (TB = 32);  sU[c][b] = -2 x~_cb
            {
                const int bl = lane & (TB - 1);
                const double cs = 0.25 * csum[wv][bl];
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    if (TB == 32 && ((c & 1) != (lane >> 5))) continue;
                    const double t = sU[c * C::S1 + bl];
                    accC[c] = fma(t * t, cs, accC[c]);
                }
            }
        }
        // end of the unit: acc2[s][jt][r] = row a = base + NS (l4 + 4 r) + s, column c = 16 jt + l15
        const int jtd = d >> 4, ld = d & 15;   // column d holds (Q 1)_a
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double q1a = 0.0;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt)
                    if (jt == jtd) q1a = acc2[s][jt][r];
                q1a = __shfl(q1a, (lane & 48) | ld, 64);
                const long a = row0 + 16 * NS * wv + NS * (l4 + 4 * r) + s;
                if (l15 == 0) accK += q1a;
#pragma unroll
                for (int jt = 0; jt < JT; ++jt) {
                    const int c = 16 * jt + l15;
                    if (c < d) {
                        const double x = g.Xa[(long)c * g.ldxa + a];   // zero beyond n
                        accS[jt] = fma(x, fma(x, q1a, acc2[s][jt][r]), accS[jt]);
                    }
                }
            }
    }
    // workgroup partials
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) {
        double s = accS[jt];
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (l4 == 0) red[wv][16 * jt + l15] = s;
    }
    {
        double s = accK, t = accT;
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_xor(s, o, 64);   // accK lives in the lanes with l15 == 0, accT anywhere
            t += __shfl_xor(t, o, 64);
        }
        if (lane == 0) {
            red[wv][C::N2] = s;
            red[wv][C::N2 + 1] = t;
        }
    }
    __syncthreads();
    // the column terms: sum over the lanes, added to the row terms of the same dimension
#pragma unroll
    for (int c = 0; c < DC; ++c) {
        double s = accC[c];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0 && c < d) red[wv][c] += s;
    }
    __syncthreads();
    if (threadIdx.x < C::N2 + 2)
        g.spart[(long)blockIdx.x * (C::N2 + 2) + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[i < d] = sum_wg spart[wg][i];  out[d] = sum Q;  out[d + 1] = tr(w)
__global__ void trace_sym_finish_kernel(const double* __restrict__ spart, int nwg, int n2, int d, double* __restrict__ out) {
    __shared__ double red[256];
    const int i = blockIdx.x;
    const int col = i < d ? i : (i == d ? n2 : n2 + 1);
    double s = 0.0;
    for (int w = threadIdx.x; w < nwg; w += 256) s += spart[(long)w * (n2 + 2) + col];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = red[0];
}


// K + diag_add I on and below the diagonal for the exact GP (GP.cpp:128-129: k(X, X) + sn2 I), on the matrix cores:
// r2 from the same contraction as cov_mfma_kernel with both point sets = the training set, work units over the triangle as
// in trace_sym_mfma_kernel; the diagonal is sf2 + diag_add EXACTLY (r2_aa forced to 0, what the direct differences of
// util.cpp:64-75 give), whole diagonal tiles are written (symmetric values above the diagonal inside them: nothing reads them).
template <int KK>
__global__ __launch_bounds__(256, 2) void cov_sym_mfma_kernel(const MfmaPairArgs g, int nunit, int GR, double diag_add) {
    using C = Cfg<KK>;
    constexpr int TB = C::TB, RB = 128;
    const int CT = GR * RB / TB;
    __shared__ __attribute__((aligned(16))) double sU[C::DK * C::S1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
    const bool vec = ((reinterpret_cast<uintptr_t>(g.out) & 15) == 0) && (g.ldo % 2 == 0);
    for (int unit = blockIdx.x; unit < nunit; unit += gridDim.x) {
        int a_ = 0;
        while (GR * (a_ + 1) * (a_ + 2) / 2 <= unit) ++a_;
        const int o_ = unit - GR * a_ * (a_ + 1) / 2;
        const int blk = GR * a_ + o_ / (a_ + 1), ch = o_ % (a_ + 1);
        const long row0 = (long)blk * RB;
        const long a0 = row0 + 32 * wv + 2 * l15;   // this lane's two rows
        const long bend = min((long)g.n, row0 + RB);
        const long bfirst = (long)ch * CT * TB, blast = min(bend, bfirst + (long)CT * TB);
        double xf[2][KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const v2d x = *reinterpret_cast<const v2d*>(g.Xa + (long)(4 * kk + l4) * g.ldxa + a0);
            xf[0][kk] = x.x;
            xf[1][kk] = x.y;
        }
        const bool rows_in = row0 + RB <= g.n;
        for (long b0 = bfirst; b0 < blast; b0 += TB) {
            __syncthreads();
            stage_u<KK, false>(g, b0, sU, nullptr);
            __syncthreads();
            const bool below = b0 + TB <= row0;
            const bool full = rows_in && vec && b0 + TB <= g.n;
#pragma unroll
            for (int bt = 0; bt < TB / 16; ++bt) {
                v4d D0 = {0, 0, 0, 0}, D1 = {0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const double ua = sU[(4 * kk + l4) * C::S1 + 16 * bt + l15];
                    D0 = mfma(ua, xf[0][kk], D0);
                    D1 = mfma(ua, xf[1][kk], D1);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long b = b0 + 16 * bt + 4 * r + l4;
                    double k0 = g.sf2 * exp_nonpos(-0.5 * fmax(D0[r], 0.0));
                    double k1 = g.sf2 * exp_nonpos(-0.5 * fmax(D1[r], 0.0));
                    if (!below) {
                        if (b == a0) k0 = g.sf2 + diag_add;
                        if (b == a0 + 1) k1 = g.sf2 + diag_add;
                    }
                    double* o = g.out + a0 + b * g.ldo;
                    if (full) {
                        *reinterpret_cast<v2d*>(o) = (v2d){k0, k1};
                    } else if (b < g.n) {
                        if (a0 < g.n) o[0] = k0;
                        if (a0 + 1 < g.n) o[1] = k1;
                    }
                }
            }
        }
    }
}

// ---- operand preparation ----------------------------------------------------------------------------
// cen[i] = mean over the m inducing points of the scaled coordinate i
__global__ void centre_kernel(const double* __restrict__ Ut, long ldu, long m, double* __restrict__ cen) {
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0.0;
    for (long b = threadIdx.x; b < m; b += 256) s += Ut[(long)i * ldu + b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) cen[i] = red[0] / (double)m;
}

// side 0: Xa = [x~ ; |x~|^2 ; 1 ; 0..]      side 1: Ua = [-2 u~ ; 1 ; |u~|^2 ; 0..] and UaT = rows 0..d transposed
__global__ void augment_kernel(const double* __restrict__ Xt, long ldx, long cnt, int d, int dk, const double* __restrict__ cen,
                               int side, double* __restrict__ A, long lda, double* __restrict__ AT, int n2) {
    const long a = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= lda) return;
    const bool in = a < cnt;
    double nrm = 0.0;
    for (int i = 0; i < d; ++i) {
        const double v = in ? Xt[(long)i * ldx + a] - cen[i] : 0.0;
        nrm = fma(v, v, nrm);
        const double o = side ? -2.0 * v : v;
        A[(long)i * lda + a] = o;
        if (AT) AT[a * n2 + i] = o;
    }
    const double one = in ? 1.0 : 0.0;
    A[(long)d * lda + a] = side ? one : nrm;
    A[(long)(d + 1) * lda + a] = side ? nrm : one;
    for (int i = d + 2; i < dk; ++i) A[(long)i * lda + a] = 0.0;
    if (AT) {
        AT[a * n2 + d] = one;
        for (int i = d + 1; i < n2; ++i) AT[a * n2 + i] = 0.0;
    }
}

__global__ __launch_bounds__(256) void colsum_finish_kernel(const double* __restrict__ cpart, int nwg, long ldua, double* __restrict__ cvec) {
    // 32 columns x 8 row groups per block (one thread per column walking all nwg rows is a chain of nwg dependent loads:
    // 200 us at nwg = 512); fixed order of the partial sums: deterministic
    __shared__ double red[8][33];
    const int c = threadIdx.x & 31, q = threadIdx.x >> 5;
    const long b = (long)blockIdx.x * 32 + c;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (b < ldua) {
        int w = q;
        for (; w + 24 < nwg; w += 32) {
            s0 += cpart[(long)w * ldua + b];
            s1 += cpart[(long)(w + 8) * ldua + b];
            s2 += cpart[(long)(w + 16) * ldua + b];
            s3 += cpart[(long)(w + 24) * ldua + b];
        }
        for (; w < nwg; w += 8) s0 += cpart[(long)w * ldua + b];
    }
    red[q][c] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && b < ldua) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][c];
        cvec[b] = s;
    }
}

// out[i] = sum_wg spart[wg][i] + sum_b u~_ib^2 colsum_b  (i < d);  out[d] = sum_ab Q_ab;  out[d+1] = 0
__global__ void trace_mfma_finish_kernel(const double* __restrict__ spart, int nwg, int n2, const double* __restrict__ Ua,
                                         long ldua, long m, const double* __restrict__ cvec, int d, double* __restrict__ out) {
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0.0;
    const int col = i < d ? i : n2;
    for (int w = threadIdx.x; w < nwg; w += 256) s += spart[(long)w * (n2 + 1) + col];
    if (i < d)
        for (long b = threadIdx.x; b < m; b += 256) {
            const double t = Ua[(long)i * ldua + b];   // -2 u~
            s = fma(0.25 * t * t, cvec[b], s);
        }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[i] = red[0];
        if (i == d) out[d + 1] = 0.0;
    }
}

#define GPX_KK_CASE(KERNEL, K, grid, st, args) \
    case K: hipLaunchKernelGGL((KERNEL(K)), dim3(grid), dim3(256), 0, st, args); break;
#define GPX_KK_SWITCH(KERNEL, kk, grid, st, args)                                                         \
    switch (kk) {                                                                                          \
        GPX_KK_CASE(KERNEL, 1, grid, st, args) GPX_KK_CASE(KERNEL, 2, grid, st, args)                      \
        GPX_KK_CASE(KERNEL, 3, grid, st, args) GPX_KK_CASE(KERNEL, 4, grid, st, args)                      \
        GPX_KK_CASE(KERNEL, 5, grid, st, args) GPX_KK_CASE(KERNEL, 6, grid, st, args)                      \
        GPX_KK_CASE(KERNEL, 7, grid, st, args) GPX_KK_CASE(KERNEL, 8, grid, st, args)                      \
        GPX_KK_CASE(KERNEL, 9, grid, st, args) GPX_KK_CASE(KERNEL, 10, grid, st, args)                     \
        GPX_KK_CASE(KERNEL, 11, grid, st, args) GPX_KK_CASE(KERNEL, 12, grid, st, args)                    \
        GPX_KK_CASE(KERNEL, 13, grid, st, args) GPX_KK_CASE(KERNEL, 14, grid, st, args)                    \
        GPX_KK_CASE(KERNEL, 15, grid, st, args)                                                            \
        default: hipLaunchKernelGGL((KERNEL(16)), dim3(grid), dim3(256), 0, st, args); break;              \
    }
#define GPX_COV_K(K) cov_mfma_kernel<K>
#define GPX_TRACE_NS(K) ((K) <= 8 ? 2 : 1)
#define GPX_TRACE_K1(K) trace_mfma_kernel<K, false, GPX_TRACE_NS(K)>
#define GPX_TRACE_K2(K) trace_mfma_kernel<K, true, GPX_TRACE_NS(K)>

#define GPX_KK8_CASE(M2, K, grid, st, args) \
    case K: hipLaunchKernelGGL((trace_mfma8_kernel<K, M2>), dim3(grid), dim3(512), 0, st, args); break;
#define GPX_KK8_SWITCH_SHALLOW(M2, kk, grid, st, args)                                                           \
    switch (kk) {                                                                                                \
        GPX_KK8_CASE(M2, 1, grid, st, args) GPX_KK8_CASE(M2, 2, grid, st, args) GPX_KK8_CASE(M2, 3, grid, st, args) \
        GPX_KK8_CASE(M2, 4, grid, st, args) GPX_KK8_CASE(M2, 5, grid, st, args) GPX_KK8_CASE(M2, 6, grid, st, args) \
        GPX_KK8_CASE(M2, 7, grid, st, args)                                                                        \
        default: hipLaunchKernelGGL((trace_mfma8_kernel<8, M2>), dim3(grid), dim3(512), 0, st, args); break;      \
    }
#define GPX_KK8_SWITCH_DEEP(kk, grid, st, args)   /* one weight matrix only: with two the registers spill */       \
    switch (kk) {                                                                                                \
        GPX_KK8_CASE(false, 9, grid, st, args) GPX_KK8_CASE(false, 10, grid, st, args)                             \
        GPX_KK8_CASE(false, 11, grid, st, args) GPX_KK8_CASE(false, 12, grid, st, args)                            \
        GPX_KK8_CASE(false, 13, grid, st, args) GPX_KK8_CASE(false, 14, grid, st, args)                            \
        GPX_KK8_CASE(false, 15, grid, st, args)                                                                    \
        default: hipLaunchKernelGGL((trace_mfma8_kernel<16, false>), dim3(grid), dim3(512), 0, st, args); break;  \
    }

inline int kk_of(int d) { return (d + 2 + 3) / 4; }
inline int n2_of(int d) { return 16 * ((4 * kk_of(d) - 1 + 15) / 16); }

}  // namespace

bool gpx_mfma_pair_supported(int d) { return d >= 1 && kk_of(d) <= 16; }
long gpx_mfma_ldxa(long n) { return (n + 127) / 128 * 128; }
long gpx_mfma_ldua(long m) { return (m + 63) / 64 * 64; }
int gpx_mfma_dk(int d) { return 4 * kk_of(d); }
int gpx_mfma_n2(int d) { return n2_of(d); }
static int g_wide_mode = 0;   // GPX_WIDE_TRACE=none|all: A/B measurements of the two trace kernels
// the 8-wave kernel: deep contractions whose (padded) inducing set fits its LDS column-sum accumulator
// (one weight matrix only: with two, its registers spill and the 4-wave kernel stays)
static inline bool trace_wide(int d, long m, bool two) {
    if (gpx_mfma_ldua(m) > CACC_MAX || g_wide_mode == 1) return false;
    if (g_wide_mode == 2) return kk_of(d) <= 8 || !two;
    return kk_of(d) > 8 && !two;
}
static inline int trace_rows(int d, long m, bool two) { return (kk_of(d) <= 8 || trace_wide(d, m, two)) ? 128 : 64; }   // rows per workgroup
int gpx_mfma_trace_grid(long n, int d, long m, bool two) {
    static const bool init = [] {
        const char* e = getenv("GPX_WIDE_TRACE");
        g_wide_mode = !e ? 0 : (e[0] == 'n' ? 1 : (e[0] == 'a' ? 2 : 0));
        return true;
    }();
    (void)init;
    const long nblk = (n + trace_rows(d, m, two) - 1) / trace_rows(d, m, two);
    const long cap = trace_wide(d, m, two) ? 256 : 512;   // workgroups resident at a time: 1 or 2 per CU x 256 CUs
    return (int)(nblk < cap ? nblk : cap);           // a fixed function of (n, d, m): deterministic sums
}

int gpx_mfma_prepare(hipStream_t st, const double* Xt, long ldx, long n, const double* Ut, long ldu, long m, int d,
                     double* cen, double* Xa, double* Ua, double* UaT) {
    if (!gpx_mfma_pair_supported(d)) return GPX_ERR_UNSUPPORTED;
    const int dk = gpx_mfma_dk(d), n2 = gpx_mfma_n2(d);
    const long ldxa = gpx_mfma_ldxa(n), ldua = gpx_mfma_ldua(m);
    hipLaunchKernelGGL(centre_kernel, dim3(d), dim3(256), 0, st, Ut, ldu, m, cen);
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((ldua + 255) / 256)), dim3(256), 0, st, Ut, ldu, m, d, dk, cen, 1, Ua, ldua,
                       UaT, n2);
    if (Xa)
        hipLaunchKernelGGL(augment_kernel, dim3((unsigned)((ldxa + 255) / 256)), dim3(256), 0, st, Xt, ldx, n, d, dk, cen, 0, Xa,
                           ldxa, (double*)nullptr, n2);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

int gpx_mfma_cov_build(hipStream_t st, const double* Xa, long n, const double* Ua, long m, int d, double sf2, double* out,
                       long ldo, long pan, long pst, int max_grid) {
    if (n <= 0 || m <= 0) return GPX_OK;
    if (pan % 128) return GPX_ERR_ARG;
    MfmaPairArgs g = {};
    g.pan = pan; g.pst = pst;
    g.Xa = Xa; g.ldxa = gpx_mfma_ldxa(n); g.n = n;
    g.Ua = Ua; g.ldua = gpx_mfma_ldua(m); g.m = m;
    g.d = d; g.sf2 = sf2; g.out = out; g.ldo = ldo;
    const long nblk = (n + 127) / 128;
    if (nblk > 0x7fffffffL) return GPX_ERR_ARG;
    g.nblk = (int)nblk;
    const long grid = max_grid > 0 && max_grid < nblk ? max_grid : nblk;
    const long slot = gpx_pair_prof_open(st, 0, 8.0 * (double)n * (double)m);
    GPX_KK_SWITCH(GPX_COV_K, kk_of(d), (unsigned)grid, st, g);
    gpx_pair_prof_close(st, slot);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

// out[0..d-1] = S_i, out[d] = sum W K, out[d+1] = 0.  cpart: grid * ldua doubles, spart: grid * (N2 + 1), cvec: ldua.
int gpx_mfma_pair_trace(hipStream_t st, const double* Xa, long n, const double* Ua, const double* UaT, long m, int d, double sf2,
                        const TraceArgs& w, double* cpart, double* spart, double* cvec, double* out, long pan, long pst) {
    if (n <= 0 || m <= 0 || pan % 128) return GPX_ERR_ARG;
    MfmaPairArgs g = {};
    g.pan = pan; g.pst = pst;
    g.Xa = Xa; g.ldxa = gpx_mfma_ldxa(n); g.n = n;
    g.Ua = Ua; g.ldua = gpx_mfma_ldua(m); g.m = m;
    g.UaT = UaT;
    g.d = d; g.sf2 = sf2;
    g.M1 = w.M1; g.ldm1 = w.ldm1; g.c1 = w.c1;
    g.M2 = w.M2; g.ldm2 = w.ldm2; g.c2 = w.c2;
    g.u = w.u; g.v = w.v;
    g.cpart = cpart; g.spart = spart;
    const bool two = w.M2 != nullptr;
    const int grid = gpx_mfma_trace_grid(n, d, m, two), n2 = gpx_mfma_n2(d);
    const long nblk = (n + trace_rows(d, m, two) - 1) / trace_rows(d, m, two);
    if (nblk > 0x7fffffffL) return GPX_ERR_ARG;
    g.nblk = (int)nblk;
    HIP_TRY(hipMemsetAsync(cpart, 0, (size_t)grid * g.ldua * sizeof(double), st));
    const long slot = gpx_pair_prof_open(st, 1, 8.0 * (double)n * (double)m * (w.M2 ? 2.0 : 1.0));
    if (trace_wide(d, m, two) && kk_of(d) > 8) {
        GPX_KK8_SWITCH_DEEP(kk_of(d), (unsigned)grid, st, g);
    } else if (trace_wide(d, m, two) && two) {
        GPX_KK8_SWITCH_SHALLOW(true, kk_of(d), (unsigned)grid, st, g);
    } else if (trace_wide(d, m, two)) {
        GPX_KK8_SWITCH_SHALLOW(false, kk_of(d), (unsigned)grid, st, g);
    } else if (w.M2) {
        GPX_KK_SWITCH(GPX_TRACE_K2, kk_of(d), (unsigned)grid, st, g);
    } else {
        GPX_KK_SWITCH(GPX_TRACE_K1, kk_of(d), (unsigned)grid, st, g);
    }
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((g.ldua + 31) / 32)), dim3(256), 0, st, cpart, grid, g.ldua, cvec);
    hipLaunchKernelGGL(trace_mfma_finish_kernel, dim3(d + 1), dim3(256), 0, st, spart, grid, n2, Ua, g.ldua, m, cvec, d, out);
    gpx_pair_prof_close(st, slot);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

// The exact GP's trace (symmetric weights, both point sets = the training set): d <= 30 (two 16-row tiles per wave).
bool gpx_mfma_sym_trace_supported(int d) { return d >= 1 && kk_of(d) <= 8; }
int gpx_mfma_sym_trace_grid(long n, int d) {
    (void)d;
    const long nblk = (n + 127) / 128;
    const long units = sym_units(nblk, sym_gr(nblk));
    return (int)(units < 1024 ? units : 1024);   // 2 resident workgroups per CU x 2 rounds of units in flight
}
// Xa, Ua, UaT from gpx_mfma_prepare(st, Xt, ldx, n, Xt, ldx, n, ...); M1 valid in its lower triangle; w = M1 + u v^T;
// spart: grid * (N2 + 2) doubles.  out[0..d-1] = S_i, out[d] = sum w K, out[d+1] = tr(w)  (as gpx_pair_trace with sym = 1)
int gpx_mfma_sym_trace(hipStream_t st, const double* Xa, const double* Ua, const double* UaT, long n, int d, double sf2,
                       const double* M1, long ldm1, const double* u, const double* v, double* spart, double* out) {
    if (n <= 0 || !gpx_mfma_sym_trace_supported(d)) return GPX_ERR_ARG;
    MfmaPairArgs g = {};
    g.Xa = Xa; g.ldxa = gpx_mfma_ldxa(n); g.n = n;
    g.Ua = Ua; g.ldua = gpx_mfma_ldua(n); g.m = n;
    g.UaT = UaT;
    g.d = d; g.sf2 = sf2;
    g.M1 = M1; g.ldm1 = ldm1;
    g.u = u; g.v = v;
    g.spart = spart;
    const long nblk = (n + 127) / 128;
    const int gr = sym_gr(nblk);
    const long units = sym_units(nblk, gr);
    if (units > 0x3fffffffL) return GPX_ERR_ARG;
    const int grid = gpx_mfma_sym_trace_grid(n, d), n2 = gpx_mfma_n2(d);
    const long slot = gpx_pair_prof_open(st, 1, 4.0 * (double)n * (double)n);
    switch (kk_of(d)) {
#define GPX_SYM_CASE(K) case K: hipLaunchKernelGGL((trace_sym_mfma_kernel<K, 2>), dim3(grid), dim3(256), 0, st, g, (int)units, gr); break;
        GPX_SYM_CASE(1) GPX_SYM_CASE(2) GPX_SYM_CASE(3) GPX_SYM_CASE(4) GPX_SYM_CASE(5) GPX_SYM_CASE(6) GPX_SYM_CASE(7)
        default: hipLaunchKernelGGL((trace_sym_mfma_kernel<8, 2>), dim3(grid), dim3(256), 0, st, g, (int)units, gr); break;
#undef GPX_SYM_CASE
    }
    hipLaunchKernelGGL(trace_sym_finish_kernel, dim3(d + 2), dim3(256), 0, st, spart, grid, n2, d, out);
    gpx_pair_prof_close(st, slot);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

// K + diag_add I, lower triangle (whole tiles on the diagonal), of the training set against itself: d <= 62
int gpx_mfma_cov_sym_build(hipStream_t st, const double* Xa, const double* Ua, long n, int d, double sf2, double diag_add,
                           double* out, long ldo) {
    if (n <= 0) return GPX_OK;
    if (!gpx_mfma_pair_supported(d)) return GPX_ERR_UNSUPPORTED;
    MfmaPairArgs g = {};
    g.Xa = Xa; g.ldxa = gpx_mfma_ldxa(n); g.n = n;
    g.Ua = Ua; g.ldua = gpx_mfma_ldua(n); g.m = n;
    g.d = d; g.sf2 = sf2; g.out = out; g.ldo = ldo;
    const long nblk = (n + 127) / 128;
    const int tb = kk_of(d) <= 8 ? 64 : 32;
    int gr = sym_gr(nblk);
    if (tb == 32 && gr < 1) gr = 1;
    const long units = sym_units(nblk, gr);
    if (units > 0x3fffffffL) return GPX_ERR_ARG;
    const int grid = (int)(units < 2048 ? units : 2048);
    const long slot = gpx_pair_prof_open(st, 0, 4.0 * (double)n * (double)n);
#define GPX_COVSYM_CASE(K) case K: hipLaunchKernelGGL((cov_sym_mfma_kernel<K>), dim3(grid), dim3(256), 0, st, g, (int)units, gr, diag_add); break;
    switch (kk_of(d)) {
        GPX_COVSYM_CASE(1) GPX_COVSYM_CASE(2) GPX_COVSYM_CASE(3) GPX_COVSYM_CASE(4) GPX_COVSYM_CASE(5) GPX_COVSYM_CASE(6)
        GPX_COVSYM_CASE(7) GPX_COVSYM_CASE(8) GPX_COVSYM_CASE(9) GPX_COVSYM_CASE(10) GPX_COVSYM_CASE(11) GPX_COVSYM_CASE(12)
        GPX_COVSYM_CASE(13) GPX_COVSYM_CASE(14) GPX_COVSYM_CASE(15)
        default: hipLaunchKernelGGL((cov_sym_mfma_kernel<16>), dim3(grid), dim3(256), 0, st, g, (int)units, gr, diag_add); break;
    }
#undef GPX_COVSYM_CASE
    gpx_pair_prof_close(st, slot);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}
