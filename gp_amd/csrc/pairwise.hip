// pairwise.hip — squared-exponential kernel matrices and fused weighted-derivative sums (gfx950).
//
//  * scale_inputs : Xt[i][a] = x_ia / l_i, stored dimension-major so that a 64-point tile of one
//                   dimension is one 512-B coalesced read (CovSEard.cpp:9, CovSEiso.cpp:8-10).
//  * cov_build    : K_ab = sf2 * exp(-r2_ab / 2) from direct differences of the scaled inputs
//                   (util.cpp:64-75 is direct-difference too, so the diagonal of k(X,X) is exactly
//                   sf2), optional lower-triangle-only output and a diagonal shift (sn2 / jitter).
//  * pair_trace   : for a weight matrix w_ab given as c1_a*M1_ab + c2_a*M2_ab + u_a*v_b, the sums
//                   S_i = sum_ab w_ab K_ab ((x_ia - x_ib)/l_i)^2 for every dimension i, S_K = sum w K
//                   and tr(w): every dK/dtheta of CovSEard.cpp:12-22 contracted with the weights
//                   in ONE pass, K recomputed on the fly, dK never materialised (replaces the
//                   per-hyper loops GP.cpp:155-161, FITC.cpp:269-308, VFE.cpp:213-237).
// All three are HBM/VALU-bound O(p q d) kernels: 64x64 tiles, 256 threads, lanes along the
// first index so every global access is a contiguous 512-B row per wave.
#include "gpx_internal.h"
#include "kernels.h"
#include <atomic>
#include <mutex>
#include <type_traits>

namespace {

// ---- optional HIP-event timing of the large pairwise launches (bench.py's HBM roofline objects) ----
struct PairProf {
    std::mutex mu;
    std::atomic<bool> on{false};   // read outside the lock by every launch
    int dev = -1;                  // only launches of the device current at gpx_pair_prof_begin are timed
    double min_bytes = 0;
    std::vector<hipEvent_t> ev;  // pairs
    std::vector<double> bytes;
    std::vector<int> cls;  // 0 = cov_build, 1 = pair_trace
    size_t used = 0;
} p_prof;

long pp_open(hipStream_t st, int cls, double bytes) {
    if (!p_prof.on.load(std::memory_order_acquire)) return -1;
    std::lock_guard<std::mutex> lk(p_prof.mu);
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (!p_prof.on || cur != p_prof.dev || bytes < p_prof.min_bytes || p_prof.used + 2 > p_prof.ev.size()) return -1;
    const long slot = (long)p_prof.used;
    p_prof.used += 2;
    p_prof.bytes.push_back(bytes);
    p_prof.cls.push_back(cls);
    (void)hipEventRecord(p_prof.ev[slot], st);
    return slot;
}
void pp_close(hipStream_t st, long slot) {
    if (slot >= 0) (void)hipEventRecord(p_prof.ev[slot + 1], st);
}

constexpr int DCH = 32;  // dimensions staged in LDS at a time

__global__ void scale_inputs_kernel(const double* __restrict__ X, long n, int d, int dpad,
                                    ScaleArgs sc, double* __restrict__ Xt, long ldx) {
    const long a = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    for (int i = 0; i < dpad; ++i) {
        double v = 0.0;
        if (i < d) {
            const double x = X[i + a * (long)d];
            v = sc.divide ? x / sc.s[i] : x * sc.s[i];
        }
        Xt[(long)i * ldx + a] = v;
    }
}

__device__ __forceinline__ void decode_tile(int tile, bool lower, int tiles_p, int& ta, int& tb) {
    if (lower) {
        ta = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
        while (ta * (ta + 1) / 2 > tile) --ta;
        while ((ta + 1) * (ta + 2) / 2 <= tile) ++ta;
        tb = tile - ta * (ta + 1) / 2;
    } else {
        ta = tile % tiles_p;
        tb = tile / tiles_p;
    }
}

// stage dims [c0, c0+cw) of a 64-point tile of each input set into LDS
__device__ __forceinline__ void stage(double (*sa)[64], double (*sb)[64], const double* __restrict__ Xt1,
                                      long ld1, long p, long a0, const double* __restrict__ Xt2, long ld2,
                                      long q, long b0, int c0, int cw, int dpad) {
    for (int idx = threadIdx.x; idx < cw * 64; idx += 256) {
        const int i = idx >> 6, l = idx & 63;
        const bool din = c0 + i < dpad;
        sa[i][l] = (din && a0 + l < p) ? Xt1[(long)(c0 + i) * ld1 + a0 + l] : 0.0;
        sb[i][l] = (din && b0 + l < q) ? Xt2[(long)(c0 + i) * ld2 + b0 + l] : 0.0;
    }
}

// Register tiling shared by cov_build and pair_trace: thread (rg, cg) = (tid & 15, tid >> 4) owns the
// 4 x 4 patch rows a0 + 4 rg .., columns b0 + 4 cg .. of a 64 x 64 tile.  Per dimension it reads its
// 4 + 4 coordinates from LDS as four 16-B loads for 16 pairs (one LDS instruction per 4 pair-terms;
// one per term with a row-per-lane mapping, which made these kernels LDS-issue-bound) and a wave's
// 16 row groups touch 512 contiguous bytes of every output / weight column.
__device__ __forceinline__ void patch_r2(const double (*sa)[64], const double (*sb)[64], int cw, int rg, int cg,
                                         double (&r2)[4][4]) {
    for (int i = 0; i < cw; ++i) {
        const v2d a01 = *reinterpret_cast<const v2d*>(&sa[i][4 * rg]), a23 = *reinterpret_cast<const v2d*>(&sa[i][4 * rg + 2]);
        const v2d b01 = *reinterpret_cast<const v2d*>(&sb[i][4 * cg]), b23 = *reinterpret_cast<const v2d*>(&sb[i][4 * cg + 2]);
        const double xa[4] = {a01.x, a01.y, a23.x, a23.y}, xb[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double t = xa[r] - xb[c];
                r2[c][r] = fma(t, t, r2[c][r]);
            }
    }
}

__global__ __launch_bounds__(256) void cov_build_kernel(const double* __restrict__ Xt1, long ld1, long p,
                                                        const double* __restrict__ Xt2, long ld2, long q,
                                                        int dpad, double sf2, double* __restrict__ out,
                                                        long ldo, int lower, int tiles_p, double diag_add,
                                                        int vec) {
    __shared__ __attribute__((aligned(16))) double sa[DCH][64], sb[DCH][64];
    int ta, tb;
    decode_tile(blockIdx.x, lower, tiles_p, ta, tb);
    const long a0 = (long)ta * 64, b0 = (long)tb * 64;
    const int rg = threadIdx.x & 15, cg = threadIdx.x >> 4;
    double r2[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) r2[c][r] = 0.0;
    for (int c0 = 0; c0 < dpad; c0 += DCH) {
        const int cw = min(DCH, dpad - c0);
        __syncthreads();
        stage(sa, sb, Xt1, ld1, p, a0, Xt2, ld2, q, b0, c0, cw, dpad);
        __syncthreads();
        patch_r2(sa, sb, cw, rg, cg, r2);
    }
    const long a = a0 + 4 * rg;
    const bool full = vec && a0 + 64 <= p && b0 + 64 <= q;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const long b = b0 + 4 * cg + c;
        double v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = sf2 * exp(-0.5 * r2[c][r]);
            if (a + r == b) v[r] += diag_add;
        }
        if (full) {
            double* o = out + a + b * ldo;
            *reinterpret_cast<v2d*>(o) = (v2d){v[0], v[1]};
            *reinterpret_cast<v2d*>(o + 2) = (v2d){v[2], v[3]};
        } else if (b < q) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (a + r < p) out[a + r + b * ldo] = v[r];
        }
    }
}

// DC dimensions accumulate in registers; DL more (dims dim0+DC ..) accumulate in a per-thread LDS
// column, so that up to 64 dimensions share ONE evaluation of the weights and of K (a second register
// set of 32 accumulators spills)
template <int DC, int DL = 0>
__global__ __launch_bounds__(256) void pair_trace_kernel(const TraceArgs g) {
    __shared__ __attribute__((aligned(16))) double sa[DCH][64], sb[DCH][64];
    __shared__ double red[4][DC + DL + 2];
    __shared__ double lacc[DL > 0 ? DL : 1][64];  // one column per lane quad (the quad is summed first)
    const int rg = threadIdx.x & 15, cg = threadIdx.x >> 4;
    double acc[DC];
#pragma unroll
    for (int i = 0; i < DC; ++i) acc[i] = 0.0;
    if (threadIdx.x < 64) {
#pragma unroll
        for (int i = 0; i < DL; ++i) lacc[i][threadIdx.x] = 0.0;
    }
    double sK = 0.0, tr = 0.0;
    const bool single = g.dpad <= DCH;  // the whole input fits one staging pass

    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        int ta, tb;
        decode_tile(tile, g.sym != 0, g.tiles_p, ta, tb);
        const long a0 = (long)ta * 64, b0 = (long)tb * 64;
        double r2[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) r2[c][r] = 0.0;
        for (int c0 = 0; c0 < g.dpad; c0 += DCH) {
            const int cw = min(DCH, g.dpad - c0);
            __syncthreads();
            stage(sa, sb, g.Xt1, g.ld1, g.p, a0, g.Xt2, g.ld2, g.q, b0, c0, cw, g.dpad);
            __syncthreads();
            patch_r2(sa, sb, cw, rg, cg, r2);
        }
        // weights times K
        const long a = a0 + 4 * rg;
        // tiles on the index diagonal go element by element: tr(w), and in symmetric mode the mirrored reads
        const bool full = g.vec && ta != tb && a0 + 64 <= g.p && b0 + 64 <= g.q;
        const double symw = (g.sym && ta != tb) ? 2.0 : 1.0;
        double c1a[4], c2a[4], ua[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool ain = a + r < g.p;
            c1a[r] = (ain && g.c1) ? g.c1[a + r] : 1.0;
            c2a[r] = (ain && g.c2) ? g.c2[a + r] : 1.0;
            ua[r] = (ain && g.u) ? g.u[a + r] : 0.0;
        }
        double qk[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long b = b0 + 4 * cg + c;
            double w[4] = {0.0, 0.0, 0.0, 0.0};
            if (full) {
                const double* m1 = g.M1 + a + b * g.ldm1;
                const v2d x01 = *reinterpret_cast<const v2d*>(m1), x23 = *reinterpret_cast<const v2d*>(m1 + 2);
                w[0] = c1a[0] * x01.x; w[1] = c1a[1] * x01.y; w[2] = c1a[2] * x23.x; w[3] = c1a[3] * x23.y;
                if (g.M2) {
                    const double* m2 = g.M2 + a + b * g.ldm2;
                    const v2d y01 = *reinterpret_cast<const v2d*>(m2), y23 = *reinterpret_cast<const v2d*>(m2 + 2);
                    w[0] = fma(c2a[0], y01.x, w[0]); w[1] = fma(c2a[1], y01.y, w[1]);
                    w[2] = fma(c2a[2], y23.x, w[2]); w[3] = fma(c2a[3], y23.y, w[3]);
                }
                if (g.u) {
                    const double vb = g.v[b];
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = fma(ua[r], vb, w[r]);
                }
            } else if (b < g.q) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long ar = a + r;
                    if (ar < g.p) {
                        // in symmetric mode only the lower triangle of M1 is valid
                        const long ia = (g.sym && ar < b) ? b : ar, ib = (g.sym && ar < b) ? ar : b;
                        double x = c1a[r] * g.M1[ia + ib * g.ldm1];
                        if (g.M2) x += c2a[r] * g.M2[ar + b * g.ldm2];
                        if (g.u) x += ua[r] * g.v[b];
                        if (ar == b) tr += x;
                        w[r] = x;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                qk[c][r] = symw * w[r] * (g.sf2 * exp(-0.5 * r2[c][r]));
                sK += qk[c][r];
            }
        }
        // per-dimension sums for dims [dim0, dim0+DC), DCH dimensions in LDS at a time
        auto dims_chunk = [&](auto ch_tag) {
            constexpr int ch = decltype(ch_tag)::value;
            constexpr int cw = DC - ch * DCH < DCH ? DC - ch * DCH : DCH;
            if (!single) {
                __syncthreads();
                stage(sa, sb, g.Xt1, g.ld1, g.p, a0, g.Xt2, g.ld2, g.q, b0, g.dim0 + ch * DCH, cw, g.dpad);
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < cw; ++i) {
                const v2d a01 = *reinterpret_cast<const v2d*>(&sa[i][4 * rg]), a23 = *reinterpret_cast<const v2d*>(&sa[i][4 * rg + 2]);
                const v2d b01 = *reinterpret_cast<const v2d*>(&sb[i][4 * cg]), b23 = *reinterpret_cast<const v2d*>(&sb[i][4 * cg + 2]);
                const double xa[4] = {a01.x, a01.y, a23.x, a23.y}, xb[4] = {b01.x, b01.y, b23.x, b23.y};
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double t = xa[r] - xb[c];
                        s = fma(qk[c][r], t * t, s);
                    }
                acc[ch * DCH + i] += s;
            }
        };
        dims_chunk(std::integral_constant<int, 0>{});
        if constexpr (DC > DCH) dims_chunk(std::integral_constant<int, 1>{});
        if constexpr (DL > 0) {
            __syncthreads();
            stage(sa, sb, g.Xt1, g.ld1, g.p, a0, g.Xt2, g.ld2, g.q, b0, g.dim0 + DC, DL, g.dpad);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < DL; ++i) {
                const v2d a01 = *reinterpret_cast<const v2d*>(&sa[i][4 * rg]), a23 = *reinterpret_cast<const v2d*>(&sa[i][4 * rg + 2]);
                const v2d b01 = *reinterpret_cast<const v2d*>(&sb[i][4 * cg]), b23 = *reinterpret_cast<const v2d*>(&sb[i][4 * cg + 2]);
                const double xa[4] = {a01.x, a01.y, a23.x, a23.y}, xb[4] = {b01.x, b01.y, b23.x, b23.y};
                double s = 0.0;
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double t = xa[r] - xb[c];
                        s = fma(qk[c][r], t * t, s);
                    }
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                if ((threadIdx.x & 3) == 0) lacc[i][threadIdx.x >> 2] += s;
            }
        }
    }
    // block reduction -> one partial row per workgroup
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    auto wave_sum = [](double v) {
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
#pragma unroll
    for (int i = 0; i < DC; ++i) {
        const double s = wave_sum(acc[i]);
        if (tx == 0) red[ty][i] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DL; ++i) {
        const double s = wave_sum(ty == 0 ? lacc[i][tx] : 0.0);
        if (tx == 0) red[ty][DC + i] = s;
    }
    {
        const double s1 = wave_sum(sK), s2 = wave_sum(tr);
        if (tx == 0) {
            red[ty][DC + DL] = s1;
            red[ty][DC + DL + 1] = s2;
        }
    }
    __syncthreads();
    if (threadIdx.x < DC + DL + 2) {
        const int i = threadIdx.x;
        g.partial[(long)blockIdx.x * (DC + DL + 2) + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}

// out[off_i] = sum over workgroups of partial[wg][i]
__global__ void trace_finish_kernel(const double* __restrict__ partial, int nwg, int width, int dim0,
                                    int ndims, int d, int write_tail, double* __restrict__ out) {
    const int i = blockIdx.x;  // column of the partial table
    __shared__ double red[256];
    double s = 0.0;
    for (int w = threadIdx.x; w < nwg; w += 256) s += partial[(long)w * width + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (i < ndims) {
            if (dim0 + i < d) out[dim0 + i] = red[0];
        } else if (write_tail) {
            out[d + (i - ndims)] = red[0];  // out[d] = sum w K, out[d+1] = tr(w)
        }
    }
}


// Cov::dk_dhyp (Cov/Cov.h:15-16; CovSEard.cpp:12-22, CovSEiso.cpp:12-31): ONE materialised dK from a K that is already
// there, for callers of seam (1) (the training path never forms it).  mode 0: K o ((x1_i - x2_i) / l_i)^2 for dimension
// dim (ARD, idx < d); mode 1: K o r2 with r2 over all dimensions (iso, idx = 0); mode 2: 2 K (the sf hyper).
__global__ __launch_bounds__(256) void dk_dhyp_kernel(long p, long q, int mode, int dim, int d, const double* __restrict__ Xt1,
                                                      long ld1, const double* __restrict__ Xt2, long ld2,
                                                      const double* __restrict__ K, long ldk, double* __restrict__ out, long ldo) {
    const long a = (long)blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (a >= p || b >= q) return;
    const double k = K[a + b * ldk];
    double v;
    if (mode == 2) {
        v = 2 * k;
    } else if (mode == 0) {
        const double t = Xt1[(long)dim * ld1 + a] - Xt2[(long)dim * ld2 + b];
        v = k * (t * t);
    } else {
        double r2 = 0.0;
        for (int i = 0; i < d; ++i) {
            const double t = Xt1[(long)i * ld1 + a] - Xt2[(long)i * ld2 + b];
            r2 += t * t;   // CovSEiso.cpp:17-19 sums the squares in this order
        }
        v = k * r2;
    }
    out[a + b * ldo] = v;
}
}  // namespace

int gpx_scale_inputs(hipStream_t st, const double* X, long n, int d, int dpad, const ScaleArgs& sc,
                     double* Xt, long ldx) {
    if (n <= 0) return GPX_OK;
    hipLaunchKernelGGL(scale_inputs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, X, n, d, dpad,
                       sc, Xt, ldx);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

int gpx_cov_build(hipStream_t st, const double* Xt1, long ld1, long p, const double* Xt2, long ld2, long q,
                  int dpad, double sf2, double* out, long ldo, int lower, double diag_add) {
    if (p <= 0 || q <= 0) return GPX_OK;
    if (lower && p != q) return GPX_ERR_ARG;
    const long tp = (p + 63) / 64, tq = (q + 63) / 64;
    const long ntiles = lower ? tp * (tp + 1) / 2 : tp * tq;
    if (ntiles > 0x7fffffffL) return GPX_ERR_ARG;
    const int vec = ((reinterpret_cast<uintptr_t>(out) & 15) == 0 && ldo % 2 == 0) ? 1 : 0;
    const long slot = pp_open(st, 0, lower ? 4.0 * (double)p * (double)(p + 1) : 8.0 * (double)p * (double)q);
    hipLaunchKernelGGL(cov_build_kernel, dim3((unsigned)ntiles), dim3(256), 0, st, Xt1, ld1, p, Xt2, ld2, q,
                       dpad, sf2, out, ldo, lower, (int)tp, diag_add, vec);
    pp_close(st, slot);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

int gpx_pair_trace(hipStream_t st, TraceArgs g, int d, double* partial_ws, long ws_doubles, double* out) {
    if (g.p <= 0 || g.q <= 0) return GPX_ERR_ARG;
    if (g.sym && g.p != g.q) return GPX_ERR_ARG;
    const long tp = (g.p + 63) / 64, tq = (g.q + 63) / 64;
    const long ntiles = g.sym ? tp * (tp + 1) / 2 : tp * tq;
    if (ntiles > 0x7fffffffL) return GPX_ERR_ARG;
    g.ntiles = (int)ntiles;
    g.tiles_p = (int)tp;
    g.partial = partial_ws;
    g.vec = ((reinterpret_cast<uintptr_t>(g.M1) & 15) == 0 && g.ldm1 % 2 == 0 &&
             (!g.M2 || ((reinterpret_cast<uintptr_t>(g.M2) & 15) == 0 && g.ldm2 % 2 == 0))) ? 1 : 0;
    // up to 32 dimensions per pass in registers, plus up to 32 more in LDS accumulators when there are
    // more: d <= 64 is ONE pass over the weights
    const int dc = g.dpad <= 8 ? 8 : g.dpad <= 16 ? 16 : g.dpad <= 24 ? 24 : 32;
    const long slot = pp_open(st, 1, (g.sym ? 4.0 * (double)g.p * (double)(g.p + 1) : 8.0 * (double)g.p * (double)g.q) * (g.M2 ? 2.0 : 1.0));
    for (int dim0 = 0; dim0 < g.dpad; dim0 += 64) {
        const int rest = g.dpad - dim0;
        const int dl = rest <= 32 ? 0 : rest <= 48 ? 16 : rest <= 56 ? 24 : 32;  // (a DL = 8 instance spills)
        const int width = dc + dl + 2;
        long nwg = ntiles < 2048 ? ntiles : 2048;
        if (nwg * width > ws_doubles) nwg = ws_doubles / width;
        if (nwg <= 0) return GPX_ERR_ARG;
        g.dim0 = dim0;
        const dim3 gr((unsigned)nwg), bl(256);
        if (dl == 0) {
            switch (dc) {
                case 8: hipLaunchKernelGGL((pair_trace_kernel<8, 0>), gr, bl, 0, st, g); break;
                case 16: hipLaunchKernelGGL((pair_trace_kernel<16, 0>), gr, bl, 0, st, g); break;
                case 24: hipLaunchKernelGGL((pair_trace_kernel<24, 0>), gr, bl, 0, st, g); break;
                default: hipLaunchKernelGGL((pair_trace_kernel<32, 0>), gr, bl, 0, st, g); break;
            }
        } else {
            switch (dl) {
                case 16: hipLaunchKernelGGL((pair_trace_kernel<32, 16>), gr, bl, 0, st, g); break;
                case 24: hipLaunchKernelGGL((pair_trace_kernel<32, 24>), gr, bl, 0, st, g); break;
                default: hipLaunchKernelGGL((pair_trace_kernel<32, 32>), gr, bl, 0, st, g); break;
            }
        }
        hipLaunchKernelGGL(trace_finish_kernel, dim3(width), dim3(256), 0, st, partial_ws, (int)nwg, width, dim0,
                           dc + dl, d, dim0 == 0 ? 1 : 0, out);
    }
    pp_close(st, slot);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

long gpx_pair_prof_open(hipStream_t st, int cls, double bytes) { return pp_open(st, cls, bytes); }
void gpx_pair_prof_close(hipStream_t st, long slot) { pp_close(st, slot); }

int gpx_pair_prof_begin(double min_bytes, int max_launches) {
    std::lock_guard<std::mutex> lk(p_prof.mu);
    HIP_TRY(hipGetDevice(&p_prof.dev));
    if (p_prof.ev.size() < (size_t)max_launches * 2) {
        const size_t old = p_prof.ev.size();
        p_prof.ev.resize((size_t)max_launches * 2);
        for (size_t i = old; i < p_prof.ev.size(); ++i) HIP_TRY(hipEventCreate(&p_prof.ev[i]));
    }
    p_prof.on = true;
    p_prof.min_bytes = min_bytes;
    p_prof.used = 0;
    p_prof.bytes.clear();
    p_prof.cls.clear();
    return GPX_OK;
}

int gpx_pair_prof_end(int cls, double* total_ms, double* total_bytes, long* launches) {
    std::lock_guard<std::mutex> lk(p_prof.mu);
    p_prof.on = false;
    double ms = 0, by = 0;
    long cnt = 0;
    for (size_t i = 0; i + 1 < p_prof.used; i += 2) {
        if (p_prof.cls[i / 2] != cls) continue;
        HIP_TRY(hipEventSynchronize(p_prof.ev[i + 1]));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, p_prof.ev[i], p_prof.ev[i + 1]));
        ms += t;
        by += p_prof.bytes[i / 2];
        ++cnt;
    }
    *total_ms = ms;
    *total_bytes = by;
    *launches = cnt;
    return GPX_OK;
}

int gpx_dk_dhyp(hipStream_t st, long p, long q, int mode, int dim, int d, const double* Xt1, long ld1, const double* Xt2, long ld2,
                const double* K, long ldk, double* out, long ldo) {
    if (p <= 0 || q <= 0) return GPX_OK;
    if (q > 65535) return GPX_ERR_ARG;
    hipLaunchKernelGGL(dk_dhyp_kernel, dim3((unsigned)((p + 255) / 256), (unsigned)q), dim3(256), 0, st, p, q, mode, dim, d, Xt1, ld1,
                       Xt2, ld2, K, ldk, out, ldo);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}
