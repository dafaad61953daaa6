// gemm_f64.hip — fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64), gfx950 only.
//
// One kernel serves every dense contraction of the GP path (Cholesky trailing SYRK/GEMM, the
// triangular inverse, W^T W, the tall-skinny FITC/VFE products): C = alpha*op(A)*op(B)^T + beta*C,
// column-major, either operand M/N-contiguous or K-contiguous, optional lower-triangle-only
// output and per-tile K ranges that skip the zero half of triangular operands.
//
// Layout of one workgroup: 256 threads = 4 waves in a 2(M) x 2(N) grid, block tile
// (32*TM) x (32*TN) with BK = 16; each wave owns TM x TN MFMA tiles of 16x16 (4x4 -> 64x64 per
// wave, 128 accumulator VGPRs).  Three K loops, chosen per tile:
//   1. interior 128x128 tiles: global -> LDS directly (global_load_lds_dwordx4 issued from inline
//      assembly, two LDS stages, one barrier per K tile, explicit vmcnt waits, L2 warm-up load);
//   2. other interior tiles: global -> registers (next K tile in flight during the MFMAs) -> LDS;
//   3. edge tiles: the same with bounds tests.
// The MFMA's "A" operand is fed from matrix B and its "B" operand from matrix A, so that the
// lane index (lane & 15) runs along M: accumulator stores and the beta*C read are 128-B
// contiguous per 16 lanes in the column-major C.
// Tile order (blockIdx -> tile): split-K slabs (XCD-private when the slab count is a multiple of
// 8), per-XCD rows for tall-skinny outputs, one 8x8 square of tiles per XCD with alternating K
// direction for large outputs, LOWER products as a strictly-lower launch + a diagonal band launch.
#include <array>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <type_traits>
#include "gpx_internal.h"
#include "tile_order.h"

namespace {

struct GemmArgs : TileOrder {
    const double* A;
    const double* B;
    double* C;
    long lda, ldb, ldc;
    int M, N, K;
    double alpha, beta;
    long slab;     // doubles between split-K slabs
    int fastA, fastB;  // base pointer and ld allow 16-B vector loads
    const double* kscale;  // optional: op(A)(m,k) is multiplied by kscale[k] on its way into the MFMAs (K-contiguous A only)
    // optional row-dot epilogue (128-tiles, beta == 0, no split-K): per part p of the output (64 columns: those one wave of the tile owns, see sub16)
    //   R1[p*ldr + m] = sum_{n in part} C(m,n) * E(m,n)   (E == nullptr: E = C, i.e. row sums of squares)
    //   R2[p*ldr + m] = sum_{n in part} E(m,n) * evec[n]   (when E and evec are given)
    // so that the row statistics of a tall-skinny product do not cost a pass of their own over it
    // or the operand dot (RD == 3: E == R1 == nullptr): R2[m] = sum_k A(m,k) * evec[k], formed from the A fragments of the
    // tiles of the first tile column on their way to the MFMAs (Kxu * a of VFE.cpp:176 without a pass over Kxu)
    const double* E; long lde;
    const double* evec;
    double* R1; double* R2; long ldr;
    // Row-panel layout of the tall n x m matrices of the sparse models (see GemmPanels in gpx_internal.h): the CONTIGUOUS
    // index c of an operand (m for an M-contiguous A / C / E, n for an N-contiguous B, k for a K-contiguous A / B) is cut
    // into panels of pan rows, element (c, s) lives at (c / pan) * pst + c % pan + s * ld.  A workgroup never leaves a
    // panel (pan is a multiple of 128; K-panelled products are split-K with kper K tiles = one panel per slab), so one
    // base-pointer shift per workgroup, (c / pan) * (pst - pan), makes every address formula below valid as it stands.
    long panA, pstA, panB, pstB, panC, pstC, panE, pstE;
    int kper;   // K tiles per split-K slab (0: ceil(K tiles / slabs))
};

// Offset of a wave's j-th 16-column group inside the block tile, relative to the wave's base column.  With four groups per
// wave (128-tiles) the column ownership is INTERLEAVED: groups 0, 1 lie in the first 64 columns of the tile and groups
// 2, 3 in the second 64 (wave base 32 (wave >> 1)), so that a K range that only concerns one half of the tile's columns
// (triangular operands cut at 64) halves the work of EVERY wave alike.  (Rows stay contiguous per wave, 64 (wave & 1) +
// 16 i: interleaving them as well cost the exact n = 65536 evaluation 1.2 %, same-box A/B, and nothing uses it.)
template <int T>
__device__ __forceinline__ constexpr int sub16(int i) { return T == 4 ? (i & 1) * 16 + (i >> 1) * 64 : 16 * i; }

// the row-dot epilogue of one wave: acc[j][i][r] = C(mb + 16 i + l15, nb + sub16(j) + l4 + 4 r) / alpha; part: the wave's
// slot among the 2 * ceil(N / 128) row-dot parts
// RD == 1: E = C (row sums of squares, R1 only); RD == 2: E, evec, R1 and R2 all given
template <int TM, int TN, bool FULL, int RD>
__device__ __forceinline__ void rowdot_epilogue(const GemmArgs& g, const double* gE, const v4d (&acc)[TN][TM], int mb, int nb, int part, int l15, int l4) {
    double s1[TM], s2[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) s1[i] = s2[i] = 0.0;
    // one running pointer per operand (the 16 i offsets are immediates): no per-element 64-bit address registers
    const double* ep = (RD == 2 && FULL) ? gE + (mb + l15) + (long)(nb + l4) * g.lde : nullptr;
    const double* vp = (RD == 2 && FULL) ? g.evec + nb + l4 : nullptr;
    const long estep = 4 * g.lde;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool nin = FULL || nb + sub16<TN>(j) + l4 + 4 * r < g.N;
            if (RD == 2) {
                double e[TM], ev;
                if (FULL) {
                    if (TN == 4 && j == 2 && r == 0) ep += 8 * estep;   // groups 2, 3: the second 64 columns of the tile
                    ev = vp[sub16<TN>(j) + 4 * r];
#pragma unroll
                    for (int i = 0; i < TM; ++i) e[i] = ep[16 * i];
                    ep += estep;
                } else {   // edge tiles: clamped addresses and 0/1 masks instead of guarded loads (no divergent branches)
                    const int n = nb + sub16<TN>(j) + l4 + 4 * r;
                    const double nmask = n < g.N ? 1.0 : 0.0;
                    const double* col = gE + (long)min(n, g.N - 1) * g.lde;
                    ev = g.evec[min(n, g.N - 1)] * nmask;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int m = mb + 16 * i + l15;
                        e[i] = col[min(m, g.M - 1)] * (m < g.M ? nmask : 0.0);
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    s1[i] = fma(g.alpha * acc[j][i][r], e[i], s1[i]);
                    s2[i] = fma(e[i], ev, s2[i]);
                }
                if (r & 1) __builtin_amdgcn_sched_barrier(0);   // at most two columns of E in flight: all accumulators are live here
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const double c = nin ? g.alpha * acc[j][i][r] : 0.0;
                    s1[i] = fma(c, c, s1[i]);
                }
            }
        }
    const long off = (long)part * g.ldr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        double a = s1[i], b = s2[i];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (RD == 2) {
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
        }
        const int m = mb + 16 * i + l15;
        if (l4 == 0 && (FULL || m < g.M)) {
            g.R1[off + m] = a;
            if (RD == 2) g.R2[off + m] = b;
        }
    }
}

// full 128-tile of the RD == 2 variant: per 16-column group the 16 loads of E are issued first, the group's C values
// are stored while they are in flight, then the two row dots are formed (separate store and dot passes leave the
// matrix pipe idle for 64 load round trips in 8 batches per tile: Cu GEMM 29.8 -> 32.2 ms at n = 2^20)
template <int TM, int TN>
__device__ __forceinline__ void store_rowdot_full(const GemmArgs& g, const double* gE, const v4d (&acc)[TN][TM], double* Cb, int mb, int nb, int part, int l15, int l4) {
    double s1[TM], s2[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) s1[i] = s2[i] = 0.0;
    const double* ep = gE + (mb + l15) + (long)(nb + l4) * g.lde;
    const double* vp = g.evec + nb + l4;
    double* cp = Cb + (mb + l15) + (long)(nb + l4) * g.ldc;
    const long estep = 4 * g.lde, cstep = 4 * g.ldc;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        double e[4][TM], ev[4];
        if (TN == 4 && j == 2) {   // groups 2, 3: the second 64 columns of the tile
            ep += 8 * estep;
            cp += 8 * cstep;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ev[r] = vp[sub16<TN>(j) + 4 * r];
#pragma unroll
            for (int i = 0; i < TM; ++i) e[r][i] = ep[16 * i];
            ep += estep;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < TM; ++i) cp[16 * i] = g.alpha * acc[j][i][r];
            cp += cstep;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                s1[i] = fma(g.alpha * acc[j][i][r], e[r][i], s1[i]);
                s2[i] = fma(e[r][i], ev[r], s2[i]);
            }
        __builtin_amdgcn_sched_barrier(0);   // one group of E in flight at a time: all accumulators are live here
    }
    const long off = (long)part * g.ldr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        double a = s1[i], b = s2[i];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (l4 == 0) {
            g.R1[off + mb + 16 * i + l15] = a;
            g.R2[off + mb + 16 * i + l15] = b;
        }
    }
}

constexpr int BK = 16;

// global -> registers: BMN*BK doubles per operand tile, BMN/32 double2 per thread
template <int BMN, bool KC, int NT = 256>
__device__ __forceinline__ void gload(const double* __restrict__ P, long ld, int mn0, int k0, int MN,
                                      int kend, bool fast, v2d (&r)[BMN * 8 / NT]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < BMN * 8 / NT; ++i) {
        const int e = tid + NT * i;
        if (!KC) {
            const int m = mn0 + 2 * (e % (BMN / 2));
            const int k = k0 + e / (BMN / 2);
            const double* p = P + m + (long)k * ld;
            if (fast) {
                r[i] = *reinterpret_cast<const v2d*>(p);
            } else {
                const bool kin = k < kend;
                r[i].x = (kin && m < MN) ? p[0] : 0.0;
                r[i].y = (kin && m + 1 < MN) ? p[1] : 0.0;
            }
        } else {
            const int k = k0 + 2 * (e % 8);
            const int m = mn0 + e / 8;
            const double* p = P + k + (long)m * ld;
            if (fast) {
                r[i] = *reinterpret_cast<const v2d*>(p);
            } else {
                const bool min_ = m < MN;
                r[i].x = (min_ && k < kend) ? p[0] : 0.0;
                r[i].y = (min_ && k + 1 < kend) ? p[1] : 0.0;
            }
        }
    }
}

// registers -> LDS.  M-contiguous image: [BK][BMN+16] (row stride = 128 B mod 256 B, so the four
// k-rows one MFMA fragment read touches fall on disjoint banks); K-contiguous image: [BMN][BK+2]
// (row stride 144 B = 16*9: sixteen consecutive rows cover all 16-B bank slots once).
template <int BMN, bool KC, int NT = 256>
__device__ __forceinline__ void sstore(double* __restrict__ s, const v2d (&r)[BMN * 8 / NT]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < BMN * 8 / NT; ++i) {
        const int e = tid + NT * i;
        if (!KC) {
            const int m2 = e % (BMN / 2), k = e / (BMN / 2);
            *reinterpret_cast<v2d*>(&s[k * (BMN + 16) + 2 * m2]) = r[i];
        } else {
            const int k2 = e % 8, m = e / 8;
            *reinterpret_cast<v2d*>(&s[m * (BK + 2) + 2 * k2]) = r[i];
        }
    }
}

template <int BMN, bool KC>
__device__ __forceinline__ double sfrag(const double* __restrict__ s, int row, int k) {
    return KC ? s[row * (BK + 2) + k] : s[k * (BMN + 16) + row];
}

template <int TM, int TN, bool AKC, bool BKC, int RD = 0, int SK = 0>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int block) {
    constexpr int NT = 256;
    constexpr int BM = 32 * TM, BN = 32 * TN;
    constexpr int NLA = BM * 8 / NT, NLB = BN * 8 / NT;  // 16-B loads per thread and operand tile
    constexpr int LA = AKC ? BM * (BK + 2) : BK * (BM + 16);
    constexpr int LB = BKC ? BN * (BK + 2) : BK * (BN + 16);
    __shared__ __attribute__((aligned(16))) double smem[LA + LB];
    __shared__ __attribute__((aligned(16))) double smem1[LA + LB];  // 2nd stage (direct-to-LDS loop of the 128-tiles, ring loop of the 64-tiles)
    double* sA = smem;
    double* sB = smem + LA;

    int bi, bj, ks;
    bool desc;
    if (!tile_of_block(g, block, bi, bj, ks, desc)) return;
    const int m0 = __builtin_amdgcn_readfirstlane(bi * BM), n0 = __builtin_amdgcn_readfirstlane(bj * BN);

    int klo = 0, khi = g.K;
    if (g.flags & GEMM_KLO_M) klo = max(klo, m0);
    if (g.flags & GEMM_KLO_N) klo = max(klo, n0);
    if (g.flags & GEMM_KHI_M) khi = min(khi, m0 + BM);
    if (g.flags & GEMM_KHI_N) khi = min(khi, n0 + BN);
    klo = klo / BK * BK;
    int nkt = khi > klo ? (khi - klo + BK - 1) / BK : 0;
    int kt0 = 0, kt1 = nkt;
    if (g.splitk > 1) {
        const int per = g.kper ? g.kper : (nkt + g.splitk - 1) / g.splitk;
        kt0 = min(nkt, ks * per);
        kt1 = min(nkt, kt0 + per);
    }

    // row-panel layout: one base shift per workgroup and operand (see GemmArgs)
    const double* gA = g.A;
    const double* gB = g.B;
    double* gC = g.C;
    const double* gE = g.E;
    if (g.panA) gA += (long)((AKC ? klo + kt0 * BK : m0) / g.panA) * (g.pstA - g.panA);
    if (g.panB) gB += (long)((BKC ? klo + kt0 * BK : n0) / g.panB) * (g.pstB - g.panB);
    if (g.panC) gC += (long)(m0 / g.panC) * (g.pstC - g.panC);
    if (g.panE) gE += (long)(m0 / g.panE) * (g.pstE - g.panE);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave & 1) * (16 * TM), wn = (wave >> 1) * (TN == 4 ? 32 : 16 * TN);   // see sub16
    const int l15 = lane & 15, l4 = lane >> 4;

    v4d acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = (v4d){0.0, 0.0, 0.0, 0.0};

    double sv[TM];   // RD == 3, tiles of the first tile column: this lane's share of sum_k A(m0 + wm + 16 i + l15, k) evec[k]
#pragma unroll
    for (int i = 0; i < TM; ++i) sv[i] = 0.0;
    bool dot_done = false;
    const bool tfA = g.fastA && (m0 + BM <= g.M);
    const bool tfB = g.fastB && (n0 + BN <= g.N);
    v2d ra[NLA], rb[NLB];
    const int tid = threadIdx.x;
    // Interior tiles (every operand tile of the K range is full and 16-B aligned; true for all but the
    // edge tiles) take a loop whose load phase is 16 global loads and nothing else: a uniform 64-bit
    // base per operand that advances by one K tile per step (scalar ALU) plus ONE loop-invariant
    // 32-bit byte offset per thread.  Vector instructions issued next to the MFMAs of the co-resident
    // wave delay them, so the address arithmetic of the generic path below costs matrix-pipe time.
    const bool interior = tfA && tfB && (klo + kt1 * BK <= g.K) && kt0 < kt1 && !(g.flags & GEMM_NO_FAST32);
    if (TM == 4 && TN == 4 && interior && !(g.flags & GEMM_NO_DLDS)) {
        // Direct-to-LDS loop: global_load_lds_dwordx4 writes 64 lanes x 16 B = 1 KB straight into LDS
        // (no staging registers, no ds_write, no address VALU beyond one add per load), two LDS stages,
        // ONE barrier per K tile.  An M-contiguous wave-load is one k-row of the [16][BM+16] image; a
        // K-contiguous one covers 8 rows of an unpadded [BM][16] image whose 16-B chunks are XOR-swizzled
        // by (row >> 1) & 7 so that the fragment reads stay conflict-free.
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const bool down = __builtin_amdgcn_readfirstlane(desc);
        const long k00 = klo + (down ? kt1 - 1 : kt0) * BK;
        const char* pa = reinterpret_cast<const char*>(gA) + (AKC ? k00 + (long)m0 * g.lda : m0 + k00 * g.lda) * 8;
        const char* pb = reinterpret_cast<const char*>(gB) + (BKC ? k00 + (long)n0 * g.ldb : n0 + k00 * g.ldb) * 8;
        const long ka = (AKC ? (long)BK : BK * g.lda) * (down ? -8 : 8), kb = (BKC ? (long)BK : BK * g.ldb) * (down ? -8 : 8);
        // wave-load q = wv + 4 i (i = 0..3): M-contiguous -> k-row q; K-contiguous -> rows 8q .. 8q+7
        const long ia = (AKC ? 32 * g.lda : 4 * g.lda) * 8, ib = (BKC ? 32 * g.ldb : 4 * g.ldb) * 8;
        const int swz = (4 * wv + (lane >> 4)) & 7;  // ((row >> 1) & 7) of this lane's row, the same for every i
        const unsigned ta = (unsigned)((AKC ? 2 * ((lane & 7) ^ swz) + (long)(8 * wv + (lane >> 3)) * g.lda
                                            : 2 * lane + (long)wv * g.lda) * 8);
        const unsigned tb = (unsigned)((BKC ? 2 * ((lane & 7) ^ swz) + (long)(8 * wv + (lane >> 3)) * g.ldb
                                            : 2 * lane + (long)wv * g.ldb) * 8);
        constexpr int QA = AKC ? 128 : BM + 16, QB = BKC ? 128 : BN + 16;  // doubles per wave-load in the LDS image
        // The loads are issued from inline assembly: the compiler's wait-count pass would otherwise
        // wait for EVERY outstanding LDS-DMA load before each barrier, and the address form below (uniform
        // 64-bit base in SGPRs + one loop-invariant 32-bit VGPR offset) needs no vector add per load.
        // All vmcnt waits of this loop are therefore explicit.
        auto lds_addr = [](const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; };
        auto dlds16 = [](const char* base, unsigned voff, unsigned dst) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(dst)
                         : "memory", "m0");
        };
        auto fragA = [&](const double* st, int row, int k) {
            return AKC ? st[row * 16 + ((((k >> 1) ^ (row >> 1)) & 7) << 1) + (k & 1)] : st[k * (BM + 16) + row];
        };
        auto fragB = [&](const double* st, int row, int k) {
            return BKC ? st[LA + row * 16 + ((((k >> 1) ^ (row >> 1)) & 7) << 1) + (k & 1)] : st[LA + k * (BN + 16) + row];
        };
        // L2 / TLB warm-up two tiles ahead: once operands stream from HBM over many pages (n >= 32768) a
        // one-tile prefetch distance no longer covers the fetch (68.5 TFLOP/s against 73 with L2-resident
        // operands).  Each thread touches ONE 128-B line of the tile after next with a 4-byte direct-to-LDS
        // load into a dummy LDS word (no destination register to keep alive); the waits below leave that
        // one load outstanding (vmcnt(1)), so only the tile that is needed next is waited for.
        __shared__ int warm_sink[256];
        const int th = tid & 127;
        const bool wa = wv < 2;  // waves 0-1 warm A, waves 2-3 warm B (wave-uniform: the base stays in SGPRs)
        const long wld = wa ? g.lda : g.ldb;
        const bool wkc = wa ? AKC : BKC;
        const unsigned woff = (unsigned)(wkc ? (long)th * wld * 8 : (long)(th >> 3) * wld * 8 + (th & 7) * 128);
        const unsigned wdst = lds_addr(warm_sink + 64 * wv);
        auto warm = [&]() {  // pa / pb point at the tile after the one just issued
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(woff), "s"(wa ? pa : pb), "s"(wdst)
                         : "memory", "m0");
        };
        // K-dimension scale (A^T diag(s) B, K-contiguous A): the 16 scale values of a K tile travel to LDS
        // beside the operand tiles (wave 0: one 4-byte load per lane, the upper 32 lanes write a second
        // copy) and multiply the A fragments on their way to the MFMAs.
        __shared__ __attribute__((aligned(16))) double sks[2][32];
        const char* ps = reinterpret_cast<const char*>(RD == 3 ? g.evec : g.kscale) + k00 * 8;
        const long ksd = down ? -(long)BK * 8 : (long)BK * 8;
        const unsigned soff = (unsigned)((lane & 31) * 4);
        // Half-tile skipping (see sub16; SK != 0 instantiations, chosen by the host from the flags).  A K tile of a
        // triangular operand that only concerns the first / second 64 columns of the tile (K ranges cut at 64 instead of 128:
        // 1/18 of the flops of a 1024-wide triangular product) is left out by EVERY wave alike.  The K loop is cut into
        // three consecutive loops, each with ONE compile-time shape of the multiply - half, full, half - and run-time trip
        // counts (no branch inside a loop: branches between differently shaped multiplies, a switch or if-chain per K
        // tile, cost the register allocator 300 - 1000 spilled registers and the FITC evaluation 113 -> 141 ms).
        //   SK 1: GEMM_KHI_N, K tiles with k >= n0 + 64 multiply column groups 2, 3 only
        //   SK 2: GEMM_KLO_N, K tiles with k <  n0 + 64 multiply column groups 0, 1 only
        // Measured at n = 2^20, m = 1024: V 17.73 -> 17.27 ms, B 17.73 -> 17.44 ms (half of the 1/18: the half-shaped K
        // tiles load as much as the full ones).  A third shape for the diagonal tiles of a LOWER product (all but rows
        // 0-63 x columns 64-127) made the split-K Gram products SLOWER, 17.7 -> 18.6 ms: the diagonal tiles run ahead of the
        // other tiles of their slab and fetch their operands from HBM a second time; not kept.
        // RD == 2: the epilogue reads the 128 x 128 tile of E in four batches per wave, one L2-miss round trip each.
        // With the loads of the LAST K tile, every thread touches four of the tile's 1024 lines (4-byte direct-to-LDS
        // loads into the dummy word, as the operand warm-up), two K tiles before the epilogue needs them.
        const char* pe = RD == 2 ? reinterpret_cast<const char*>(gE) + ((long)m0 + (long)n0 * g.lde) * 8 : nullptr;
        const unsigned eoff = (unsigned)(((tid & 7) * 16 + (long)(tid >> 3) * g.lde) * 8);   // 16-row chunk, column tid / 8
        auto warm_e = [&]() {
#pragma unroll
            for (int q = 0; q < 4; ++q)   // columns 32 q + tid / 8
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(eoff), "s"(pe + q * 32 * g.lde * 8),
                             "s"(wdst)
                             : "memory", "m0");
        };
        auto run = [&](auto SC, auto DOT) {
            constexpr bool scaled = decltype(SC)::value;
            constexpr bool dot = decltype(DOT)::value;   // the vector of this K tile travels like the K scale
            auto issue = [&](int stage) {
                double* st = stage ? smem1 : smem;
                const unsigned dA = lds_addr(st + wv * QA), dB = lds_addr(st + LA + wv * QB);
#pragma unroll
                for (int i = 0; i < 4; ++i) dlds16(pa + i * ia, ta, dA + 4 * i * QA * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i) dlds16(pb + i * ib, tb, dB + 4 * i * QB * 8);
                if (scaled || dot) {
                    if (wv == 0)
                        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(soff), "s"(ps),
                                     "s"(lds_addr(sks[stage]))
                                     : "memory", "m0");
                    ps += ksd;
                }
                pa += ka;
                pb += kb;
            };
            // mode 0: everything; 1: column groups 0, 1; 2: column groups 2, 3
            auto multiply = [&](int stage, auto MODE) {
                constexpr int mode = decltype(MODE)::value;
                constexpr int J0 = mode == 2 ? TN / 2 : 0, J1 = mode == 1 ? TN / 2 : TN;
                const double* st = stage ? smem1 : smem;
#pragma unroll
                for (int kk = 0; kk < BK / 4; ++kk) {
                    double fa[TM], fb[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[i] = fragA(st, wm + 16 * i + l15, kk * 4 + l4);
#pragma unroll
                    for (int j = J0; j < J1; ++j) fb[j] = fragB(st, wn + sub16<TN>(j) + l15, kk * 4 + l4);
                    if (scaled) {
                        const double sc = sks[stage][kk * 4 + l4];
#pragma unroll
                        for (int i = 0; i < TM; ++i) fa[i] *= sc;
                    }
                    if (dot) {
                        const double ek = sks[stage][kk * 4 + l4];
#pragma unroll
                        for (int i = 0; i < TM; ++i) sv[i] = fma(fa[i], ek, sv[i]);
                    }
#pragma unroll
                    for (int j = J0; j < J1; ++j)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[j][i], 0, 0, 0);
                }
            };
            // `cnt` K tiles in one shape; rem: tiles not multiplied yet, all segments together.  cnt is even unless the
            // segment is the last one.  Waiting for a tile leaves the warm-up load issued after it outstanding, if there was one.
            int rem = kt1 - kt0;
            auto segment = [&](auto MODE, int cnt) {
                for (; cnt >= 2; cnt -= 2, rem -= 2) {
                    __builtin_amdgcn_s_waitcnt(0x0f71);  // vmcnt(1): this wave's part of the tile is in LDS
                    __syncthreads();                     // ... and every other wave's; stage 1 is free again
                    issue(1);
                    const bool more = rem > 2;
                    if (more) warm();
                    else if (RD == 2) warm_e();
                    multiply(0, MODE);
                    if (more) __builtin_amdgcn_s_waitcnt(0x0f71);
                    else if (RD == 2) __builtin_amdgcn_s_waitcnt(0x0f74);   // vmcnt(4): the four touches of E stay in flight
                    else __builtin_amdgcn_s_waitcnt(0x0f70);
                    __syncthreads();
                    if (more) {
                        issue(0);
                        if (rem > 3) warm();
                    }
                    multiply(1, MODE);
                }
                if (cnt == 1) {
                    __builtin_amdgcn_s_waitcnt(0x0f70);
                    __syncthreads();
                    multiply(0, MODE);
                    rem -= 1;
                }
            };
            int n_first = 0, n_last = 0;   // half-shaped K tiles before / after the full ones, in the order of the walk
            if (SK == 1 || SK == 2) {
                const int kbeg = klo + kt0 * BK, kend = klo + kt1 * BK, nmid = n0 + BN / 2;
                const int nh = SK == 1 ? min(max((kend - nmid) / BK, 0), rem) : min(max((nmid - kbeg) / BK, 0), rem);
                const bool at_end = (SK == 1) != down;   // SK 1: the half tiles are the last of an ascending walk
                if (at_end ? !((rem - nh) & 1) : !(nh & 1)) {   // a segment that another one follows has an even count
                    if (at_end) n_last = nh;
                    else n_first = nh;
                }
            }
            issue(0);
            if (rem > 1) warm();
            using HALF = std::integral_constant<int, SK == 1 ? 2 : SK == 2 ? 1 : 0>;
            if (SK != 0) segment(HALF{}, n_first);
            segment(std::integral_constant<int, 0>{}, rem - n_last);   // rem: n_first is done
            if (SK != 0) segment(HALF{}, n_last);
        };
        if (RD == 3 && bj == 0) {
            run(std::false_type{}, std::true_type{});
            dot_done = true;
        } else if (AKC && g.kscale != nullptr) {
            run(std::true_type{}, std::false_type{});
        } else {
            run(std::false_type{}, std::false_type{});
        }
    } else if (interior && TM <= 2 && TN <= 2) {
        // 64x64 tiles: the products of the factorisation chains below 2048 rows, a few tiles on a few CUs with nothing
        // co-resident to hide a load behind.  One K step of MFMAs is 0.43 us, an L2 / HBM load takes longer, so the
        // operand registers are a ring of TWO K tiles: the loads of tile kt + 2 are issued while tile kt is multiplied
        // (measured 0.75 us per K step with a one-tile distance, i.e. 0.3 us of exposed latency per step).
        const long k00 = klo + kt0 * BK;
        const char* pa = reinterpret_cast<const char*>(gA) + (AKC ? k00 + (long)m0 * g.lda : m0 + k00 * g.lda) * 8;
        const char* pb = reinterpret_cast<const char*>(gB) + (BKC ? k00 + (long)n0 * g.ldb : n0 + k00 * g.ldb) * 8;
        const long ia = (AKC ? (NT / 8) * g.lda : (2 * NT / BM) * g.lda) * 8;  // between a thread's consecutive loads
        const long ib = (BKC ? (NT / 8) * g.ldb : (2 * NT / BN) * g.ldb) * 8;
        const long ka = (AKC ? (long)BK : BK * g.lda) * 8;            // per K tile
        const long kb = (BKC ? (long)BK : BK * g.ldb) * 8;
        const unsigned ta = (unsigned)((AKC ? 2 * (tid & 7) + (long)(tid >> 3) * g.lda
                                            : 2 * (tid % (BM / 2)) + (long)(tid / (BM / 2)) * g.lda) * 8);
        const unsigned tb = (unsigned)((BKC ? 2 * (tid & 7) + (long)(tid >> 3) * g.ldb
                                            : 2 * (tid % (BN / 2)) + (long)(tid / (BN / 2)) * g.ldb) * 8);
        const bool scaled = AKC && g.kscale != nullptr;
        const double* ps = scaled ? g.kscale + k00 + 2 * (tid & 7) : nullptr;
        v2d ra1[NLA], rb1[NLB];
        v2d rs = (v2d){1.0, 1.0}, rs1 = (v2d){1.0, 1.0};
        auto fetch = [&](v2d (&xa)[NLA], v2d (&xb)[NLB], v2d& xs) {   // the tile pa / pb / ps point at, then advance
#pragma unroll
            for (int i = 0; i < NLA; ++i) xa[i] = *reinterpret_cast<const v2d*>(pa + i * ia + ta);
#pragma unroll
            for (int i = 0; i < NLB; ++i) xb[i] = *reinterpret_cast<const v2d*>(pb + i * ib + tb);
            if (scaled) {
                xs = *reinterpret_cast<const v2d*>(ps);
                ps += BK;
            }
            pa += ka;
            pb += kb;
        };
        // two LDS stages as well: tile kt + 1 is written to the other stage before tile kt is multiplied, ONE barrier per
        // K step, and the 16 fragment reads of a step are issued together
        auto stash = [&](double* st, v2d (&xa)[NLA], v2d (&xb)[NLB], v2d& xs, bool refill) {
            if (scaled) {
#pragma unroll
                for (int i = 0; i < NLA; ++i) xa[i] = xa[i] * xs;
            }
            sstore<BM, AKC, NT>(st, xa);
            sstore<BN, BKC, NT>(st + LA, xb);
            if (refill) fetch(xa, xb, xs);
        };
        auto multiply = [&](const double* st) {
            double fa[BK / 4][TM], fb[BK / 4][TN];
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[kk][i] = sfrag<BM, AKC>(st, wm + 16 * i + l15, kk * 4 + l4);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[kk][j] = sfrag<BN, BKC>(st + LA, wn + sub16<TN>(j) + l15, kk * 4 + l4);
            }
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kk][j], fa[kk][i], acc[j][i], 0, 0, 0);
        };
        fetch(ra, rb, rs);                              // tile kt0
        if (kt0 + 1 < kt1) fetch(ra1, rb1, rs1);        // tile kt0 + 1
        stash(smem, ra, rb, rs, kt0 + 2 < kt1);         // stage 0 <- tile kt0; ra <- tile kt0 + 2
        __syncthreads();
        for (int kt = kt0; kt < kt1; kt += 2) {
            if (kt + 1 < kt1) stash(smem1, ra1, rb1, rs1, kt + 3 < kt1);
            multiply(smem);
            __syncthreads();
            if (kt + 1 < kt1) {
                if (kt + 2 < kt1) stash(smem, ra, rb, rs, kt + 4 < kt1);
                multiply(smem1);
                __syncthreads();
            }
        }
    } else if (interior) {
        const long k00 = klo + kt0 * BK;
        const char* pa = reinterpret_cast<const char*>(gA) + (AKC ? k00 + (long)m0 * g.lda : m0 + k00 * g.lda) * 8;
        const char* pb = reinterpret_cast<const char*>(gB) + (BKC ? k00 + (long)n0 * g.ldb : n0 + k00 * g.ldb) * 8;
        const long ia = (AKC ? (NT / 8) * g.lda : (2 * NT / BM) * g.lda) * 8;  // between a thread's consecutive loads
        const long ib = (BKC ? (NT / 8) * g.ldb : (2 * NT / BN) * g.ldb) * 8;
        const long ka = (AKC ? (long)BK : BK * g.lda) * 8;            // per K tile
        const long kb = (BKC ? (long)BK : BK * g.ldb) * 8;
        const unsigned ta = (unsigned)((AKC ? 2 * (tid & 7) + (long)(tid >> 3) * g.lda
                                            : 2 * (tid % (BM / 2)) + (long)(tid / (BM / 2)) * g.lda) * 8);
        const unsigned tb = (unsigned)((BKC ? 2 * (tid & 7) + (long)(tid >> 3) * g.ldb
                                            : 2 * (tid % (BN / 2)) + (long)(tid / (BN / 2)) * g.ldb) * 8);
#pragma unroll
        for (int i = 0; i < NLA; ++i) ra[i] = *reinterpret_cast<const v2d*>(pa + i * ia + ta);
#pragma unroll
        for (int i = 0; i < NLB; ++i) rb[i] = *reinterpret_cast<const v2d*>(pb + i * ib + tb);
        // a K-contiguous thread holds the same two k for all its loads: one 16-B scale load per tile
        const bool scaled = AKC && g.kscale != nullptr;
        const double* ps = scaled ? g.kscale + k00 + 2 * (tid & 7) : nullptr;
        v2d rs = (v2d){1.0, 1.0};
        if (scaled) rs = *reinterpret_cast<const v2d*>(ps);
        for (int kt = kt0; kt < kt1; ++kt) {
            if (scaled) {
#pragma unroll
                for (int i = 0; i < NLA; ++i) ra[i] = ra[i] * rs;
            }
            __syncthreads();
            sstore<BM, AKC, NT>(sA, ra);
            sstore<BN, BKC, NT>(sB, rb);
            __syncthreads();
            if (kt + 1 < kt1) {
                pa += ka;
                pb += kb;
#pragma unroll
                for (int i = 0; i < NLA; ++i) ra[i] = *reinterpret_cast<const v2d*>(pa + i * ia + ta);
#pragma unroll
                for (int i = 0; i < NLB; ++i) rb[i] = *reinterpret_cast<const v2d*>(pb + i * ib + tb);
                if (scaled) {
                    ps += BK;
                    rs = *reinterpret_cast<const v2d*>(ps);
                }
            }
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = sfrag<BM, AKC>(sA, wm + 16 * i + l15, kk * 4 + l4);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = sfrag<BN, BKC>(sB, wn + sub16<TN>(j) + l15, kk * 4 + l4);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[j][i], 0, 0, 0);
            }
        }
    } else {
        const bool scaled = AKC && g.kscale != nullptr;
        auto scale_tile = [&](int k0) {
            const int k = k0 + 2 * (tid & 7);
            const v2d rs = (v2d){k < g.K ? g.kscale[k] : 0.0, k + 1 < g.K ? g.kscale[k + 1] : 0.0};
#pragma unroll
            for (int i = 0; i < NLA; ++i) ra[i] = ra[i] * rs;
        };
        if (kt0 < kt1) {
            const int k0 = klo + kt0 * BK;
            const bool kin = k0 + BK <= g.K;
            gload<BM, AKC, NT>(gA, g.lda, m0, k0, g.M, g.K, tfA && kin, ra);
            gload<BN, BKC, NT>(gB, g.ldb, n0, k0, g.N, g.K, tfB && kin, rb);
            if (scaled) scale_tile(k0);
        }
        for (int kt = kt0; kt < kt1; ++kt) {
            __syncthreads();
            sstore<BM, AKC, NT>(sA, ra);
            sstore<BN, BKC, NT>(sB, rb);
            __syncthreads();
            if (kt + 1 < kt1) {
                const int k0 = klo + (kt + 1) * BK;
                const bool kin = k0 + BK <= g.K;
                gload<BM, AKC, NT>(gA, g.lda, m0, k0, g.M, g.K, tfA && kin, ra);
                gload<BN, BKC, NT>(gB, g.ldb, n0, k0, g.N, g.K, tfB && kin, rb);
                if (scaled) scale_tile(k0);
            }
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = sfrag<BM, AKC>(sA, wm + 16 * i + l15, kk * 4 + l4);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = sfrag<BN, BKC>(sB, wn + sub16<TN>(j) + l15, kk * 4 + l4);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[j][i], 0, 0, 0);
            }
        }
    }

    // epilogue: lane holds C[m = .. + (lane&15)][n = .. + (lane>>4) + 4r]
    double* Cb = gC + (g.splitk > 1 ? (long)ks * g.slab : 0);
    const bool plain = RD != 0 || g.splitk > 1 || g.beta == 0.0;   // the row-dot variants are beta == 0 only: no old-value path in them
    if (RD == 3 && bj == 0) {
        if (dot_done) {   // the two waves of a row half hold the same sums: one of them writes
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                double a = sv[i];
                a += __shfl_xor(a, 16, 64);
                a += __shfl_xor(a, 32, 64);
                if (l4 == 0 && (wave >> 1) == 0) g.R2[m0 + wm + 16 * i + l15] = a;
            }
        } else {   // edge tiles and launches outside the direct-to-LDS loop: one row per thread, straight from memory
            const int m = m0 + tid;
            if (tid < BM && m < g.M) {
                double a = 0.0;
                for (int k = 0; k < g.K; ++k) a = fma(AKC ? gA[k + (long)m * g.lda] : gA[m + (long)k * g.lda], g.evec[k], a);
                g.R2[m] = a;
            }
        }
    }
    const int part = 2 * bj + (wave >> 1);   // row-dot parts: two per tile column (one per wave column)
    if (RD == 2 && m0 + BM <= g.M && n0 + BN <= g.N) {
        store_rowdot_full<TM, TN>(g, gE, acc, Cb, m0 + wm, n0 + wn, part, l15, l4);
        return;
    }
    if (m0 + BM <= g.M && n0 + BN <= g.N) {
        // full tile: no bounds tests; with beta != 0 the old values come in batches of TM*4 loads per
        // wait instead of one round trip per element
        double* c0 = Cb + (m0 + wm + l15) + (long)(n0 + wn + l4) * g.ldc;
        const long s4 = 4 * g.ldc;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            double* cj = c0 + (long)sub16<TN>(j) * g.ldc;
            if (plain) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < TM; ++i) cj[r * s4 + 16 * i] = g.alpha * acc[j][i][r];
            } else {
                double old[4][TM];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < TM; ++i) old[r][i] = cj[r * s4 + 16 * i];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < TM; ++i) cj[r * s4 + 16 * i] = g.alpha * acc[j][i][r] + g.beta * old[r][i];
            }
        }
        if (RD == 1 || RD == 2) rowdot_epilogue<TM, TN, true, RD>(g, gE, acc, m0 + wm, n0 + wn, part, l15, l4);
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm + 16 * i + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn + sub16<TN>(j) + l4 + 4 * r;
                if (m < g.M && n < g.N) {
                    double* c = Cb + m + (long)n * g.ldc;
                    const double v = g.alpha * acc[j][i][r];
                    *c = plain ? v : v + g.beta * *c;
                }
            }
        }
    if (RD == 1 || RD == 2) rowdot_epilogue<TM, TN, false, RD>(g, gE, acc, m0 + wm, n0 + wn, part, l15, l4);   // a part without columns gets zeros
}

template <int TM, int TN, bool AKC, bool BKC, int RD = 0, int SK = 0>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs g) {
    gemm_body<TM, TN, AKC, BKC, RD, SK>(g, (int)blockIdx.x);
}

// TWO independent small products in ONE launch: workgroups [0, grid0) run the first (M/N-contiguous operands), the rest the
// second (B K-contiguous).  The node of the factor-and-invert recursion uses it for A22 -= L21 L21^T and T = L21 W11, which
// both wait for L21 only: below 4096 rows such products are a few workgroups bound by the launch-to-launch step, not by
// the matrix pipe, so two of them side by side cost what one does (chol.hip potrf_inv_rec).
template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm_pair_kernel(const GemmArgs g0, const GemmArgs g1, const int grid0) {
    if ((int)blockIdx.x < grid0) gemm_body<TM, TN, false, false>(g0, (int)blockIdx.x);
    else gemm_body<TM, TN, false, true>(g1, (int)blockIdx.x - grid0);
}

__global__ void splitk_reduce_kernel(double* __restrict__ C, long ldc, const double* __restrict__ W,
                                     long slab, int splitk, int M, int N, double beta, int lower_bt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * N) return;
    const int m = (int)(idx % M), n = (int)(idx / M);
    if (lower_bt && (n / lower_bt) > (m / lower_bt)) return;  // tiles above the diagonal were never written
    double s = 0.0;
    for (int k = 0; k < splitk; ++k) s += W[(long)k * slab + idx];
    double* c = C + m + (long)n * ldc;
    *c = beta == 0.0 ? s : s + beta * *c;
}

template <int TM, int TN, int SK = 0>
void launch(hipStream_t st, int a_kc, int b_kc, const GemmArgs& g, int grid) {
    const dim3 blk(256);
    if (!a_kc && !b_kc) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, false, 0, SK>), dim3(grid), blk, 0, st, g);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, true, 0, SK>), dim3(grid), blk, 0, st, g);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_kernel<TM, TN, true, false, 0, SK>), dim3(grid), blk, 0, st, g);
    else hipLaunchKernelGGL((gemm_kernel<TM, TN, true, true, 0, SK>), dim3(grid), blk, 0, st, g);
}

// which half-tile skipping instantiation of the 128-tile kernel a product takes (see the kernel's direct-to-LDS loop)
int half_skip_kind(int flags) {
    static const bool off = getenv("GPX_NO_HALF_SKIP") != nullptr;   // measurement only
    if (off || (flags & GEMM_NO_HALF_SKIP)) return 0;
    const int kf = flags & (GEMM_KLO_M | GEMM_KLO_N | GEMM_KHI_M | GEMM_KHI_N);
    if (flags & GEMM_LOWER) return 0;
    return kf == GEMM_KHI_N ? 1 : kf == GEMM_KLO_N ? 2 : 0;
}

// ---- optional per-launch HIP-event timing of the large GEMMs (bench.py's roofline leg) -------
struct GemmProf {
    std::mutex mu;  // contexts on different host threads launch GEMMs concurrently
    std::atomic<bool> on{false};   // read outside the lock by every launch
    int dev = -1;   // the device that was current at gpx_gemm_prof_begin: only its launches are timed (the events live there)
    bool only128 = false;  // time the 128x128-tile launches only (the kernel the roofline object names)
    double min_flops = 0;
    std::vector<hipEvent_t> ev;  // pairs
    std::vector<double> flops;
    std::vector<std::array<long, 6>> shape;
    size_t used = 0;
    double skipped_flops = 0;
    long skipped = 0;
} g_prof;

double useful_flops(long M, long N, long K, int flags) {
    double f = 2.0 * (double)M * (double)N * (double)K;
    if (flags & GEMM_LOWER) f *= 0.5 * (1.0 + 1.0 / (double)((M + 127) / 128));
    if (flags & (GEMM_KLO_M | GEMM_KLO_N | GEMM_KHI_M | GEMM_KHI_N)) {
        // triangular operand: half of the K range on average; W^T W on lower tiles sums over
        // k >= max(m, n): a third of the lower half (n^3/3 in all)
        f *= (flags & GEMM_LOWER) ? 1.0 / 3.0 : 0.5;
    }
    return f;
}

struct GemmDeferred {   // a product gemm_impl has planned but not launched
    GemmArgs g;
    int bt, grid, a_kc, b_kc;
    bool ready;
};

int gemm_impl(hipStream_t st, int a_kc, int b_kc, long M, long N, long K, double alpha,
              const double* A, long lda, const double* B, long ldb, double beta, double* C, long ldc,
              int flags, int splitk, double* work, const double* kscale = nullptr, const GemmRowdot* rd = nullptr,
              const GemmPanels* pn = nullptr, GemmDeferred* defer = nullptr) {
    if (defer) defer->ready = false;
    if (M <= 0 || N <= 0) return GPX_OK;
    int kper = 0;
    if (pn) {
        for (long v : {pn->panA, pn->panB, pn->panC, pn->panE})
            if (v < 0 || v % 128) return GPX_ERR_ARG;   // a tile never straddles two panels
        const long kpan = (a_kc && pn->panA) ? pn->panA : ((b_kc && pn->panB) ? pn->panB : 0);
        if (kpan) {
            // K-panelled operand(s): one panel per split-K slab, full K range of every tile
            if ((a_kc && b_kc && pn->panA != pn->panB) || (flags & (GEMM_KLO_M | GEMM_KLO_N | GEMM_KHI_M | GEMM_KHI_N))) return GPX_ERR_ARG;
            kper = (int)(kpan / BK);
            const long need = (K + kpan - 1) / kpan;
            if (need > 1 && splitk != need) return GPX_ERR_ARG;   // gpx_gemm_splitk derives the slab count from the panels
        }
    }
    const bool opdot = rd && !rd->R1 && !rd->E && rd->evec && rd->R2;   // R2 = A evec from the operand fragments
    if (opdot) {
        if (a_kc || b_kc || beta != 0.0 || splitk > 1 || (flags & (GEMM_LOWER | GEMM_TILE64 | GEMM_KLO_M | GEMM_KHI_M | GEMM_KLO_N | GEMM_KHI_N))) return GPX_ERR_ARG;
        if (reinterpret_cast<uintptr_t>(rd->evec) & 3) return GPX_ERR_ARG;
        flags |= GEMM_TILE128;
    } else if (rd && !rd->R1) {
        return GPX_ERR_ARG;
    }
    if (rd && rd->R1) {
        if (a_kc || b_kc || beta != 0.0 || splitk > 1 || (flags & (GEMM_LOWER | GEMM_TILE64)) || (rd->E && !(rd->evec && rd->R2)) || (!rd->E && (rd->evec || rd->R2))) return GPX_ERR_ARG;
        flags |= GEMM_TILE128;   // the parts are the columns one wave of a 128-tile owns
    }
    if (kscale && (!a_kc || (reinterpret_cast<uintptr_t>(kscale) & 15))) return GPX_ERR_ARG;
    if (M > 0x7fffffffL - 256 || N > 0x7fffffffL - 256 || K > 0x7fffffffL - 256) return GPX_ERR_ARG;
    if ((flags & GEMM_LOWER) && M != N) return GPX_ERR_ARG;
    // K ranges that grow with the tile index: walk the tiles backwards so the long ones start first
    if (!(flags & GEMM_LOWER) && (flags & (GEMM_KHI_M | GEMM_KHI_N))) flags |= GEMM_HEAVY_LAST;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    g.M = (int)M; g.N = (int)N; g.K = (int)K;
    g.alpha = alpha; g.beta = beta;
    g.kscale = kscale;
    g.E = nullptr; g.lde = 0; g.evec = nullptr; g.R1 = g.R2 = nullptr; g.ldr = 0;
    if (rd && (rd->R1 || opdot)) {
        g.E = rd->E; g.lde = rd->lde; g.evec = rd->evec; g.R1 = rd->R1; g.R2 = rd->R2; g.ldr = rd->ldr;
    }
    g.panA = pn ? pn->panA : 0; g.pstA = pn ? pn->pstA : 0;
    g.panB = pn ? pn->panB : 0; g.pstB = pn ? pn->pstB : 0;
    g.panC = pn ? pn->panC : 0; g.pstC = pn ? pn->pstC : 0;
    g.panE = pn ? pn->panE : 0; g.pstE = pn ? pn->pstE : 0;
    g.kper = kper;
    g.fastA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0) && (lda % 2 == 0) && (!pn || pn->pstA % 2 == 0);
    g.fastB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (ldb % 2 == 0) && (!pn || pn->pstB % 2 == 0);
    const TilePlan plan = plan_tiles(M, N, flags, splitk, kscale != nullptr);
    static_cast<TileOrder&>(g) = plan.main;
    if (!gemm_lane_offsets_fit_32bit(lda, a_kc != 0) || !gemm_lane_offsets_fit_32bit(ldb, b_kc != 0)) g.flags |= GEMM_NO_FAST32;
    if (g.E && !gemm_lane_offsets_fit_32bit(g.lde, true)) g.flags |= GEMM_NO_FAST32;   // the E warm-up offsets: 31 columns of lde
    const int bt = plan.bt;
    const bool band = plan.band;
    const long grid = plan.grid;
    g.slab = M * N;
    if (g.splitk > 1) {
        g.C = work;
        g.ldc = M;
    }
    if (grid > 0x7fffffffL) return GPX_ERR_ARG;
    const double fl = useful_flops(M, N, K, flags);
    bool timed = false;
    size_t slot = 0;
    if (g_prof.on.load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (g_prof.on && cur == g_prof.dev && fl >= g_prof.min_flops && (!g_prof.only128 || bt == 128) && g_prof.used + 2 <= g_prof.ev.size()) {
            timed = true;
            slot = g_prof.used;
            g_prof.used += 2;
            g_prof.flops.push_back(fl);
            g_prof.shape.push_back({M, N, K, (long)a_kc, (long)b_kc, (long)flags});
            (void)hipEventRecord(g_prof.ev[slot], st);
        } else {
            g_prof.skipped_flops += fl;
            g_prof.skipped++;
        }
    }
    if (defer && !timed && !band && g.splitk == 1 && !g.R1 && !opdot && !kscale && bt <= 64) {
        // a plain small product: the caller launches it together with another one (gemm_pair_kernel)
        defer->g = g;
        defer->bt = bt;
        defer->grid = (int)grid;
        defer->a_kc = a_kc;
        defer->b_kc = b_kc;
        defer->ready = true;
        return GPX_OK;
    }
    const int sk = bt == 128 ? half_skip_kind(flags) : 0;
    if (opdot) hipLaunchKernelGGL((gemm_kernel<4, 4, false, false, 3>), dim3((unsigned)grid), dim3(256), 0, st, g);
    else if (g.R1 && g.E) hipLaunchKernelGGL((gemm_kernel<4, 4, false, false, 2>), dim3((unsigned)grid), dim3(256), 0, st, g);
    else if (g.R1 && sk == 1) hipLaunchKernelGGL((gemm_kernel<4, 4, false, false, 1, 1>), dim3((unsigned)grid), dim3(256), 0, st, g);
    else if (g.R1) hipLaunchKernelGGL((gemm_kernel<4, 4, false, false, 1>), dim3((unsigned)grid), dim3(256), 0, st, g);
    else if (bt == 128 && sk == 1) launch<4, 4, 1>(st, a_kc, b_kc, g, (int)grid);
    else if (bt == 128 && sk == 2) launch<4, 4, 2>(st, a_kc, b_kc, g, (int)grid);
    else if (bt == 128) launch<4, 4>(st, a_kc, b_kc, g, (int)grid);
    else if (bt == 64) launch<2, 2>(st, a_kc, b_kc, g, (int)grid);
    else launch<1, 1>(st, a_kc, b_kc, g, (int)grid);
    if (band) {
        GemmArgs b = g;
        static_cast<TileOrder&>(b) = plan.band_order;
        b.flags |= g.flags & GEMM_NO_FAST32;
        if (plan.band_bt == 32) launch<1, 1>(st, a_kc, b_kc, b, (int)plan.band_grid);
        else launch<2, 2>(st, a_kc, b_kc, b, (int)plan.band_grid);
    }
    if (g.splitk > 1) {
        const long tot = M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, C,
                           ldc, work, g.slab, g.splitk, (int)M, (int)N, beta,
                           (flags & GEMM_LOWER) ? bt : 0);
    }
    if (timed) (void)hipEventRecord(g_prof.ev[slot + 1], st);
    HIP_TRY(hipGetLastError());
    return GPX_OK;
}

// bare MFMA loop: what the matrix pipe delivers with operands in registers
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(double* out, int iters) {
    // the GEMM's own register picture: 16 independent 16x16 accumulators, 4 + 4 operand values
    v4d acc[4][4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) acc[j][i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = 1.0 + (threadIdx.x + i) * 1e-3;
        b[i] = 1.0 - (threadIdx.x + i) * 1e-3;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j], a[i], acc[j][i], 0, 0, 0);
    }
    double s = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) s += acc[j][i][0] + acc[j][i][1] + acc[j][i][2] + acc[j][i][3];
    out[(long)blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace

int gpx_gemm_prof_begin(double min_flops, int max_launches) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    HIP_TRY(hipGetDevice(&g_prof.dev));
    g_prof.only128 = min_flops < 0;  // negative threshold: "every 128x128-tile launch, nothing else"
    if (min_flops < 0) min_flops = 0;
    if (g_prof.ev.size() < (size_t)max_launches * 2) {
        const size_t old = g_prof.ev.size();
        g_prof.ev.resize((size_t)max_launches * 2);
        for (size_t i = old; i < g_prof.ev.size(); ++i) HIP_TRY(hipEventCreate(&g_prof.ev[i]));
    }
    g_prof.on = true;
    g_prof.min_flops = min_flops;
    g_prof.used = 0;
    g_prof.flops.clear();
    g_prof.shape.clear();
    g_prof.skipped_flops = 0;
    g_prof.skipped = 0;
    return GPX_OK;
}

int gpx_gemm_prof_end(double* total_ms, double* total_flops, long* launches, double* skipped_flops) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = false;
    double ms = 0, fl = 0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        HIP_TRY(hipEventSynchronize(g_prof.ev[i + 1]));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]));
        ms += t;
        fl += g_prof.flops[i / 2];
        if (getenv("GPX_GEMM_LOG")) {  // per-launch shapes and rates of the timed evaluation(s), to stderr
            const auto& q = g_prof.shape[i / 2];
            fprintf(stderr, "gpx gemm M=%ld N=%ld K=%ld akc=%ld bkc=%ld flags=%ld  %.3f ms  %.1f TFLOP/s\n", q[0], q[1], q[2],
                    q[3], q[4], q[5], t, g_prof.flops[i / 2] / (t * 1e-3) / 1e12);
        }
    }
    *total_ms = ms;
    *total_flops = fl;
    *launches = (long)(g_prof.used / 2);
    *skipped_flops = g_prof.skipped_flops;
    return GPX_OK;
}

int gpx_mfma_peak(double* tflops) {
    // 2 waves per SIMD and 16 independent accumulators per wave, as in the GEMM
    const int blocks = 256 * 2, iters = 8192;
    double* out = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&out), (size_t)blocks * 256 * 8));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, 100);
    HIP_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *tflops = (double)blocks * 4 * iters * 16 * 2048.0 / (ms * 1e-3) / 1e12;
    (void)hipFree(out);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return GPX_OK;
}

int gpx_gemm(hipStream_t st, int a_kc, int b_kc, long M, long N, long K, double alpha,
             const double* A, long lda, const double* B, long ldb, double beta, double* C, long ldc,
             int flags, const GemmPanels* pn) {
    return gemm_impl(st, a_kc, b_kc, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, flags, 1, nullptr, nullptr, nullptr, pn);
}

// C0 = alpha0 A0 B0^T + beta0 C0 (all operands M/N-contiguous) and C1 = alpha1 A1 op(B1)^T + beta1 C1 (B1 K-contiguous), two
// independent products.  One launch when both are plain small products on the same tile size, two launches otherwise.
int gpx_gemm_pair(hipStream_t st, long M0, long N0, long K0, double alpha0, const double* A0, long lda0, const double* B0, long ldb0,
                  double beta0, double* C0, long ldc0, int flags0, long M1, long N1, long K1, double alpha1, const double* A1,
                  long lda1, const double* B1, long ldb1, double beta1, double* C1, long ldc1, int flags1) {
    static const bool off = getenv("GPX_NO_GEMM_PAIR") != nullptr;   // A/B switch
    GemmDeferred d0, d1;
    d0.ready = d1.ready = false;
    int rc = off ? GPX_OK : gemm_impl(st, 0, 0, M0, N0, K0, alpha0, A0, lda0, B0, ldb0, beta0, C0, ldc0, flags0, 1, nullptr, nullptr, nullptr, nullptr, &d0);
    if (rc) return rc;
    if (d0.ready) {
        rc = gemm_impl(st, 0, 1, M1, N1, K1, alpha1, A1, lda1, B1, ldb1, beta1, C1, ldc1, flags1, 1, nullptr, nullptr, nullptr, nullptr, &d1);
        if (rc) return rc;
        static const long max_grid = [] { const char* e = getenv("GPX_PAIR_MAX_GRID"); return e ? atol(e) : 2048L; }();   // beyond that the halved occupancy of the pair (46 KB of LDS) costs more than a launch (exact n = 8192: 13.30 ms unpaired, 12.85 with the limit at 2048, 12.95 without limit)
        if (d1.ready && d1.bt == d0.bt && (long)d0.grid + d1.grid <= max_grid) {
            const unsigned grid = (unsigned)(d0.grid + d1.grid);
            if (d0.bt == 32) hipLaunchKernelGGL((gemm_pair_kernel<1, 1>), dim3(grid), dim3(256), 0, st, d0.g, d1.g, d0.grid);
            else hipLaunchKernelGGL((gemm_pair_kernel<2, 2>), dim3(grid), dim3(256), 0, st, d0.g, d1.g, d0.grid);
            HIP_TRY(hipGetLastError());
            return GPX_OK;
        }
        // not a pair after all: the first product was only planned, the second launched itself unless it was deferred too
        rc = gemm_impl(st, 0, 0, M0, N0, K0, alpha0, A0, lda0, B0, ldb0, beta0, C0, ldc0, flags0, 1, nullptr);
        if (rc) return rc;
        if (!d1.ready) return GPX_OK;
        return gemm_impl(st, 0, 1, M1, N1, K1, alpha1, A1, lda1, B1, ldb1, beta1, C1, ldc1, flags1, 1, nullptr);
    }
    if (off || M0 <= 0 || N0 <= 0) {   // (an empty first product returns without planning anything)
        rc = gemm_impl(st, 0, 0, M0, N0, K0, alpha0, A0, lda0, B0, ldb0, beta0, C0, ldc0, flags0, 1, nullptr);
        if (rc) return rc;
    }
    // the first product launched itself inside its planning call (too large, timed, banded ...): the second follows alone
    return gemm_impl(st, 0, 1, M1, N1, K1, alpha1, A1, lda1, B1, ldb1, beta1, C1, ldc1, flags1, 1, nullptr);
}

int gpx_gemm_rowdot(hipStream_t st, int a_kc, int b_kc, long M, long N, long K, double alpha, const double* A, long lda,
                    const double* B, long ldb, double* C, long ldc, int flags, const GemmRowdot& rd, const GemmPanels* pn) {
    return gemm_impl(st, a_kc, b_kc, M, N, K, alpha, A, lda, B, ldb, 0.0, C, ldc, flags, 1, nullptr, nullptr, &rd, pn);
}

// slab count and tile size of a split-K product (flags gains GEMM_TILE128 where 128-tiles pay)
static int choose_splitk(long M, long N, long K, int& flags, long work_doubles);

// rows per panel of the K-contiguous operands of a split-K Gram product over K = n rows: the K range of one slab of the
// slab count the product would take anyway, rounded up to a multiple of 128 rows
long gpx_gemm_splitk_panel_rows(long M, long N, long K, int flags, long work_doubles) {
    int f = flags;
    const int sk = choose_splitk(M, N, K, f, work_doubles);
    const long kt = (K + BK - 1) / BK, per = (kt + sk - 1) / sk;
    return (per + 7) / 8 * 128;
}

int gpx_gemm_splitk(hipStream_t st, int a_kc, int b_kc, long M, long N, long K, double alpha,
                    const double* A, long lda, const double* B, long ldb, double beta, double* C,
                    long ldc, int flags, double* work, long work_doubles, const double* kscale, const GemmPanels* pn) {
    int splitk = work ? choose_splitk(M, N, K, flags, work_doubles) : 1;
    const long kpan = pn ? ((a_kc && pn->panA) ? pn->panA : ((b_kc && pn->panB) ? pn->panB : 0)) : 0;
    if (kpan) {   // K-panelled operands: one panel per slab
        splitk = (int)((K + kpan - 1) / kpan);
        if (splitk > 1 && (!work || (long)splitk * M * N > work_doubles)) return GPX_ERR_ARG;
    }
    return gemm_impl(st, a_kc, b_kc, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, flags, splitk, work, kscale, nullptr, pn);
}

static int choose_splitk(long M, long N, long K, int& flags, long work_doubles) {
    int splitk = 1;
    {
        const long kt = (K + BK - 1) / BK;
        // 128x128 tiles (twice the flops per operand byte) whenever enough slabs fit the workspace to
        // fill the chip with them; 64x64 tiles otherwise
        const long t128 = ((M + 127) / 128) * ((N + 127) / 128) / ((flags & GEMM_LOWER) ? 2 : 1);
        int s128 = 1;
        while (s128 < 64 && t128 * s128 < 2048 && kt / (s128 * 2) >= 32 && (long)(s128 * 2) * M * N <= work_doubles)
            s128 *= 2;
        if (!(flags & (GEMM_TILE64 | GEMM_TILE128)) && t128 * s128 >= 1024) {
            flags |= GEMM_TILE128;
            splitk = s128;
            // 512 workgroups run at a time (2 per CU): take the slab count whose tiles x slabs fill whole
            // rounds best (m = 1024 LOWER: 36 tiles x 64 = 4.5 rounds, x 56 = 3.94).  A multiple of 8 lets
            // every slab live on ONE XCD (see kernel), which is worth more than the last percent of a
            // round: m = 2048: 136 x 15 = 3.98 rounds, but 136 x 64 = 17.0 with XCD-private slabs.
            const long tm = (M + 127) / 128, tn = (N + 127) / 128;
            const long ta = (flags & GEMM_LOWER) ? tm * (tm + 1) / 2 : tm * tn;
            int smax = s128;
            while (smax + 8 <= 64 && kt / (smax + 8) >= 32 && (long)(smax + 8) * M * N <= work_doubles) smax += 8;
            double best = 0;
            for (int sk = smax; sk > s128 / 2 && sk >= 1; --sk) {
                if (sk > s128 && sk % 8) continue;
                const long w = ta * sk;
                double score = (double)w / (double)((w + 511) / 512 * 512) - 0.0005 * sk;
                if (sk % 8 == 0) score += 0.05;
                if (score > best + 1e-9) best = score, splitk = sk;
            }
        } else {
            const long tiles = ((M + 63) / 64) * ((N + 63) / 64) / ((flags & GEMM_LOWER) ? 2 : 1);
            while (splitk < 64 && tiles * splitk < 1024 && kt / (splitk * 2) >= 16 &&
                   (long)(splitk * 2) * M * N <= work_doubles)
                splitk *= 2;
        }
    }
    return splitk;
}
