// tile_order.h — which output tile (and split-K slab) a workgroup of gemm_kernel computes, and how the host
// sizes the launch.  Plain C++ on purpose: the kernel and gemm_impl use it, and tests/cpp/tile_order_check.cpp
// enumerates every block of many launch plans on the CPU to check that each tile is computed exactly once.
#pragma once
#include <cstdlib>
#include <cmath>
#include "gemm_flags.h"

#ifdef __HIPCC__
#define GPX_HD __host__ __device__
#else
#define GPX_HD
#endif

struct TileOrder {
    int flags;
    int tiles_m, tiles_n, ntiles;
    int splitk;
    int xcd_rows;            // tall-skinny outputs: the N tiles of one M tile run on ONE XCD
    int sq_m, sq_n, nsuper;  // 8x8-tile squares, one per XCD at a time (nsuper == 0: plain tile order)
};

// block -> (bi, bj) tile indices, split-K slab ks, K walk direction.  false: the block has no tile.
GPX_HD inline bool tile_of_block(const TileOrder& g, int block, int& bi, int& bj, int& ks, bool& desc) {
    int tile = block;
    ks = 0;
    if (g.splitk > 1) {
        // slab-major: the workgroups that run together are the tiles of ONE K slab, walking it in step,
        // so the operand rows they share come out of L2 (tile-major order re-read every operand tile
        // from HBM once per output tile: 1.2 TB for the VFE Gram product instead of 69 GB)
        if (g.splitk % 8 == 0) {
            // workgroup b runs on XCD b % 8: give every XCD its own slabs, so that the operand rows of a
            // slab enter ONE L2 (the m = 1024 Gram product fetched 65 GB for 8.6 GB of operands when the
            // tiles of each slab were dealt round the eight XCDs)
            const int l = tile >> 3;
            ks = (l / g.ntiles) * 8 + (tile & 7);
            tile = l % g.ntiles;
        } else {
            ks = tile / g.ntiles;
            tile -= ks * g.ntiles;
        }
    }
    desc = false;  // walk the K range downwards (direct-to-LDS loop only; any order is correct)
    if (g.flags & GEMM_DIAG_BAND32) {
        // the diagonal 128x128 blocks of a LOWER product, ten 32x32 tiles each (rows 0..3 of the block's lower triangle)
        const int d = tile / 10, q = tile - 10 * d;
        const int r = (q >= 1) + (q >= 3) + (q >= 6);
        bi = 4 * d + r;
        bj = 4 * d + (q - r * (r + 1) / 2);
        if (bi >= g.tiles_m) return false;
    } else if (g.flags & GEMM_DIAG_BAND) {
        // the diagonal 128x128 blocks of a LOWER product, three 64x64 tiles each (see gemm_impl)
        const int d = tile / 3, q = tile - 3 * d;
        bi = 2 * d + (q > 0);
        bj = 2 * d + (q > 1);
        if (bi >= g.tiles_m) return false;
    } else if (g.xcd_rows) {
        // Workgroup b runs on XCD b % 8.  For a tall-skinny output (few N tiles) give each XCD its own
        // M tiles and let that M tile's N tiles follow one another on it: the A rows are then fetched
        // into ONE L2 once instead of once per N tile from HBM / Infinity Cache.
        const int j = (tile >> 3) % g.tiles_n;
        bi = ((tile >> 3) / g.tiles_n) * 8 + (tile & 7);
        bj = j;
        if (bi >= g.tiles_m) return false;
    } else if (g.nsuper) {
        // Workgroup b runs on XCD b % 8 and an XCD holds 64 workgroups (128-tiles, 2 per CU) at a time: let
        // those 64 be ONE 8x8 square of tiles that marches through K in step, so that each XCD's L2 fetches
        // 8 + 8 operand tiles per K step for 64 consumers.  In plain tile order the sharers of an operand
        // tile run 32 positions apart on their XCD, out of phase: at M = N = 32768 every A tile then came
        // from HBM once per output tile (2.4 TB/s of L2 misses, 71 instead of 76 TFLOP/s).
        //
        // Triangular K ranges (one of KLO_* / KHI_*): the eight rows (or columns) of a square differ in
        // trip count by 8 K tiles each, and a slot freed early is refilled early, so squares that all walk
        // K the same way drift apart.  Squares therefore ALTERNATE on each XCD between
        //   S: all 64 start together, walk K from the end of the range the tiles have in common (phases
        //      equal), finish shortest row first, and
        //   F: dispatched longest row first into the slots as they free up, walk K TOWARDS the common end:
        //      a row that is 8 K tiles shorter starts 8 K tiles later, the phases are equal again and
        //      all 64 finish together, so the next S square starts together.
        const int w = (tile >> 3) & 63, sq = tile >> 9;
        int sg = sq * 8 + (tile & 7);
        if (sg >= g.nsuper) {
            // the 36-tile squares on the diagonal of a LOWER product come last, packed densely
            if (!(g.flags & GEMM_LOWER)) return false;
            const int t = tile - ((g.nsuper + 7) / 8) * 512;
            if (t < 0) return false;
            const bool strict = g.flags & GEMM_STRICT_LOWER;  // the diagonal tiles run in a launch of their own
            const int per = strict ? 28 : 36;
            const int d = t / per, u = t % per;
            int r = 0;
            while ((r + 1) * (r + 2) / 2 <= u) ++r;
            bi = d * 8 + r + (strict ? 1 : 0);
            bj = d * 8 + (u - r * (r + 1) / 2);
            if (bi >= g.tiles_m) return false;
        } else {
            if (g.flags & GEMM_HEAVY_LAST) sg = g.nsuper - 1 - sg;
            const bool depM = g.flags & (GEMM_KLO_M | GEMM_KHI_M), depN = g.flags & (GEMM_KLO_N | GEMM_KHI_N);
            int sm, sn;
            if (g.flags & GEMM_LOWER) {  // squares strictly below the diagonal, row-major
                sm = (int)((sqrt(8.0 * (double)sg + 1.0) - 1.0) * 0.5);
                while (sm * (sm + 1) / 2 > sg) --sm;
                while ((sm + 1) * (sm + 2) / 2 <= sg) ++sm;
                sn = sg - sm * (sm + 1) / 2;
                sm += 1;
            } else if (depM) {  // the 8 XCDs work on squares of equal trip counts
                sm = sg / g.sq_n;
                sn = sg % g.sq_n;
            } else {
                sm = sg % g.sq_m;
                sn = sg / g.sq_m;
            }
            // major index = the one the trip count depends on, longest first
            const bool khi = g.flags & (GEMM_KHI_M | GEMM_KHI_N);
            const int maj = khi ? 7 - (w >> 3) : (w >> 3), mnr = w & 7;
            bi = sm * 8 + (depM ? maj : mnr);
            bj = sn * 8 + (depM ? mnr : maj);
            if (bi >= g.tiles_m || bj >= g.tiles_n) return false;
            if (depM || depN) desc = khi ? (sq & 1) : !(sq & 1);
        }
    } else {
        if (g.flags & GEMM_HEAVY_LAST) tile = g.ntiles - 1 - tile;
        if (g.flags & GEMM_LOWER) {
            bi = (int)((sqrt(8.0 * (double)tile + 1.0) - 1.0) * 0.5);
            while (bi * (bi + 1) / 2 > tile) --bi;
            while ((bi + 1) * (bi + 2) / 2 <= tile) ++bi;
            bj = tile - bi * (bi + 1) / 2;
            if (g.flags & GEMM_STRICT_LOWER) bi += 1;
        } else if (g.flags & (GEMM_KLO_M | GEMM_KHI_M)) {
            // K range depends on the M tile: neighbouring workgroups share it (same trip count, same A
            // rows in L2, marching through K together) when the N index runs fastest
            bi = tile / g.tiles_n;
            bj = tile % g.tiles_n;
        } else {
            bi = tile % g.tiles_m;
            bj = tile / g.tiles_m;
        }
    }
    return true;
}

// The interior K loops of gemm_kernel address an operand tile as a 64-bit scalar base plus ONE 32-bit byte offset
// per lane: up to 31 rows of ld doubles for a K-contiguous operand (127 rows for its L2 warm-up load), up to 7
// k-rows plus 128 B for an M/N-contiguous one.  False: the offsets would wrap, the launch must take the generic
// 64-bit path (GEMM_NO_FAST32).
inline bool gemm_lane_offsets_fit_32bit(long ld, bool k_contiguous) {
    const double rows = k_contiguous ? 127.0 : 7.0;
    return (rows * (double)ld + 128.0) * 8.0 < 4294967296.0;
}

struct TilePlan {
    TileOrder main;
    long grid;
    int bt;              // 128, 64 or 32
    bool band;           // LOWER product: second launch over the diagonal blocks (64-tiles, or 32-tiles: band_bt)
    TileOrder band_order;
    long band_grid;
    int band_bt;
};

// Launch plan of C[M x N] with option bits `flags`, `splitk` K slabs (>= 1), optional K-scale operand.
inline TilePlan plan_tiles(long M, long N, int flags, int splitk, bool has_kscale) {
    TilePlan p{};
    TileOrder& g = p.main;
    g.flags = flags;
    g.splitk = splitk > 1 ? splitk : 1;
    // tile choice: 128x128 when that still fills the chip, 64x64 otherwise
    int bt = 128;
    {
        const long t128 = ((M + 127) / 128) * ((N + 127) / 128) / ((flags & GEMM_LOWER) ? 2 : 1);
        if (t128 * (splitk > 1 ? splitk : 1) < 384) bt = 64;
        // a 64x64 tile is bound by the matrix pipe of the ONE CU it runs on (0.43 us per 16-deep K step) and the longest
        // tile of a triangular product bounds the launch: products with at most 1024 such tiles (the chains of 2048 rows
        // and less) take 32x32 tiles, four times the workgroups per product (exact n = 8192 on one box: 14.67 ms with the
        // limit at 64 tiles, 14.34 at 256, 14.07 at 1024)
        const long t64 = ((M + 63) / 64) * ((N + 63) / 64) / ((flags & GEMM_LOWER) ? 2 : 1);
        static const long t32max = [] { const char* e = getenv("GPX_TILE32_MAX"); return e ? atol(e) : 1024L; }();
        if (bt == 64 && t64 * (splitk > 1 ? splitk : 1) <= t32max && !(flags & GEMM_NO_TILE32)) bt = 32;
    }
    if (flags & GEMM_TILE64) bt = 64;
    if (flags & GEMM_TILE128) bt = 128;
    g.tiles_m = (int)((M + bt - 1) / bt);
    g.tiles_n = (int)((N + bt - 1) / bt);
    g.ntiles = (flags & GEMM_LOWER) ? g.tiles_m * (g.tiles_m + 1) / 2 : g.tiles_m * g.tiles_n;
    g.xcd_rows = (!(flags & (GEMM_LOWER | GEMM_KLO_M | GEMM_KLO_N | GEMM_KHI_M | GEMM_KHI_N)) && g.splitk == 1 && g.tiles_n <= 32 && g.tiles_m >= 64 && !(flags & GEMM_NO_XCD)) ? 1 : 0;
    // 8x8 squares of 128-tiles per XCD (see kernel): full products, LOWER with a full K range, and K ranges
    // that depend on ONE of the tile indices
    g.sq_m = (g.tiles_m + 7) / 8;
    g.sq_n = (g.tiles_n + 7) / 8;
    g.nsuper = 0;
    long tail = 0;
    // LOWER products with a full K range: T(T+1)/2 tiles of 128 overshoot a whole number of rounds of 512
    // workgroups by T/2 (2080 = 4.06 rounds at 8192 rows: the launch paid for 5).  The main launch therefore
    // takes the T(T-1)/2 tiles strictly below the diagonal (3.94 rounds) and a second, short launch the
    // diagonal 128x128 blocks as three 64x64 tiles each (their upper right quarter is never needed).
    const bool band = (flags & GEMM_LOWER) && !(flags & (GEMM_KLO_M | GEMM_KLO_N | GEMM_KHI_M | GEMM_KHI_N | GEMM_NO_BAND)) &&
                      bt == 128 && g.splitk == 1 && g.tiles_m >= 32 && !has_kscale;
    if (band) {
        g.flags |= GEMM_STRICT_LOWER;
        g.ntiles = g.tiles_m * (g.tiles_m - 1) / 2;
    }
    {
        const int tri = flags & (GEMM_KLO_M | GEMM_KLO_N | GEMM_KHI_M | GEMM_KHI_N);
        const bool one_dep = tri == GEMM_KLO_M || tri == GEMM_KLO_N || tri == GEMM_KHI_M || tri == GEMM_KHI_N;
        const bool form_ok = (flags & GEMM_LOWER) ? tri == 0 : (tri == 0 || one_dep);
        if (!(flags & GEMM_NO_SQUARES) && form_ok && bt == 128 && !g.xcd_rows && g.splitk == 1 && ((g.tiles_m >= 16 && (g.tiles_n >= 16 || g.tiles_n == 8)) || (flags & GEMM_FORCE_SQUARES))) {
            if (flags & GEMM_LOWER) {
                g.nsuper = g.sq_m * (g.sq_m - 1) / 2;
                tail = (band ? 28L : 36L) * g.sq_m;
            } else {
                g.nsuper = g.sq_m * g.sq_n;
            }
            // fewer than 8 squares per XCD leave the XCDs unevenly loaded (8192 LOWER: 28 squares, 9.5 ms
            // against 8.7 ms in plain order)
            if (g.nsuper < 64 && !(flags & GEMM_FORCE_SQUARES)) g.nsuper = 0, tail = 0;
        }
    }
    const long grid = g.xcd_rows ? (long)((g.tiles_m + 7) / 8) * 8 * g.tiles_n
                      : g.nsuper ? (long)((g.nsuper + 7) / 8) * 512 + tail
                                 : (long)g.ntiles * g.splitk;
    p.grid = grid;
    p.bt = bt;
    p.band = band;
    p.band_grid = 0;
    p.band_bt = 64;
    if (band) {
        TileOrder& b = p.band_order;
        b = g;
        b.flags = (g.flags & ~GEMM_STRICT_LOWER) | GEMM_DIAG_BAND;
        b.nsuper = 0;
        b.xcd_rows = 0;
        const int t128 = g.tiles_m;
        // up to 64 diagonal blocks: 32x32 tiles (the band is one long K loop per tile on a CU of its own: 96 64-tiles with
        // K = 4096 took 178 us behind a main launch of 1 ms; ten 32-tiles per block are 10/12 of the flops on 3.3 times
        // the CUs); more blocks fill the chip with 64-tiles
        p.band_bt = t128 <= 64 && !(flags & GEMM_NO_TILE32) ? 32 : 64;
        if (p.band_bt == 32) {
            b.flags |= GEMM_DIAG_BAND32;
            b.tiles_m = b.tiles_n = (int)((M + 31) / 32);
            b.ntiles = 10 * t128;
        } else {
            b.tiles_m = b.tiles_n = (int)((M + 63) / 64);
            b.ntiles = 3 * t128;
        }
        p.band_grid = b.ntiles;
    }
    return p;
}
