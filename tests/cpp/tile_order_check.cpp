// tile_order_check.cpp — CPU check of gp_amd/csrc/tile_order.h: for many launch plans of the fp64 GEMM,
// enumerate every workgroup index of the launch(es) and verify that each output tile (and split-K slab)
// is computed EXACTLY once, that nothing outside the output is computed, and that the K walk direction is
// only ever reversed for K ranges that depend on one tile index.  Prints "case <name> ok|FAIL ..." lines.
#include <cstdio>
#include <map>
#include <tuple>
#include <vector>
#include "../../gp_amd/csrc/tile_order.h"

static int check(const char* name, long M, long N, int flags, int splitk) {
    const TilePlan p = plan_tiles(M, N, flags, splitk, false);
    const int bt = p.bt;
    const long tm = (M + bt - 1) / bt, tn = (N + bt - 1) / bt;
    const bool lower = flags & GEMM_LOWER;
    std::map<std::tuple<int, int, int>, int> seen;  // (bi, bj, ks) in units of the MAIN launch's tile size
    long bad = 0, reversed = 0;
    for (long b = 0; b < p.grid; ++b) {
        int bi, bj, ks;
        bool desc;
        if (!tile_of_block(p.main, (int)b, bi, bj, ks, desc)) continue;
        if (bi < 0 || bj < 0 || bi >= tm || bj >= tn || ks < 0 || ks >= p.main.splitk) ++bad;
        if (lower && bj > bi) ++bad;
        if (p.band && bi == bj) ++bad;  // the diagonal belongs to the band launch
        if (desc) ++reversed;
        ++seen[{bi, bj, ks}];
    }
    // expected tiles of the main launch
    long expect = 0;
    for (long i = 0; i < tm; ++i)
        for (long j = 0; j < tn; ++j) {
            if (lower && (j > i || (p.band && j == i))) continue;
            for (int k = 0; k < p.main.splitk; ++k) {
                ++expect;
                auto it = seen.find({(int)i, (int)j, k});
                if (it == seen.end() || it->second != 1) ++bad;
            }
        }
    if ((long)seen.size() != expect) ++bad;
    const int tri = flags & (GEMM_KLO_M | GEMM_KLO_N | GEMM_KHI_M | GEMM_KHI_N);
    const bool one_dep = tri == GEMM_KLO_M || tri == GEMM_KLO_N || tri == GEMM_KHI_M || tri == GEMM_KHI_N;
    if (reversed && !(one_dep && !lower)) ++bad;
    // band launch: the three 64 x 64 (or ten 32 x 32) tiles on and below the diagonal of every diagonal 128-block
    long band_tiles = 0;
    if (p.band) {
        std::map<std::pair<int, int>, int> bs;
        const int bbt = p.band_bt, per = 128 / bbt;   // tiles per block side
        if (bbt != 32 && bbt != 64) ++bad;
        const long t64 = (M + bbt - 1) / bbt;
        for (long b = 0; b < p.band_grid; ++b) {
            int bi, bj, ks;
            bool desc;
            if (!tile_of_block(p.band_order, (int)b, bi, bj, ks, desc)) continue;
            if (bi >= t64 || bj > bi || bi / per != bj / per || ks != 0 || desc) ++bad;
            ++bs[{bi, bj}];
        }
        for (long i = 0; i < t64; ++i)
            for (long j = (i / per) * per; j <= i; ++j) {
                ++band_tiles;
                auto it = bs.find({(int)i, (int)j});
                if (it == bs.end() || it->second != 1) ++bad;
            }
        if ((long)bs.size() != band_tiles) ++bad;
    }
    printf("case %s %s M=%ld N=%ld flags=%d splitk=%d bt=%d grid=%ld squares=%d xcd_rows=%d band=%d tiles=%ld band_tiles=%ld reversed=%ld\n",
           name, bad ? "FAIL" : "ok", M, N, flags, splitk, bt, p.grid, p.main.nsuper, p.main.xcd_rows, (int)p.band, expect,
           band_tiles, reversed);
    return bad ? 1 : 0;
}

int main() {
    int fails = 0;
    // 32-bit lane offsets of the interior loops: the bench shapes fit, an n = 17M-row K-contiguous operand does not
    const bool off_ok = gemm_lane_offsets_fit_32bit(65536 + 16, false) && gemm_lane_offsets_fit_32bit((1L << 22) + 16, true) &&
                        gemm_lane_offsets_fit_32bit((1L << 22) + 16, false) && !gemm_lane_offsets_fit_32bit(1L << 23, true) &&
                        !gemm_lane_offsets_fit_32bit(17400000L, true) && gemm_lane_offsets_fit_32bit(17400000L, false) &&
                        !gemm_lane_offsets_fit_32bit(80000000L, false);
    std::printf("case lane_offsets_32bit %s M=0 N=0 flags=0 splitk=0\n", off_ok ? "ok" : "FAIL");
    fails += off_ok ? 0 : 1;
    const long sizes[] = {64, 200, 1024, 2048, 2434, 4096, 4166, 8192, 9000, 16384, 17500, 32768};
    for (long n : sizes) {
        fails += check("full", n, n, 0, 1);
        fails += check("lower", n, n, GEMM_LOWER, 1);
        fails += check("lower_noband", n, n, GEMM_LOWER | GEMM_NO_BAND, 1);
        fails += check("lower_forced_squares", n, n, GEMM_LOWER | GEMM_FORCE_SQUARES | GEMM_TILE128, 1);
        fails += check("lower_wtw", n, n, GEMM_LOWER | GEMM_KLO_M, 1);
        fails += check("klo_m", n, n, GEMM_KLO_M, 1);
        fails += check("klo_n", n, n, GEMM_KLO_N, 1);
        fails += check("khi_m_heavy", n, n, GEMM_KHI_M | GEMM_HEAVY_LAST, 1);
        fails += check("khi_n_heavy", n, n, GEMM_KHI_N | GEMM_HEAVY_LAST, 1);
        fails += check("plain_order", n, n, GEMM_NO_SQUARES | GEMM_NO_XCD, 1);
        fails += check("forced_squares", n, n, GEMM_FORCE_SQUARES | GEMM_TILE128, 1);
    }
    // tall-skinny outputs of the sparse models, ragged row counts included
    const long talls[] = {131072, 70000, 1048576};
    for (long n : talls) {
        fails += check("tall_full", n, 1024, 0, 1);
        fails += check("tall_full_2048", n, 2048, 0, 1);
        fails += check("tall_klo_n", n, 1024, GEMM_KLO_N, 1);
        fails += check("tall_khi_n", n, 1024, GEMM_KHI_N | GEMM_HEAVY_LAST, 1);
        fails += check("tall_odd_n", n, 1000, GEMM_KLO_N, 1);
    }
    // split-K Gram products: power-of-two, XCD-private (multiple of 8) and odd slab counts
    const int slabs[] = {2, 7, 8, 15, 16, 56, 64};
    for (int s : slabs) {
        fails += check("gram_1024", 1024, 1024, GEMM_LOWER | GEMM_TILE128, s);
        fails += check("gram_2048", 2048, 2048, GEMM_LOWER | GEMM_TILE128, s);
        fails += check("gram_64tiles", 96, 160, 0, s);
    }
    printf("total_fails %d\n", fails);
    return fails ? 1 : 0;
}
