// gemm_flags.h — option bits of the fp64 GEMM (gemm_f64.hip); plain C++, shared with the host-only tile
// order model (tile_order.h) and its CPU test.
#pragma once
#define GEMM_LOWER 1        // M == N, only tiles on or below the diagonal are computed
#define GEMM_KLO_M 2        // A(m,k) == 0 for k < m  -> tile skips k < m0
#define GEMM_KLO_N 4        // B(n,k) == 0 for k < n
#define GEMM_KHI_M 8        // A(m,k) == 0 for k > m  -> tile stops at k = m0 + BM
#define GEMM_KHI_N 16       // B(n,k) == 0 for k > n
#define GEMM_HEAVY_LAST 32  // tiles with large m0 carry the most work: launch them first
#define GEMM_TILE64 64      // force 64x64 tiles
#define GEMM_TILE128 128    // force 128x128 tiles
#define GEMM_NO_DLDS 256     // register-staged interior loop instead of direct-to-LDS loads (measurement only)
#define GEMM_NO_XCD 512      // keep the plain tile order for tall-skinny outputs (measurement only)
#define GEMM_NO_SQUARES 1024 // plain tile order instead of one 8x8 square of 128-tiles per XCD (measurement only)
#define GEMM_FORCE_SQUARES 2048  // the square order even where it does not pay (few squares; tests)
#define GEMM_NO_BAND 4096   // LOWER products: keep the diagonal tiles in the main launch (measurement only)
#define GEMM_NO_TILE32 8192  // keep 64x64 tiles for products of at most 64 of them (measurement only)
#define GEMM_NO_HALF_SKIP 16384  // 128-tiles: multiply every K tile of a triangular operand in full (measurement only)
// internal to gemm_f64.hip (set by the host side, not by callers):
#define GEMM_STRICT_LOWER (1 << 16)  // main launch of a LOWER product: tiles strictly below the diagonal only
#define GEMM_DIAG_BAND (1 << 17)     // second launch: the 128x128 diagonal blocks as three 64x64 tiles each
#define GEMM_DIAG_BAND32 (1 << 19)   // ... as ten 32x32 tiles each (with GEMM_DIAG_BAND; few diagonal blocks)
#define GEMM_NO_FAST32 (1 << 18)     // an operand's leading dimension overflows the 32-bit lane offsets of the interior loops
