"""GPU tests of the fp64 MFMA GEMM (gp_amd/csrc/gemm_f64.hip) through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lib():
    from gp_amd import lib
    return lib


def test_mfma_layout_identity_asymmetric():
    """A = I against an ASYMMETRIC B: a swapped C write or a wrong fragment map cannot hide."""
    lib = _lib()
    n = 64
    B = np.arange(n * n, dtype=np.float64).reshape(n, n) + 0.25  # B[j, k]
    C = lib.dgemm(np.eye(n), B)       # C = I @ B.T
    assert np.array_equal(C, B.T)
    C = lib.dgemm(B, np.eye(n))       # C = B @ I
    assert np.array_equal(C, B)


@pytest.mark.parametrize("a_kc", [False, True])
@pytest.mark.parametrize("b_kc", [False, True])
@pytest.mark.parametrize("shape", [(128, 128, 64), (257, 130, 77), (64, 300, 16), (500, 500, 123), (1024, 896, 512)])
def test_gemm_layouts_and_ragged_shapes(a_kc, b_kc, shape):
    lib = _lib()
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.uniform(-1, 1, (M, K))
    B = rng.uniform(-1, 1, (N, K))
    C0 = rng.uniform(-1, 1, (M, N))
    ref = 0.5 * A @ B.T - 2.0 * C0
    for flags in (lib.GEMM_TILE64, lib.GEMM_TILE128):
        C = lib.dgemm(A, B, C0, a_kc=a_kc, b_kc=b_kc, alpha=0.5, beta=-2.0, flags=flags)
        assert np.max(np.abs(C - ref)) <= 1e-13 * K


def test_gemm_lower_and_triangular_k_ranges():
    lib = _lib()
    n = 384
    rng = np.random.default_rng(5)
    L = np.tril(rng.uniform(-1, 1, (n, n)))
    D = rng.uniform(-1, 1, (n, n))
    for tile in (lib.GEMM_TILE64, lib.GEMM_TILE128):
        # SYRK-like: lower tiles only
        C = lib.dgemm(D, D, flags=lib.GEMM_LOWER | tile)
        assert np.max(np.abs(np.tril(C) - np.tril(D @ D.T))) < 1e-11
        # A lower triangular (k <= m): stop at the diagonal tile
        C = lib.dgemm(L, D.T.copy(), flags=lib.GEMM_KHI_M | lib.GEMM_HEAVY_LAST | tile, b_kc=True)
        assert np.max(np.abs(C - L @ D)) < 1e-11
        # B(n,k) = W[k][n] with W lower triangular (nonzero k >= n): start at the diagonal tile
        C = lib.dgemm(D, L.T.copy(), flags=lib.GEMM_KLO_N | tile, b_kc=True)
        assert np.max(np.abs(C - D @ L)) < 1e-11
        # W^T W, lower tiles, both operands K-contiguous with k >= max(m, n)
        C = lib.dgemm(L.T.copy(), L.T.copy(), a_kc=True, b_kc=True, flags=lib.GEMM_LOWER | lib.GEMM_KLO_M | tile)
        assert np.max(np.abs(np.tril(C) - np.tril(L.T @ L))) < 1e-11


@pytest.mark.parametrize("n", [2048, 2304 + 130])
def test_gemm_square_tile_order_all_forms(n):
    """The 8x8-square tile order (one square per XCD, K walked up or down in alternate squares) must cover
    every output tile exactly once for each operand form it applies to, ragged edges included."""
    lib = _lib()
    K = n
    rng = np.random.default_rng(n)
    D = rng.uniform(-1, 1, (n, K))
    E = rng.uniform(-1, 1, (n, K))
    L = np.tril(rng.uniform(-1, 1, (n, n)))
    sq = lib.GEMM_TILE128 | lib.GEMM_FORCE_SQUARES  # the default from 64 squares on (n >= 8192); forced here
    tol = 1e-13 * K
    C0 = rng.uniform(-1, 1, (n, n))
    C = lib.dgemm(D, E, C0, alpha=0.5, beta=-2.0, flags=sq)
    assert np.max(np.abs(C - (0.5 * D @ E.T - 2.0 * C0))) <= tol
    C = lib.dgemm(D, D, flags=lib.GEMM_LOWER | sq)
    assert np.max(np.abs(np.tril(C) - np.tril(D @ D.T))) <= tol
    # A lower triangular, k <= m (trip count grows with the M tile), heavy tiles first
    C = lib.dgemm(L, E.T.copy(), flags=lib.GEMM_KHI_M | lib.GEMM_HEAVY_LAST | sq, b_kc=True)
    assert np.max(np.abs(C - L @ E)) <= tol
    # B(n,k) = L[n,k], k <= n (trip count grows with the N tile)
    C = lib.dgemm(D, L, flags=lib.GEMM_KHI_N | sq)
    assert np.max(np.abs(C - D @ L.T)) <= tol
    # ... the K tiles of the diagonal 128-block cut at 64 (gemm_kernel<4,4,..,SK=1>) only leave out products with
    # structural zeros: bit-identical to the launch that multiplies them (GEMM_NO_HALF_SKIP), both K directions
    assert np.array_equal(C, lib.dgemm(D, L, flags=lib.GEMM_KHI_N | sq | lib.GEMM_NO_HALF_SKIP))
    assert np.array_equal(lib.dgemm(D, L, flags=lib.GEMM_KHI_N | lib.GEMM_TILE128),
                          lib.dgemm(D, L, flags=lib.GEMM_KHI_N | lib.GEMM_TILE128 | lib.GEMM_NO_HALF_SKIP))
    # B(n,k) = W[k][n], W lower: k >= n
    C = lib.dgemm(D, L.T.copy(), flags=lib.GEMM_KLO_N | sq, b_kc=True)
    assert np.max(np.abs(C - D @ L)) <= tol
    assert np.array_equal(C, lib.dgemm(D, L.T.copy(), flags=lib.GEMM_KLO_N | sq | lib.GEMM_NO_HALF_SKIP, b_kc=True))   # SK = 2
    assert np.array_equal(lib.dgemm(D, L.T.copy(), flags=lib.GEMM_KLO_N | lib.GEMM_TILE128, b_kc=True),
                          lib.dgemm(D, L.T.copy(), flags=lib.GEMM_KLO_N | lib.GEMM_TILE128 | lib.GEMM_NO_HALF_SKIP, b_kc=True))
    # A(m,k) = W[k][m], W lower: k >= m
    C = lib.dgemm(L.T.copy(), E, flags=lib.GEMM_KLO_M | sq, a_kc=True)
    assert np.max(np.abs(C - L.T @ E.T)) <= tol


@pytest.mark.parametrize("n", [4096, 4096 + 70])
@pytest.mark.parametrize("order", ["plain", "squares"])
def test_gemm_lower_strict_launch_plus_diagonal_band(n, order):
    """LOWER products of >= 32 tiles run as two launches: the tiles strictly below the diagonal, then the
    diagonal 128 x 128 blocks as ten 32 x 32 tiles each (up to 64 blocks) or three 64 x 64 tiles each (more
    blocks, or GEMM_NO_TILE32).  Together they must give every element of the lower triangle, and nothing may be
    written 128 or more columns right of the diagonal."""
    lib = _lib()
    K = 192
    rng = np.random.default_rng(n)
    D = rng.uniform(-1, 1, (n, K))
    C0 = rng.uniform(-1, 1, (n, n))
    flags = lib.GEMM_LOWER | (lib.GEMM_FORCE_SQUARES if order == "squares" else 0)
    C = lib.dgemm(D, D, C0, alpha=-1.0, beta=1.0, flags=flags)
    ref = C0 - D @ D.T
    assert np.max(np.abs(np.tril(C) - np.tril(ref))) <= 1e-13 * K
    assert np.array_equal(np.triu(C, 128), np.triu(C0, 128))
    C1 = lib.dgemm(D, D, C0, alpha=-1.0, beta=1.0, flags=flags | lib.GEMM_NO_BAND)   # one launch, as before
    assert np.max(np.abs(np.tril(C1) - np.tril(ref))) <= 1e-13 * K
    C2 = lib.dgemm(D, D, C0, alpha=-1.0, beta=1.0, flags=flags | lib.GEMM_NO_TILE32)   # the band of 64 x 64 tiles
    assert np.max(np.abs(np.tril(C2) - np.tril(ref))) <= 1e-13 * K
    assert np.array_equal(np.triu(C2, 128), np.triu(C0, 128))


def test_gemm_splitk_tall():
    lib = _lib()
    M, N, K = 96, 160, 40000
    rng = np.random.default_rng(9)
    A = rng.uniform(-1, 1, (M, K))
    B = rng.uniform(-1, 1, (N, K))
    C0 = rng.uniform(-1, 1, (M, N))
    C = lib.dgemm(A, B, C0, a_kc=True, b_kc=True, beta=1.0, splitk_work=64 * M * N)
    assert np.max(np.abs(C - (A @ B.T + C0))) < 1e-10
    D = rng.uniform(-1, 1, (M, K))
    C = lib.dgemm(D, D, a_kc=True, b_kc=True, flags=lib.GEMM_LOWER, splitk_work=64 * M * M)
    assert np.max(np.abs(np.tril(C) - np.tril(D @ D.T))) < 1e-10


def test_gemm_splitk_gram_slab_count_that_fills_whole_rounds():
    """m = 1024 LOWER over a long K: 36 tiles of 128, the slab count drops from 64 to 56 (3.94 rounds of
    512 workgroups instead of 4.5), so the K range is cut into unequal, non-power-of-two slabs."""
    lib = _lib()
    M, K = 1024, 1 << 18
    rng = np.random.default_rng(11)
    D = rng.uniform(-1, 1, (M, K))
    C0 = np.tril(rng.uniform(-1, 1, (M, M)))
    C = lib.dgemm(D, D, C0, a_kc=True, b_kc=True, beta=1.0, flags=lib.GEMM_LOWER, splitk_work=64 * M * M)
    assert np.max(np.abs(np.tril(C) - np.tril(D @ D.T + C0))) < 1e-13 * K


@pytest.mark.parametrize("shape", [(96, 160, 40000), (1024, 1024, 8192), (130, 70, 1000)])
def test_gemm_kscale_weighted_gram(shape):
    """A^T diag(s) B with signed weights, the K-dimension scale applied on the way to LDS (interior
    and edge tiles, with and without split-K): the FITC products Kux Gamma^-1 Kxu and B^T diag(w) B."""
    lib = _lib()
    M, N, K = shape
    rng = np.random.default_rng(M + K)
    A = rng.uniform(-1, 1, (M, K))
    B = rng.uniform(-1, 1, (N, K))
    s = rng.normal(size=K)
    ref = (A * s) @ B.T
    C = lib.dgemm(A, B, a_kc=True, b_kc=True, kscale=s)
    assert np.max(np.abs(C - ref)) < 1e-11 * K ** 0.5 * 10
    C = lib.dgemm(A, B, a_kc=True, b_kc=True, kscale=s, splitk_work=64 * M * N)
    assert np.max(np.abs(C - ref)) < 1e-11 * K ** 0.5 * 10
    if M == N:
        C = lib.dgemm(A, A, a_kc=True, b_kc=True, kscale=s, flags=lib.GEMM_LOWER, splitk_work=64 * M * N)
        assert np.max(np.abs(np.tril(C) - np.tril((A * s) @ A.T))) < 1e-11 * K ** 0.5 * 10


@pytest.mark.parametrize("shape", [(256, 128, 64), (1000, 200, 96), (130, 65, 20), (4096, 1024, 1024), (517, 1, 33)])
@pytest.mark.parametrize("with_e", [False, True])
def test_gemm_rowdot_epilogue_ragged_shapes(shape, with_e):
    """The row-dot epilogue (gpx_dgemm_rowdot: rowsum(V o V), rowsum(C o Kxu) and Kxu a of FITC.cpp:215-220, 262-263
    leave the GEMM that forms V / C) against NumPy, full and edge tiles, per part of 64 columns; also with the triangular
    K range the V product uses."""
    lib = _lib()
    M, N, K = shape
    rng = np.random.default_rng(M + N + K)
    A, B = rng.standard_normal((M, K)), rng.standard_normal((N, K))
    E = rng.standard_normal((M, N)) if with_e else None
    v = rng.standard_normal(N) if with_e else None
    C, R1, R2 = lib.dgemm_rowdot(A, B, E, v, alpha=0.7)
    ref = 0.7 * A @ B.T
    assert np.max(np.abs(C - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref))) * K
    W = E if with_e else ref
    parts = 2 * ((N + 127) // 128)
    assert R1.shape == (parts, M)
    cols = np.arange(N)
    for p in range(parts):
        sl = cols[2 * (cols // 128) + (cols % 64) // 32 == p]   # the columns one wave of the 128-tile owns
        r1 = np.sum(ref[:, sl] * W[:, sl], axis=1)
        assert np.max(np.abs(R1[p] - r1)) <= 1e-11 * max(1.0, np.max(np.abs(r1))) * K
        if with_e:
            r2 = E[:, sl] @ v[sl]
            assert np.max(np.abs(R2[p] - r2)) <= 1e-12 * max(1.0, np.max(np.abs(r2))) * 64
    if not with_e and K == N:   # V = Kxu Lu^-T: B upper-triangular as an (N, K) operand = zero for k > n
        Bt = np.tril(B)
        C, R1, _ = lib.dgemm_rowdot(A, Bt, flags=lib.GEMM_KHI_N)
        ref = A @ Bt.T
        assert np.max(np.abs(C - ref)) <= 1e-12 * np.max(np.abs(ref)) * K
        assert np.max(np.abs(R1.sum(axis=0) - np.sum(ref * ref, axis=1))) <= 1e-11 * np.max(np.sum(ref * ref, axis=1)) * K


@pytest.mark.parametrize("shape", [(256, 128, 64), (1000, 200, 96), (130, 65, 20), (8192, 2048, 2048), (517, 1, 33)])
def test_gemm_operand_dot_from_the_fragments(shape):
    """R2 = A evec (Kxu a of VFE.cpp:176) leaves the GEMM Kxu P of VFE.cpp:205-209: from the A fragments of the first
    tile column in the direct-to-LDS loop (full tiles), straight from memory for edge tiles; against NumPy."""
    lib = _lib()
    M, N, K = shape
    rng = np.random.default_rng(M + 3 * N + K)
    A, B, v = rng.standard_normal((M, K)), rng.standard_normal((N, K)), rng.standard_normal(K)
    C, R2 = lib.dgemm_opdot(A, B, v, alpha=0.7)
    ref = 0.7 * A @ B.T
    assert np.max(np.abs(C - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref))) * K
    r2 = A @ v
    assert R2.shape == (M,) and np.max(np.abs(R2 - r2)) <= 1e-13 * max(1.0, np.max(np.abs(r2))) * K


@pytest.mark.parametrize("layout", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_32_tiles_for_products_of_few_tiles(layout):
    """Products of at most 64 tiles of 64x64 take 32x32 tiles (gemm_kernel<1,1,..>, the chains below 1024 rows): every
    operand layout, full, LOWER and triangular K ranges, against NumPy and against the 64x64-tile launch of the same
    product (GEMM_NO_TILE32)."""
    lib = _lib()
    a_kc, b_kc = layout
    rng = np.random.default_rng(5)
    for (M, N, K) in [(512, 512, 512), (200, 130, 77), (96, 96, 300)]:
        A, B = rng.standard_normal((M, K)), rng.standard_normal((N, K))
        C0 = rng.standard_normal((M, N))
        ref = 0.5 * A @ B.T - 2.0 * C0
        C = lib.dgemm(A, B, C0, a_kc=a_kc, b_kc=b_kc, alpha=0.5, beta=-2.0)
        C64 = lib.dgemm(A, B, C0, a_kc=a_kc, b_kc=b_kc, alpha=0.5, beta=-2.0, flags=lib.GEMM_NO_TILE32)
        tol = 1e-13 * K * max(1.0, np.max(np.abs(ref)))
        assert np.max(np.abs(C - ref)) <= tol and np.max(np.abs(C64 - ref)) <= tol
    n = 384
    D = rng.standard_normal((n, n))
    L = np.tril(rng.standard_normal((n, n)))
    C = lib.dgemm(D, D, flags=lib.GEMM_LOWER)
    assert np.max(np.abs(np.tril(C) - np.tril(D @ D.T))) <= 1e-11
    C = lib.dgemm(D, L, flags=lib.GEMM_KHI_N)            # B(n,k) = L[n,k], zero for k > n
    assert np.max(np.abs(C - D @ L.T)) <= 1e-11
    C = lib.dgemm(L, D.T.copy(), flags=lib.GEMM_KHI_M | lib.GEMM_HEAVY_LAST, b_kc=True)   # A(m,k) = L[m,k], zero for k > m
    assert np.max(np.abs(C - L @ D)) <= 1e-11


@pytest.mark.parametrize("n,m,scale", [(50000, 200, False), (40000, 384, True), (300, 96, False)])
def test_splitk_gram_with_row_panelled_operands(n, m, scale):
    """S = Kux [G] Kxu with the n x m operand in ROW PANELS (GemmPanels: element (a, b) at (a / pan) pst + a % pan + b pan,
    one split-K slab per panel) against NumPy, and against the plain column-major call on the same numbers."""
    import ctypes as C
    gpx = _lib()
    L = gpx.lib
    _vp, _i64, _i, _d = C.c_void_p, C.c_int64, C.c_int, C.c_double
    L.gpx_dgemm_splitk_panelled.argtypes = [_i64, _i64, _i64, _d, _vp, _vp, _i64, _i64, _d, _vp, _i64, _i, _vp, _i64, _vp]
    L.gpx_dgemm_splitk_panel_rows.argtypes = [_i64, _i64, _i64, _i, _i64]
    L.gpx_dgemm_splitk_panel_rows.restype = _i64
    work = 64 * m * m
    pan = L.gpx_dgemm_splitk_panel_rows(m, m, n, gpx.GEMM_LOWER, work)
    assert pan % 128 == 0 and pan > 0
    npan = (n + pan - 1) // pan
    pst = pan * m
    rng = np.random.default_rng(n)
    K = rng.normal(size=(m, n))
    g = rng.uniform(0.5, 1.5, size=n)
    P = np.full((npan, m, pan), np.nan)                      # rows of the last panel beyond n are never read
    for p in range(npan):
        r0, r1 = p * pan, min(n, (p + 1) * pan)
        P[p, :, :r1 - r0] = K[:, r0:r1]
    dP, dg = gpx.DeviceBuffer.from_numpy(P), gpx.DeviceBuffer.from_numpy(g)
    w, c = gpx.DeviceBuffer(work), gpx.DeviceBuffer(m * m)
    c.zero()
    gpx.check(L.gpx_dgemm_splitk_panelled(m, m, n, 1.0, dP.ptr, dP.ptr, pan, pst, 0.0, c.ptr, m, gpx.GEMM_LOWER, w.ptr, work,
                                          dg.ptr if scale else None), "panelled")
    got = c.to_numpy().reshape(m, m).T
    want = (K * g) @ K.T if scale else K @ K.T
    low = np.tril(np.ones((m, m), bool))
    assert np.max(np.abs(got - want)[low]) <= 1e-12 * np.max(np.abs(want))
    plain = gpx.dgemm(K, K, a_kc=True, b_kc=True, flags=gpx.GEMM_LOWER, splitk_work=work, kscale=g if scale else None)
    assert np.max(np.abs(got - plain)[low]) <= 1e-12 * np.max(np.abs(want))


@pytest.mark.parametrize("n2,n1", [(128, 128), (200, 136), (512, 512), (1024, 1024), (2048, 2048), (4096, 2048)])
def test_paired_products_of_a_recursion_node(n2, n1):
    """gpx_gemm_pair (one launch for A22 -= L21 L21^T and T = L21 W11 below the 128-tile sizes, two launches above and
    beyond GPX_PAIR_MAX_GRID workgroups) against NumPy and, bit for bit, against the two separate products."""
    import ctypes as C
    gpx = _lib()
    L = gpx.lib
    _vp, _i64 = C.c_void_p, C.c_int64
    L.gpx_dgemm_pair_node.argtypes = [_i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64]
    rng = np.random.default_rng(n2 + n1)
    L21 = rng.normal(size=(n2, n1))
    W11 = np.tril(rng.normal(size=(n1, n1)))
    A22 = rng.normal(size=(n2, n2))
    dL, dW = gpx.DeviceBuffer.from_numpy(L21.T.copy()), gpx.DeviceBuffer.from_numpy(W11.T.copy())   # column-major
    dC0, dC1 = gpx.DeviceBuffer.from_numpy(A22.T.copy()), gpx.DeviceBuffer(n2 * n1)
    dC1.zero()
    gpx.check(L.gpx_dgemm_pair_node(n2, n1, dL.ptr, n2, dW.ptr, n1, dC0.ptr, n2, dC1.ptr, n2), "pair")
    got0, got1 = dC0.to_numpy().reshape(n2, n2).T, dC1.to_numpy().reshape(n1, n2).T
    want0, want1 = A22 - L21 @ L21.T, L21 @ W11
    low = np.tril(np.ones((n2, n2), bool))
    assert np.max(np.abs(got0 - want0)[low]) <= 1e-12 * np.max(np.abs(want0))
    assert np.max(np.abs(got1 - want1)) <= 1e-12 * np.max(np.abs(want1))
    sep0 = gpx.dgemm(L21, L21, C_in=A22, alpha=-1.0, beta=1.0, flags=gpx.GEMM_LOWER)
    sep1 = gpx.dgemm(L21, W11.T.copy(), b_kc=True, flags=gpx.GEMM_KLO_N)
    assert np.array_equal(got0[low], sep0[low]) and np.array_equal(got1, sep1)
