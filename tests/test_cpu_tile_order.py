"""CPU test of the GEMM's tile orders (gp_amd/csrc/tile_order.h, shared verbatim by the HIP kernel and the
host launch code): tests/cpp/tile_order_check.cpp, compiled here with g++, enumerates every workgroup index
of 168 launch plans (full / LOWER / triangular-K products from 64 to 32768 rows, the tall outputs of the
sparse models, split-K slab counts) and checks that every output tile and K slab is computed exactly once."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cases(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("tile") / "tile_order_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "tile_order_check.cpp")],
                   check=True, timeout=120)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300).stdout
    rows = []
    for line in out.splitlines():
        f = line.split()
        if f and f[0] == "case":
            rows.append({"name": f[1], "ok": f[2] == "ok", **{k: int(v) for k, v in (x.split("=") for x in f[3:])}})
    assert "total_fails 0" in out, out[-2000:]
    return rows


def test_every_tile_exactly_once(cases):
    assert len(cases) >= 160 and all(c["ok"] for c in cases), [c for c in cases if not c["ok"]]


def test_the_orders_that_the_measurements_chose_are_the_ones_planned(cases):
    by = {(c["name"], c["M"], c["N"]): c for c in cases if c["splitk"] == 1}
    # one 8x8 square per XCD from 64 squares on; alternating K direction for one-index K ranges only
    assert by[("full", 32768, 32768)]["squares"] == 1024 and by[("full", 32768, 32768)]["reversed"] == 0
    assert by[("khi_m_heavy", 16384, 16384)]["squares"] == 256 and by[("khi_m_heavy", 16384, 16384)]["reversed"] == 8192
    assert by[("full", 4096, 4096)]["squares"] == 0 and by[("lower", 8192, 8192)]["squares"] == 0
    assert by[("lower_wtw", 16384, 16384)]["squares"] == 0          # K range depends on both indices
    # LOWER from 32 tiles on: strictly-lower launch (whole rounds of 512) + diagonal band: ten 32-tiles per diagonal
    # 128-block up to 64 blocks, three 64-tiles per block beyond
    c = by[("lower", 8192, 8192)]
    assert c["band"] == 1 and c["tiles"] == 2016 and c["band_tiles"] == 640
    assert by[("lower", 4096, 4096)]["band_tiles"] == 320 and by[("lower", 16384, 16384)]["band_tiles"] == 384
    assert by[("lower", 2048, 2048)]["band"] == 0 and by[("lower_noband", 8192, 8192)]["tiles"] == 2080
    # tall outputs: per-XCD rows for full products, squares one square wide for the triangular ones
    assert by[("tall_full", 1048576, 1024)]["xcd_rows"] == 1 and by[("tall_klo_n", 1048576, 1024)]["squares"] == 1024
