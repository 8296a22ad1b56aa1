"""Host logic of the tile kernel (falcon_amd/csrc/fa_tiles.hip): the builder is plain C++, so it is compiled with g++
next to a small driver (tests/native/tile_builder_check.cpp) and its tables are checked without a GPU:
every node is owned by exactly one tile, every element's per-point outputs have exactly one writer, every (node,
neighbour) block of the CSR matrix is written by exactly one lane (directly or as the mirror of its transpose), the
(element, a, b) entries are consistent with the connectivity and element-ascending, the LDS footprint respects the budget."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "falcon_amd", "csrc")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("tiles") / "tile_builder_check")
    cmd = ["g++", "-O2", "-std=c++17", "-I", CSRC, "-include", "cstddef", "-x", "c++",
           os.path.join(ROOT, "tests", "native", "tile_builder_check.cpp"), os.path.join(CSRC, "fa_tiles.hip"), "-o", exe, "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True)
    vals = dict(line.split() for line in r.stdout.strip().splitlines())
    assert r.returncode == 0, r.stdout + r.stderr
    return {k: float(v) for k, v in vals.items()}


@pytest.mark.parametrize("rank,n", [(3, 13), (3, 32), (2, 50), (2, 257)])
@pytest.mark.parametrize("row_order", [0, 1])
def test_tile_tables_are_consistent(checker, rank, n, row_order):
    v = run(checker, rank, n, row_order)
    assert v["owned_once"] == v["nodes"] and v["elements_written_once"] == v["elements"]
    assert v["blocks_written_once"] == v["blocks"] and v["bad"] == 0
    assert v["lds_bytes"] <= v["lds_budget"]
    if row_order:
        assert v["mirrored"] == 0 and v["lanes"] == v["blocks"]
    else:
        assert v["mirrored"] > 0 and v["lanes"] + v["mirrored"] == v["blocks"]


def test_tiles_are_compact_on_a_structured_hex8_mesh(checker):
    """4x4x2-node tiles (75 elements) on a regular grid: every Gauss point is evaluated ~2.3 times per pass, not 8."""
    v = run(checker, 3, 64, 0)
    assert v["redundancy"] < 2.6


def test_large_records_shrink_the_tiles_but_stay_valid(checker):
    v = run(checker, 3, 16, 0, 120)          # records as large as a 27-point general-tangent element's
    assert v["bad"] == 0 and v["blocks_written_once"] == v["blocks"] and v["lds_bytes"] <= v["lds_budget"]
    w = run(checker, 3, 16, 0, 120, 1)       # the same in the whole-LDS geometry: larger tiles, fewer evaluations per point
    assert w["bad"] == 0 and w["blocks_written_once"] == w["blocks"] and w["lds_bytes"] <= w["lds_budget"]
    assert w["redundancy"] < v["redundancy"] and w["lds_budget"] > v["lds_budget"]


@pytest.mark.parametrize("rank,n,max_el", [(3, 13, 128), (3, 32, 64), (2, 50, 192), (2, 257, 128)])
@pytest.mark.parametrize("fvec_layout", [0, 1])
def test_element_tiles_hold_every_element_once(checker, rank, n, max_el, fvec_layout):
    """build_element_tiles (GET_INT_VECTOR passes): a partition of the elements, every tile lists all nodes of its elements."""
    rec = 16
    v = run(checker, rank, n, fvec_layout, rec, 0, max_el)
    assert v["elements_once"] == v["elements"] and v["bad"] == 0 and v["redundancy"] == 1.0
    assert v["lds_bytes"] <= v["lds_budget"]
    # compact runs of the Morton curve: far fewer nodes than max_el * npe
    assert v["nodes_per_tile"] < (0.35 if rank == 3 else 0.45) * max_el * (8 if rank == 3 else 4)
