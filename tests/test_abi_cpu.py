"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/falcon_assemble.h declares,
and fails loudly (no silent CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from falcon_amd import capi, meshgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "falcon_assemble.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fa_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(capi.LIB_PATH)
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(L, n), n
    assert sorted(capi.EXPORTS) == names


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    coords, conn = meshgen.grid_quad4(2, 2)
    with pytest.raises(capi.FaError, match="hip"):
        capi.Assembler(capi.QUAD4_G2, capi.LINEAR_ELASTICITY, capi.HOOKE, coords, conn)


def test_argument_errors_before_any_gpu_call():
    coords, conn = meshgen.grid_quad4(2, 2)
    with pytest.raises(capi.FaError, match="PhaseFractureMaterial"):
        capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.HOOKE, coords, conn)
    with pytest.raises(capi.FaError, match="Hooke"):
        capi.Assembler(capi.QUAD4_G2, capi.LINEAR_ELASTICITY, capi.AMOR_PHASE, coords, conn)
    with pytest.raises(capi.FaError, match="fracture type"):
        capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, frac_type=17)
    bad = conn.copy(); bad[0, 0] = 99
    with pytest.raises(capi.FaError, match="out of range"):
        capi.Assembler(capi.QUAD4_G2, capi.LINEAR_ELASTICITY, capi.HOOKE, coords, bad)
    with pytest.raises(capi.FaError, match="rank"):
        capi.Assembler(capi.HEX8_G2, capi.LINEAR_ELASTICITY, capi.HOOKE, coords, conn)


@pytest.mark.parametrize("shape", range(9))
def test_shape_tables_match_the_oracle(shape):
    """The product's element tables (host code of the library) against the checker's, and partition of unity."""
    from oracle import pyoracle as po
    rank, npe, nip, N, dN, w = capi.shape_tables(shape)
    No, dNo, wo = po.shape_tables(shape)
    assert N.shape == np.asarray(No).shape and dN.shape == np.asarray(dNo).shape
    assert np.abs(N - No).max() < 1e-15 and np.abs(dN - dNo).max() < 1e-14 and np.abs(w - wo).max() < 1e-16
    assert np.abs(N.sum(1) - 1).max() < 1e-14 and np.abs(dN.sum(1)).max() < 1e-14
    assert abs(w.sum() - {2: 4.0, 3: 8.0}[rank] * (0.125 if npe in (3, 6) else 1.0)) < 1e-14


def _custom_pattern(drop=None, half=False):
    """Q4 u-phi mesh with a caller-supplied CSR pattern (rows reversed so that it is not one of the canonical layouts)."""
    from oracle import pyoracle as po
    coords, conn = meshgen.grid_quad4(3, 3)
    rp, ci = po.pattern(coords.shape[0], conn, 3, 2)
    rows = [list(ci[rp[r]:rp[r + 1]][::-1]) for r in range(len(rp) - 1)]
    if half:
        rows = [[c for c in row if c >= r] for r, row in enumerate(rows)]
    if drop is not None:
        r, k = drop
        del rows[r][k]
    rp2 = np.zeros(len(rows) + 1, dtype=np.int64); rp2[1:] = np.cumsum([len(x) for x in rows])
    ci2 = np.array([c for row in rows for c in row], dtype=np.int32)
    return coords, conn, rp2, ci2


def test_caller_pattern_is_validated_before_any_gpu_call():
    """fa_create checks a caller's CSR pattern / dof map on the host: offsets, ranges, and that every entry the model
    writes exists -- a missing entry is FA_ERR_ARG, never a silent drop (MatrixBuilder::addBlock would add or throw)."""
    kw = dict(young=210.0, gc=2.7e-3, l0=0.2)
    mk = lambda rp, ci, c, n, **k: capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, c, n, row_ptr=rp,
                                                  col_idx=ci, ndof=len(rp) - 1, **kw, **k)
    coords, conn, rp, ci = _custom_pattern(drop=(7, 2))
    with pytest.raises(capi.FaError, match="has no entry"):
        mk(rp, ci, coords, conn)
    coords, conn, rp, ci = _custom_pattern(half=True)                 # symmetric-half storage: refused, not half-assembled
    with pytest.raises(capi.FaError, match="has no entry"):
        mk(rp, ci, coords, conn)
    coords, conn, rp, ci = _custom_pattern()
    # the u-phi entries are not written with keepOffDiags = false: a pattern without them is complete
    keep = np.ones(ci.size, bool)
    for r in range(len(rp) - 1):
        for k in range(rp[r], rp[r + 1]):
            if (r % 3 == 2) != (ci[k] % 3 == 2):
                keep[k] = False
    rows = np.repeat(np.arange(len(rp) - 1), np.diff(rp))[keep]
    rp_w = np.zeros(len(rp), dtype=np.int64); np.add.at(rp_w, rows + 1, 1); rp_w = np.cumsum(rp_w)
    try:
        mk(rp_w, ci[keep], coords, conn)
    except capi.FaError as e:                                         # complete pattern: only the missing GPU may stop it
        assert "hip" in str(e).lower()
    with pytest.raises(capi.FaError, match="has no entry"):           # ... but keepOffDiags = true needs them
        mk(rp_w, ci[keep], coords, conn, keep_off_diags=True)
    bad = rp.copy(); bad[0] = 1
    with pytest.raises(capi.FaError, match=r"row_ptr\[0\]"):
        mk(bad, ci, coords, conn)
    bad = rp.copy(); bad[5] = bad[4] - 1
    with pytest.raises(capi.FaError, match="decreases"):
        mk(bad, ci, coords, conn)
    bad = ci.copy(); bad[10] = 10 ** 6
    with pytest.raises(capi.FaError, match="outside"):
        mk(rp, bad, coords, conn)
    nd = np.arange(coords.shape[0] * 3, dtype=np.int64); nd[4] = -3
    with pytest.raises(capi.FaError, match="node_dofs"):
        mk(rp, ci, coords, conn, node_dofs=nd)
