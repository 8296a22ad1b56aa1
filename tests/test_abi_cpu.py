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
