"""Element-matrix tile kernel (fa_emat_impl.h; Quad4 / Triangle3) against the CPU oracle, through the C ABI: every
model family, both lane splits, accumulate / written pattern / shuffled numbering, force-only passes, meshes of
hundreds of tiles, bitwise repeatability (the kernel has no atomics)."""
import numpy as np
import pytest

from falcon_amd import capi, meshgen
from oracle import pyoracle as po
from tests.conftest import rel_rowmax_err
from tests.test_gpu_parity import run_both, run_micro, vec_err, TOL

pytestmark = pytest.mark.gpu
SHAPES = [capi.QUAD4_G2, capi.TRI3_G1]


@pytest.fixture(params=[1, 2], ids=["lpe1", "lpe2"])
def emat(request, monkeypatch):
    monkeypatch.setenv("FA_EMAT", "1")
    monkeypatch.setenv("FA_EMAT_LPE", str(request.param))
    monkeypatch.setenv("FA_EMAT_LPE_F", str(request.param))
    return request.param


@pytest.mark.parametrize("shape", SHAPES)
def test_emat_model_families(shape, emat):
    run_both(shape, capi.LINEAR_ELASTICITY, capi.HOOKE, state2d=capi.PLANE_STRESS)
    for material in (capi.AMOR_PHASE, capi.BOURDIN_PHASE, capi.MIEHE_PHASE):
        run_both(shape, capi.PHASE_FRACTURE, material, keep=0)
        run_both(shape, capi.PHASE_FRACTURE, material, keep=1, kind=capi.PATTERN_WRITTEN, flags=capi.ACCUMULATE, shuffle=True)
        run_both(shape, capi.PHASE_FRACTURE, material, want_K=False)
    run_both(shape, capi.PHASE_FRACTURE_EXT, capi.AMOR_PHASE)
    run_both(shape, capi.PHASE_FRACTURE_EXTIT, capi.MIEHE_PHASE, flags=capi.ACCUMULATE)
    run_both(shape, capi.PHASE_FRACTURE_EXTIT, capi.BOURDIN_PHASE, want_K=False)
    run_both(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, keep=1, frac=capi.CZM_CORNELISSEN, sigma_t=3.0, young=3e4, gc=0.12, l0=0.25)


@pytest.mark.parametrize("shape", SHAPES)
def test_emat_micromorphic(shape, emat):
    run_micro(shape, capi.MIEHE_PHASE, keep=1, commit=True)
    run_micro(shape, capi.BOURDIN_PHASE, keep=0, want_K=False)
    run_micro(shape, capi.MIEHE_PHASE, keep=0, model=capi.MICRO_PHASE_FRACTURE_EXTIT, commit=True)


@pytest.mark.parametrize("shape,mesh", [(capi.QUAD4_G2, lambda: meshgen.grid_quad4(150, 131)), (capi.TRI3_G1, lambda: meshgen.grid_tri3(120, 97))])
@pytest.mark.parametrize("shuffle", [False, True])
def test_emat_many_tiles_and_bitwise_repeatability(shape, mesh, shuffle, emat):
    coords, conn = mesh()
    coords = meshgen.distort(coords, conn, 0.2, 3)
    if shuffle:
        coords, conn, _ = meshgen.shuffle_nodes(coords, conn, 3)
    kw = dict(young=210.0, poisson=0.3, gc=2.7e-3, l0=0.05, keep_off_diags=True)
    asm = capi.Assembler(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, **kw)
    st = meshgen.smooth_state(coords, 3, l0=0.05)
    cfg = po.make_config(shape, po.PHASE_FRACTURE, po.AMOR, **kw)
    rp, ci = asm.pattern()
    ref = po.assemble(cfg, coords, conn, rp, ci, st)
    K, f, _ = asm.assemble_host(st, np.zeros(asm.nnz), flags=capi.OVERWRITE)
    assert rel_rowmax_err(rp, K, ref["K"]) < TOL and vec_err(f, ref["f"]) < TOL
    assert vec_err(asm.get_history(1), ref["hist_new"]) < TOL
    K2, f2, _ = asm.assemble_host(st, np.zeros(asm.nnz), flags=capi.OVERWRITE | capi.DETERMINISTIC)
    assert np.array_equal(K, K2) and np.array_equal(f, f2)          # no atomics anywhere: every run gives the same bits
    _, f3, _ = asm.assemble_host(st, None, np.zeros(asm.ndof), flags=capi.OVERWRITE)     # default: element tiles, atomic f rows
    assert vec_err(f3, ref["f"]) < TOL
    _, f3, _ = asm.assemble_host(st, None, np.zeros(asm.ndof), flags=capi.OVERWRITE | capi.DETERMINISTIC)
    assert vec_err(f3, ref["f"]) < TOL
    _, f4, _ = asm.assemble_host(st, None, np.zeros(asm.ndof), flags=capi.OVERWRITE | capi.DETERMINISTIC)
    assert np.array_equal(f3, f4)


def test_emat_empty_single_and_unaligned(emat):
    coords = np.array([[0., 0.], [2., 0.], [0., 1.]])
    conn = np.array([[0, 1, 2]], dtype=np.int32)
    asm = capi.Assembler(capi.TRI3_G1, capi.LINEAR_ELASTICITY, capi.HOOKE, coords, conn, young=10.0)
    cfg = po.make_config(po.TRI3_G1, po.LINEAR_ELASTICITY, po.HOOKE, young=10.0)
    rp, ci = asm.pattern()
    u = np.arange(6) * 1e-3
    K, f, _ = asm.assemble_host(u, np.zeros(asm.nnz), flags=capi.OVERWRITE)
    ref = po.assemble(cfg, coords, conn, rp, ci, u)
    assert rel_rowmax_err(rp, K, ref["K"]) < TOL and vec_err(f, ref["f"]) < TOL
    asm0 = capi.Assembler(capi.TRI3_G1, capi.LINEAR_ELASTICITY, capi.HOOKE, coords, np.zeros((0, 3), np.int32))
    K, f, _ = asm0.assemble_host(u, np.zeros(0), np.ones(6), flags=capi.OVERWRITE)
    assert np.all(f == 0)
