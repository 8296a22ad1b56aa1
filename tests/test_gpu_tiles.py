"""Tile kernel (falcon_amd/csrc/fa_tile_impl.h) on meshes large enough for hundreds of tiles with halos: parity with the
oracle to 1e-12 through the C ABI, for the shapes it is the default of, with shuffled node numbering, the Ext state
arrays, off-diagonal blocks, += semantics, GET_INT_VECTOR passes, and bitwise repeatability of K (every entry has one
writer and a fixed summation order)."""
import os

import numpy as np
import pytest

from falcon_amd import capi, meshgen
from oracle import pyoracle as po
from tests.test_gpu_parity import rel_rowmax_err, vec_err

pytestmark = pytest.mark.gpu

CASES = [  # shape, mesh, model, material, keep
    (capi.HEX8_G2, lambda: meshgen.grid_hex8(14, 12, 11), capi.PHASE_FRACTURE, capi.AMOR_PHASE, 0),
    (capi.HEX8_G2, lambda: meshgen.grid_hex8(9, 10, 12), capi.PHASE_FRACTURE_EXTIT, capi.BOURDIN_PHASE, 1),
    (capi.HEX8_G2, lambda: meshgen.grid_hex8(12, 9, 9), capi.LINEAR_ELASTICITY, capi.HOOKE, 0),
    (capi.HEX20_G2, lambda: meshgen.grid_hex20(6, 5, 5), capi.PHASE_FRACTURE, capi.MIEHE_PHASE, 1),
    (capi.HEX20_G3, lambda: meshgen.grid_hex20(5, 4, 4), capi.LINEAR_ELASTICITY, capi.HOOKE, 0),
    (capi.QUAD8_G3, lambda: meshgen.grid_quad8(40, 33), capi.PHASE_FRACTURE, capi.AMOR_PHASE, 1),
    (capi.QUAD8_G2, lambda: meshgen.grid_quad8(31, 37), capi.MICRO_PHASE_FRACTURE_EXT, capi.MIEHE_PHASE, 0),
    (capi.TRI6_G3, lambda: meshgen.grid_tri6(35, 41), capi.PHASE_FRACTURE_EXT, capi.AMOR_PHASE, 0),
    (capi.TRI3_G3, lambda: meshgen.grid_tri3(50, 45), capi.PHASE_FRACTURE, capi.MIEHE_PHASE, 1),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_tile_kernel_parity_on_many_tiles(case):
    import torch
    shape, mesh, model, material, keep = CASES[case]
    coords, conn = mesh()
    linear = shape in (capi.HEX8_G2, capi.TRI3_G3)
    coords = meshgen.distort(coords, conn, 0.12 if linear else 0.05, 3)
    coords, conn, _ = meshgen.shuffle_nodes(coords, conn, 5)               # the builder sorts spatially, whatever the numbering
    rank = coords.shape[1]
    pf = model != capi.LINEAR_ELASTICITY
    nd = rank + (1 if pf else 0)
    micro = model >= capi.MICRO_PHASE_FRACTURE
    kw = dict(young=210.0, poisson=0.3, frac_type=capi.AT2, gc=2.7e-3, l0=0.2, beta=250.0 if micro else 6500.0, keep_off_diags=keep)
    cfg = po.make_config(shape, model, material, **kw)
    asm = capi.Assembler(shape, model, material, coords, conn, **kw)
    rp, ci = asm.pattern()
    rpo, cio = po.pattern(coords.shape[0], conn, nd, rank, po.PATTERN_FULL, po.keep_code(cfg))
    assert np.array_equal(rp, rpo) and np.array_equal(ci, cio)
    rng = np.random.default_rng(11)
    st = meshgen.smooth_state(coords, nd, seed=2, amp=5e-3, noise=1e-4, l0=0.3)
    st0 = st.copy(); st00 = st.copy()
    if pf:
        st0[rank::nd] += rng.uniform(-0.02, 0.02, coords.shape[0]); st00[rank::nd] += rng.uniform(-0.02, 0.02, coords.shape[0])
    hist = None
    if model in (capi.PHASE_FRACTURE,) or micro:
        hist = rng.uniform(0, 0.3 if micro else 2e-3, conn.shape[0] * asm.nip)
        asm.set_history(0, hist)
    ext = model in (capi.PHASE_FRACTURE_EXT, capi.PHASE_FRACTURE_EXTIT)
    mext = model in (capi.MICRO_PHASE_FRACTURE_EXT, capi.MICRO_PHASE_FRACTURE_EXTIT)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)
    args = dict(state0=t(st0) if ext else None, state00=t(st00) if (ext or mext) else None, dt=0.7, dt0=0.4)
    wants_err = model in (capi.PHASE_FRACTURE_EXTIT, capi.MICRO_PHASE_FRACTURE_EXTIT)
    err = torch.zeros(1, dtype=torch.float64, device=dev) if wants_err else None
    K = torch.full((asm.nnz,), 9.0, dtype=torch.float64, device=dev); f = torch.full((asm.ndof,), -4.0, dtype=torch.float64, device=dev)
    asm.assemble(t(st), K, f, err_ext=err, flags=capi.OVERWRITE | capi.STORE_STRAIN, **args)
    torch.cuda.synchronize()
    ref = po.assemble(cfg, coords, conn, rp, ci, st, state0=st0 if ext else None, state00=st00 if (ext or mext) else None,
                      dt=0.7, dt0=0.4, hist_pre=hist, want_strain=True)
    assert rel_rowmax_err(rp, K.cpu().numpy(), ref["K"]) < 1e-12
    assert vec_err(f.cpu().numpy(), ref["f"]) < 1e-12
    assert np.abs(asm.get_strain_cache() - ref["strain"]).max() < 1e-12 * np.abs(ref["strain"]).max()
    if hist is not None:
        assert np.abs(asm.get_history(1) - ref["hist_new"]).max() < 1e-12 * max(1.0, np.abs(ref["hist_new"]).max())
    if wants_err:
        assert abs(err.item() - ref["err_ext"]) < 1e-12 * ref["err_ext"]
    # K is bitwise repeatable; += semantics on top of a non-zero matrix; GET_INT_VECTOR pass
    K2 = torch.zeros_like(K); f2 = torch.zeros_like(f)
    asm.assemble(t(st), K2, f2, flags=capi.OVERWRITE, **args)
    assert torch.equal(K, K2)
    K3 = torch.full_like(K, 2.0); f3 = torch.full_like(f, 1.0)
    asm.assemble(t(st), K3, f3, flags=capi.ACCUMULATE, **args)
    f4 = torch.zeros_like(f)
    asm.assemble(t(st), None, f4, flags=capi.OVERWRITE, **args)
    torch.cuda.synchronize()
    kmax = K.abs().max().item(); fmax = f.abs().max().item()
    # without a strain cache the 3-D elastic kernels take f_int from the pair blocks (K u) instead of the Gauss points
    assert (f2 - f).abs().max().item() < 1e-12 * fmax
    assert (K3 - 2.0 - K).abs().max().item() < 1e-12 * kmax and (f3 - 1.0 - f).abs().max().item() < 1e-12 * fmax
    assert (f4 - f).abs().max().item() < 1e-12 * fmax
    # the GET_INT_VECTOR pass (Hex8: the force-vector kernel) rewrites newHist_ as well, and repeats bit for bit
    if hist is not None:
        assert np.abs(asm.get_history(1) - ref["hist_new"]).max() < 1e-12 * max(1.0, np.abs(ref["hist_new"]).max())
    # += semantics of a GET_INT_VECTOR pass (3-D shapes: element tiles, f rows added with global atomics)
    f6 = torch.full_like(f, 1.0)
    asm.assemble(t(st), None, f6, flags=capi.ACCUMULATE, **args)
    assert (f6 - 1.0 - f).abs().max().item() < 1e-12 * fmax
    if shape == capi.HEX8_G2 and os.environ.get("FA_FVEC") != "0":   # (FA_FVEC=0: f4 came from the tile kernel's LDS atomics)
        # FA_DETERMINISTIC: the force-vector kernel on node tiles, ordered sums -- repeats bit for bit; the default pass runs
        # on element tiles (atomic adds) unless FA_ELEM_TILES=0
        f5 = torch.zeros_like(f); f7 = torch.full_like(f, 3.0)
        asm.assemble(t(st), None, f5, flags=capi.OVERWRITE | capi.DETERMINISTIC, **args)
        asm.assemble(t(st), None, f7, flags=capi.OVERWRITE | capi.DETERMINISTIC, **args)
        assert torch.equal(f5, f7)
        assert (f5 - f).abs().max().item() < 1e-12 * fmax
        if os.environ.get("FA_ELEM_TILES") == "0":
            assert torch.equal(f4, f5)


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4])
def test_force_only_on_node_tiles(case, monkeypatch):
    """FA_ELEM_TILES=0: the GET_INT_VECTOR passes of the 3-D shapes on node tiles (what FA_DETERMINISTIC and FA_PART_* passes
    run on); the default is element tiles (fa_tiles.h)."""
    monkeypatch.setenv("FA_ELEM_TILES", "0")
    test_tile_kernel_parity_on_many_tiles(case)


@pytest.mark.parametrize("case", [0, 1])
def test_hex8_force_only_through_the_tile_kernel(case, monkeypatch):
    """FA_FVEC=0: what a Hex8 mesh with a node of more than 8 elements gets for GET_INT_VECTOR passes."""
    monkeypatch.setenv("FA_FVEC", "0")
    test_tile_kernel_parity_on_many_tiles(case)


def test_lodi_sums_on_the_device():
    """LodiModel's load / displacement sums over a node group (LodiModel.cpp:263-283) from the
    device-resident f_int and state: same value as the serial loop to 1e-12 of sum|terms|, skipped entries (< 0)
    as in the reference, bitwise repeatable."""
    import torch
    coords, conn = meshgen.grid_quad4(60, 40)
    asm = capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, young=210.0, poisson=0.3, gc=2.7e-3, l0=0.2)
    dev = torch.device("cuda", 0)
    st = meshgen.smooth_state(coords, 3, seed=2, amp=5e-3, noise=1e-4, l0=0.3)
    f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    asm.assemble(torch.from_numpy(st).to(dev), None, f, flags=capi.OVERWRITE)
    top = np.nonzero(np.abs(coords[:, 1] - coords[:, 1].max()) < 1e-12)[0]             # node group "top"
    fh = f.cpu().numpy()
    for typ in range(2):                                                               # dof types dx, dy
        idofs = top * 3 + typ
        idofs[::7] = -1                                                                # dofs that do not exist: skipped
        load = 0.0; disp = 0.0; scale = 0.0
        for i in idofs:                                                                # the reference's loop
            if i >= 0:
                load += fh[i]; disp += st[i]; scale += abs(fh[i])
        idx = torch.from_numpy(idofs).to(dev)
        got = asm.gather_sum(f, idx)
        got2 = asm.gather_sum(f, idx)
        gd = asm.gather_sum(torch.from_numpy(st).to(dev), idx)
        torch.cuda.synchronize()
        assert abs(got.item() - load) <= 1e-12 * scale and got.item() == got2.item()
        assert abs(gd.item() - disp) <= 1e-12 * np.abs(st[idofs[idofs >= 0]]).sum()
    empty = asm.gather_sum(f, torch.zeros(0, dtype=torch.int64, device=dev))
    assert empty.item() == 0.0


@pytest.mark.parametrize("shape,mesh,model,material", [
    (capi.QUAD4_G2, lambda: meshgen.grid_quad4(30, 21), capi.PHASE_FRACTURE, capi.AMOR_PHASE),
    (capi.HEX8_G2, lambda: meshgen.grid_hex8(7, 6, 5), capi.PHASE_FRACTURE, capi.MIEHE_PHASE),
    (capi.TRI6_G3, lambda: meshgen.grid_tri6(12, 9), capi.LINEAR_ELASTICITY, capi.HOOKE),
    (capi.HEX20_G2, lambda: meshgen.grid_hex20(3, 3, 2), capi.LINEAR_ELASTICITY, capi.HOOKE)])
def test_strain_table_matches_get_strain(shape, mesh, model, material):
    """getStrain_ (PhaseFractureModel.cpp:1282-1395) from the device-resident strain_ cache: nodal sums and weights to
    1e-12 of the oracle's element loop, added on top of what the table already holds (addBlock semantics)."""
    import torch
    coords, conn = mesh()
    coords = meshgen.distort(coords, conn, 0.05, 2)
    coords, conn, _ = meshgen.shuffle_nodes(coords, conn, 9)
    rank = coords.shape[1]
    nd = rank + (0 if model == capi.LINEAR_ELASTICITY else 1)
    kw = dict(young=210.0, poisson=0.3, gc=2.7e-3, l0=0.2)
    asm = capi.Assembler(shape, model, material, coords, conn, **kw)
    dev = torch.device("cuda", 0)
    st = meshgen.smooth_state(coords, nd, seed=2, amp=5e-3, noise=1e-4, l0=0.3)
    f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    table = torch.full((asm.nnode, 4 if rank == 2 else 6), 0.5, dtype=torch.float64, device=dev)
    weights = torch.full((asm.nnode,), 0.25, dtype=torch.float64, device=dev)
    with pytest.raises(capi.FaError, match="FA_STORE_STRAIN"):
        asm.strain_table(table, weights)
    asm.assemble(torch.from_numpy(st).to(dev), None, f, flags=capi.OVERWRITE | capi.STORE_STRAIN)
    asm.strain_table(table, weights)
    torch.cuda.synchronize()
    cfg = po.make_config(shape, model, material, **kw)
    rp, ci = asm.pattern()
    ref = po.assemble(cfg, coords, conn, rp, ci, st, want_K=False, want_strain=True)
    tr, wr = po.strain_table(shape, coords.shape[0], conn, ref["strain"])
    assert np.abs(table.cpu().numpy() - 0.5 - tr).max() < 1e-12 * np.abs(tr).max()
    assert np.abs(weights.cpu().numpy() - 0.25 - wr).max() < 1e-12 * np.abs(wr).max()


@pytest.mark.parametrize("shape,mesh,model,material,state", [
    (capi.QUAD4_G2, lambda: meshgen.grid_quad4(17, 11), capi.PHASE_FRACTURE, capi.AMOR_PHASE, capi.PLANE_STRAIN),
    (capi.QUAD4_G2, lambda: meshgen.grid_quad4(9, 8), capi.PHASE_FRACTURE_EXT, capi.BOURDIN_PHASE, capi.PLANE_STRESS),
    (capi.QUAD8_G3, lambda: meshgen.grid_quad8(6, 5), capi.PHASE_FRACTURE, capi.MIEHE_PHASE, capi.PLANE_STRAIN),
    (capi.TRI3_G1, lambda: meshgen.grid_tri3(8, 7), capi.LINEAR_ELASTICITY, capi.HOOKE, capi.PLANE_STRESS),
    (capi.TRI6_G3, lambda: meshgen.grid_tri6(5, 6), capi.LINEAR_ELASTICITY, capi.HOOKE, capi.PLANE_STRAIN),
    (capi.HEX8_G2, lambda: meshgen.grid_hex8(5, 4, 4), capi.PHASE_FRACTURE, capi.MIEHE_PHASE, capi.PLANE_STRAIN),
    (capi.HEX8_G2, lambda: meshgen.grid_hex8(4, 4, 3), capi.MICRO_PHASE_FRACTURE, capi.BOURDIN_PHASE, capi.PLANE_STRAIN),
    (capi.HEX20_G3, lambda: meshgen.grid_hex20(2, 3, 2), capi.PHASE_FRACTURE, capi.AMOR_PHASE, capi.PLANE_STRAIN)])
def test_stress_table_matches_get_stress(shape, mesh, model, material, state):
    """getStress_ (PhaseFractureModel.cpp:1165-1271): material update from the cached strain and the cached degradation,
    all components (zz in 2-D), nodal sums and weights to 1e-12 of the oracle's element loop."""
    import torch
    coords, conn = mesh()
    coords = meshgen.distort(coords, conn, 0.05, 2)
    rank = coords.shape[1]
    pf = model != capi.LINEAR_ELASTICITY
    nd = rank + (1 if pf else 0)
    micro = model >= capi.MICRO_PHASE_FRACTURE
    kw = dict(state=state, young=210.0, poisson=0.3, gc=2.7e-3, l0=0.2, beta=250.0 if micro else 6500.0)
    asm = capi.Assembler(shape, model, material, coords, conn, **kw)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)
    st = meshgen.smooth_state(coords, nd, seed=4, amp=5e-3, noise=1e-4, l0=0.3)
    st[0::nd] -= 8e-3 * coords[:, 0]                                      # part of the mesh in compression (tr < 0 branches)
    st0 = st * 0.9; st00 = st * 0.8
    ext = model == capi.PHASE_FRACTURE_EXT
    f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    asm.assemble(t(st), None, f, state0=t(st0) if ext else None, state00=t(st00) if ext else None, dt=0.5, dt0=0.4,
                 flags=capi.OVERWRITE | capi.STORE_STRAIN)
    table = torch.zeros((asm.nnode, 4 if rank == 2 else 6), dtype=torch.float64, device=dev)
    weights = torch.zeros((asm.nnode,), dtype=torch.float64, device=dev)
    asm.stress_table(table, weights)
    torch.cuda.synchronize()
    cfg = po.make_config(shape, model, material, **kw)
    rp, ci = asm.pattern()
    ref = po.assemble(cfg, coords, conn, rp, ci, st, state0=st0 if ext else None, state00=st00 if ext else None, dt=0.5, dt0=0.4,
                      want_K=False, want_strain=True)
    tr, wr = po.stress_table(cfg, coords.shape[0], conn, ref["strain"])
    if rank == 2:
        assert (np.abs(tr[:, 2]).max() == 0.0) == (state == capi.PLANE_STRESS and material in (capi.HOOKE, capi.BOURDIN_PHASE))
    assert np.abs(table.cpu().numpy() - tr).max() < 1e-12 * np.abs(tr).max()
    assert np.abs(weights.cpu().numpy() - wr).max() < 1e-12 * np.abs(wr).max()


@pytest.mark.parametrize("shape,mesh,material,cols", [
    (capi.QUAD4_G2, lambda: meshgen.grid_quad4(13, 9), capi.AMOR_PHASE, [5, 4]),            # vtk.data = "stress_yy | stress_xx"
    (capi.QUAD4_G2, lambda: meshgen.grid_quad4(8, 7), capi.MIEHE_PHASE, list(range(8))),
    (capi.HEX8_G2, lambda: meshgen.grid_hex8(4, 4, 3), capi.BOURDIN_PHASE, [0, 7, 11, 3])])
def test_output_data_matches_get_output_data(shape, mesh, material, cols):
    """getOutputData_ (PhaseFractureModel.cpp:1408-1508): the filtered [strain.tensor | stress.tensor] columns extrapolated
    to the nodes, added to the table, weights added once per element; against the oracle's two element loops."""
    import torch
    coords, conn = mesh()
    coords = meshgen.distort(coords, conn, 0.05, 6)
    rank = coords.shape[1]
    nd = rank + 1
    kw = dict(young=210.0, poisson=0.3, gc=2.7e-3, l0=0.2)
    asm = capi.Assembler(shape, capi.PHASE_FRACTURE, material, coords, conn, **kw)
    dev = torch.device("cuda", 0)
    st = meshgen.smooth_state(coords, nd, seed=4, amp=5e-3, noise=1e-4, l0=0.3)
    st[0::nd] -= 8e-3 * coords[:, 0]
    f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    table = torch.full((asm.nnode, len(cols)), 0.5, dtype=torch.float64, device=dev)
    weights = torch.full((asm.nnode,), 0.25, dtype=torch.float64, device=dev)
    with pytest.raises(capi.FaError, match="FA_STORE_STRAIN"):
        asm.output_data(cols, table, weights)
    asm.assemble(torch.from_numpy(st).to(dev), None, f, flags=capi.OVERWRITE | capi.STORE_STRAIN)
    asm.output_data(cols, table, weights)
    torch.cuda.synchronize()
    with pytest.raises(capi.FaError, match="out of range"):
        asm.output_data([99], table[:, :1].contiguous(), weights)
    cfg = po.make_config(shape, capi.PHASE_FRACTURE, material, **kw)
    rp, ci = asm.pattern()
    ref = po.assemble(cfg, coords, conn, rp, ci, st, want_K=False, want_strain=True)
    te, we = po.strain_table(shape, coords.shape[0], conn, ref["strain"])
    ts, ws = po.stress_table(cfg, coords.shape[0], conn, ref["strain"])
    both = np.concatenate([te, ts], axis=1)[:, cols]
    got = table.cpu().numpy() - 0.5
    for j in range(len(cols)):
        scale = np.abs(both[:, j]).max()
        assert np.abs(got[:, j] - both[:, j]).max() <= 1e-12 * max(scale, 1e-300)
    assert np.abs(weights.cpu().numpy() - 0.25 - we).max() < 1e-12 * np.abs(we).max()


@pytest.mark.parametrize("shape,mesh", [(capi.HEX8_G2, lambda: meshgen.grid_hex8(6, 5, 5)), (capi.QUAD4_G2, lambda: meshgen.grid_quad4(20, 17)),
                                        (capi.TRI6_G3, lambda: meshgen.grid_tri6(9, 8))])
def test_value_array_that_is_only_8_byte_aligned(shape, mesh):
    """A caller's K array need not be 16-byte aligned (a slice of a larger allocation): same values as with an aligned one."""
    import torch
    coords, conn = mesh()
    kw = dict(young=210.0, poisson=0.3, gc=2.7e-3, l0=0.2)
    asm = capi.Assembler(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, **kw)
    dev = torch.device("cuda", 0)
    st = torch.from_numpy(meshgen.smooth_state(coords, asm.ndofpn, seed=2, amp=5e-3, noise=1e-4, l0=0.3)).to(dev)
    K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    asm.assemble(st, K, f, flags=capi.OVERWRITE)
    big = torch.full((asm.nnz + 3,), 5.0, dtype=torch.float64, device=dev)
    Ko = big[1:1 + asm.nnz]
    assert Ko.data_ptr() % 16 == 8
    f2 = torch.zeros_like(f)
    asm.assemble(st, Ko, f2, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    assert (Ko - K).abs().max().item() <= 1e-13 * K.abs().max().item()
    assert big[0].item() == 5.0 and big[-1].item() == 5.0 and big[-2].item() == 5.0      # nothing written outside


def test_full_size_invariants_cfg4():
    """BASELINE config 4 at its full size (200^3 Hex8, u-phi, AT2, Amor; 3.47 G non-zeros): too large for the oracle, so
    the tile kernel (box tiles, mirrored pair blocks, point-major Gauss-point lanes) and the Hex8 force-vector kernel
    are checked through size-independent properties, as test_full_size_invariants_cfg3 does for the lane-pair kernel:
    every entry of the pattern written, internal forces of every displacement component sum to zero and a rigid
    translation is in the null space of the u-u block (partition of unity), exact zeros in the u-phi / phi-u blocks,
    positive K4 row sums, K bitwise repeatable, FA_ACCUMULATE on a zeroed K equal to FA_OVERWRITE, and the
    GET_INT_VECTOR pass (another kernel) returning the same f_int and history."""
    import torch
    n = 200
    coords, conn = meshgen.grid_hex8(n, n, n)
    kw = dict(young=210e3, poisson=0.3, gc=2.7, l0=0.015, frac_type=capi.AT2, keep_off_diags=False)
    asm = capi.Assembler(capi.HEX8_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, **kw)
    assert asm.nel == n ** 3 and asm.nnz == 16 * (3 * n + 1) ** 3
    dev = torch.device("cuda", 0)
    st = torch.from_numpy(meshgen.smooth_state(coords, 4)).to(dev)
    del coords, conn
    K = torch.empty(asm.nnz, dtype=torch.float64, device=dev).fill_(float("nan"))
    f = torch.empty(asm.ndof, dtype=torch.float64, device=dev).fill_(float("nan"))
    asm.assemble(st, K, f, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    assert not torch.isnan(K).any() and not torch.isnan(f).any()
    fmax = f.abs().max().item(); kmax = K.abs().max().item()
    for c in range(3):
        assert abs(f[c::4].sum().item()) < 1e-9 * fmax * np.sqrt(asm.ndof)
    Kb = K.view(-1, 4)                                                           # blocked [neighbour][column type]
    # first entry group of every row from the mesh itself (fetching the 3.47 G column indices would take minutes): a node
    # of the structured grid has 2 or 3 neighbours per direction, each of its 4 rows holds that many groups of 4 entries
    c1 = np.full(n + 1, 3, dtype=np.int64); c1[0] = c1[-1] = 2
    deg = (c1[:, None, None] * c1[None, :, None] * c1[None, None, :]).ravel()    # any axis order: the product is symmetric
    grp = torch.from_numpy(np.concatenate([[0], np.cumsum(np.repeat(deg, 4))])).to(dev)
    assert int(grp[-1].item()) * 4 == asm.nnz
    rowsum = torch.empty(asm.ndof, 4, dtype=torch.float64, device=dev)
    for c in range(4):                                                           # one contiguous 1-D scan per column type
        cs = torch.zeros(Kb.shape[0] + 1, dtype=torch.float64, device=dev)
        torch.cumsum(Kb[:, c].contiguous(), 0, out=cs[1:])
        rowsum[:, c] = cs[grp[1:]] - cs[grp[:-1]]
        del cs
    urows = torch.arange(asm.ndof, device=dev) % 4 < 3
    assert rowsum[urows][:, :3].abs().max().item() < 1e-6 * kmax                 # translation null space (cumsum rounding)
    assert rowsum[~urows][:, 3].min().item() > 0.0                               # K4 row sums
    # exact zeros in the blocks the model does not write (differences of running sums cannot show that): entry by entry
    g_is_u = torch.repeat_interleave(urows, grp[1:] - grp[:-1])                  # per entry group: does its row belong to u?
    assert Kb[g_is_u][:, 3].abs().max().item() == 0.0                            # u-phi block
    assert Kb[~g_is_u][:, :3].abs().max().item() == 0.0                          # phi-u block
    del rowsum, Kb, grp, g_is_u
    h1 = asm.get_history(1).copy()
    f2 = torch.zeros_like(f)
    asm.assemble(st, None, f2, flags=capi.OVERWRITE)                             # Hex8 force-vector kernel
    torch.cuda.synchronize()
    assert (f - f2).abs().max().item() < 1e-12 * fmax
    assert np.abs(asm.get_history(1) - h1).max() <= 1e-13 * np.abs(h1).max()
    K2 = torch.zeros_like(K); f3 = torch.zeros_like(f)
    asm.assemble(st, K2, f3, flags=capi.ACCUMULATE)
    torch.cuda.synchronize()
    assert (K - K2).abs().max().item() < 1e-12 * kmax and (f - f3).abs().max().item() < 1e-12 * fmax
    K2.zero_()
    asm.assemble(st, K2, f3, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    assert torch.equal(K, K2)                                                    # one writer per entry, fixed summation order


def test_full_size_invariants_cfg5():
    """BASELINE config 5 at its full size (125 x 125 x 128 Hex20, linear elasticity, Gauss 3x3x3; 4.3 G non-zeros): the
    tile kernel's grouped pair lanes and its K-u force path through size-independent properties -- every entry written,
    internal forces in equilibrium, rigid translations in the null space of K (row sums per column type), K bitwise
    repeatable, f_int from the pair blocks (K u) equal to f_int from the Gauss points (the pass with a strain cache),
    FA_ACCUMULATE on a zeroed K equal to FA_OVERWRITE."""
    import torch
    coords, conn = meshgen.grid_hex20(125, 125, 128)
    asm = capi.Assembler(capi.HEX20_G3, capi.LINEAR_ELASTICITY, capi.HOOKE, coords, conn, young=210e3, poisson=0.3)
    assert asm.nel == 125 * 125 * 128
    dev = torch.device("cuda", 0)
    st = torch.from_numpy(meshgen.smooth_state(coords, 3)).to(dev)
    del coords, conn
    K = torch.empty(asm.nnz, dtype=torch.float64, device=dev).fill_(float("nan"))
    f = torch.empty(asm.ndof, dtype=torch.float64, device=dev).fill_(float("nan"))
    asm.assemble(st, K, f, flags=capi.OVERWRITE)                                 # f_int = K u from the pair lanes
    torch.cuda.synchronize()
    assert not torch.isnan(K).any() and not torch.isnan(f).any()
    fmax = f.abs().max().item(); kmax = K.abs().max().item()
    for c in range(3):
        assert abs(f[c::3].sum().item()) < 1e-9 * fmax * np.sqrt(asm.ndof)
    Kb = K.view(-1, 3)                                                           # blocked [neighbour][column type]
    grp = asm.row_ptr_device() // 3
    assert int(grp[-1].item()) * 3 == asm.nnz
    worst = 0.0
    for c in range(3):
        cs = torch.zeros(Kb.shape[0] + 1, dtype=torch.float64, device=dev)
        torch.cumsum(Kb[:, c].contiguous(), 0, out=cs[1:])
        worst = max(worst, (cs[grp[1:]] - cs[grp[:-1]]).abs().max().item())
        del cs
    assert worst < 1e-6 * kmax                                                   # translation null space (running-sum rounding)
    del Kb, grp
    f2 = torch.zeros_like(f)
    asm.assemble(st, None, f2, flags=capi.OVERWRITE)                             # f_int = sum w B^t sigma at the Gauss points
    torch.cuda.synchronize()
    assert (f - f2).abs().max().item() < 1e-10 * fmax
    K2 = torch.zeros_like(K); f3 = torch.zeros_like(f)
    asm.assemble(st, K2, f3, flags=capi.ACCUMULATE)
    torch.cuda.synchronize()
    assert (K - K2).abs().max().item() < 1e-12 * kmax and (f - f3).abs().max().item() < 1e-10 * fmax
    K2.zero_()
    asm.assemble(st, K2, f3, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    assert torch.equal(K, K2)
