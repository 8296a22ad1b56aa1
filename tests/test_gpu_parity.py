"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): CSR pattern / DOF indexing bit-exact, K and f_int within 1e-12
relative (per-row max scale, SURVEY.md gotcha 14)."""
import numpy as np
import pytest

from falcon_amd import capi, meshgen
from oracle import pyoracle as po
from tests.conftest import rel_rowmax_err

pytestmark = pytest.mark.gpu
TOL = 1e-12

MESH = {capi.TRI3_G1: lambda: meshgen.grid_tri3(9, 7), capi.QUAD4_G2: lambda: meshgen.grid_quad4(11, 9),
        capi.HEX8_G2: lambda: meshgen.grid_hex8(5, 4, 3),
        capi.TRI3_G3: lambda: meshgen.grid_tri3(8, 9), capi.TRI6_G3: lambda: meshgen.grid_tri6(6, 5),
        capi.QUAD8_G3: lambda: meshgen.grid_quad8(7, 5), capi.QUAD8_G2: lambda: meshgen.grid_quad8(6, 7),
        capi.HEX20_G3: lambda: meshgen.grid_hex20(3, 4, 3), capi.HEX20_G2: lambda: meshgen.grid_hex20(4, 3, 3)}
HIGH_ORDER = [capi.TRI3_G3, capi.TRI6_G3, capi.QUAD8_G3, capi.QUAD8_G2, capi.HEX20_G3, capi.HEX20_G2]


def vec_err(got, ref):
    return np.abs(got - ref).max() / np.abs(ref).max()


def row_scale(row_ptr, K):
    """max |K_row| repeated for every entry of the row: prior contents of K for the FA_ACCUMULATE cases are drawn at the
    scale of the row they sit in, so that subtracting them again costs no digits (the metric is per-row relative)."""
    lens = np.diff(row_ptr)
    rows = np.repeat(np.arange(len(row_ptr) - 1), lens)
    m = np.zeros(len(row_ptr) - 1)
    np.maximum.at(m, rows, np.abs(K))
    return m[rows]


def make_case(shape, model, material, keep=0, frac=capi.AT2, state2d=capi.PLANE_STRAIN, shuffle=False, seed=1,
              sigma_t=0.0, young=210.0, gc=2.7e-3, l0=0.2):
    coords, conn = MESH[shape]()
    linear = shape in (capi.TRI3_G1, capi.TRI3_G3, capi.QUAD4_G2, capi.HEX8_G2)
    coords = meshgen.distort(coords, conn, 0.15 if linear else 0.06, seed)    # quadratic elements: mid nodes at h/2
    if shuffle:
        coords, conn, _ = meshgen.shuffle_nodes(coords, conn, seed)
    rank = coords.shape[1]
    nd = rank + (0 if model == capi.LINEAR_ELASTICITY else 1)
    st = meshgen.smooth_state(coords, nd, seed=seed, amp=5e-3, noise=1e-4, l0=0.3)
    kw = dict(state=state2d, young=young, poisson=0.3, frac_type=frac, gc=gc, l0=l0, sigma_t=sigma_t, beta=6500.0,
              keep_off_diags=keep)
    cfg = po.make_config(shape, model, material, **kw)
    return coords, conn, st, kw, cfg


def ext_model(model):
    return model in (capi.PHASE_FRACTURE_EXT, capi.PHASE_FRACTURE_EXTIT)


def run_both(shape, model, material, *, keep=0, kind=capi.PATTERN_FULL, flags=capi.OVERWRITE, want_K=True, **kws):
    import torch
    coords, conn, st, kw, cfg = make_case(shape, model, material, keep, **kws)
    asm = capi.Assembler(shape, model, material, coords, conn, pattern_kind=kind, **kw)
    rank = coords.shape[1]
    rp, ci = asm.pattern()
    rpo, cio = po.pattern(coords.shape[0], conn, asm.ndofpn, rank, kind, po.keep_code(cfg))
    assert np.array_equal(rp, rpo) and np.array_equal(ci, cio)            # bit-exact pattern / dof indexing
    rng = np.random.default_rng(7)
    nip_t = conn.shape[0] * asm.nip
    H0 = rng.uniform(0, 2e-3, nip_t)
    st0 = st.copy(); st00 = st.copy()
    if asm.ndofpn > rank:
        st0[rank::asm.ndofpn] += rng.uniform(-0.02, 0.02, coords.shape[0])
        st00[rank::asm.ndofpn] += rng.uniform(-0.02, 0.02, coords.shape[0])
    dev = torch.device("cuda", 0)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ref = po.assemble(cfg, coords, conn, rpo, cio, st, state0=st0 if ext_model(model) else None,
                      state00=st00 if ext_model(model) else None,
                      dt=0.7, dt0=0.4, hist_pre=H0 if model == capi.PHASE_FRACTURE else None, want_K=want_K,
                      want_strain=True)
    # what the matrix / vector hold before an FA_ACCUMULATE call ("other models of the MultiModel added first"), at the
    # scale of the row / vector it is added to
    K0 = rng.uniform(-1, 1, asm.nnz) * (row_scale(rp, ref["K"]) if want_K else 1.0)
    f0 = rng.uniform(-1, 1, asm.ndof) * np.abs(ref["f"]).max()
    K = d(K0) if want_K else None
    f = d(f0)
    err = torch.zeros(1, dtype=torch.float64, device=dev)
    if model == capi.PHASE_FRACTURE:
        asm.set_history(0, H0)
    ext = model in (capi.PHASE_FRACTURE_EXT, capi.PHASE_FRACTURE_EXTIT)
    asm.assemble(d(st), K, f, state0=d(st0) if ext else None, state00=d(st00) if ext else None, dt=0.7, dt0=0.4,
                 err_ext=err, flags=flags | capi.STORE_STRAIN)
    torch.cuda.synchronize()
    over = bool(flags & capi.OVERWRITE)
    if want_K:
        got = K.cpu().numpy() - (0 if over else K0)
        assert rel_rowmax_err(rp, got, ref["K"]) < TOL
    gotf = f.cpu().numpy() - (0 if over else f0)
    assert vec_err(gotf, ref["f"]) < TOL
    if model == capi.PHASE_FRACTURE:
        assert vec_err(asm.get_history(1), ref["hist_new"]) < TOL
        assert np.array_equal(asm.get_history(0), H0)
    if model == capi.PHASE_FRACTURE_EXTIT:
        assert abs(err.item() - ref["err_ext"]) < 1e-12 * ref["err_ext"]
    sc = asm.get_strain_cache()
    assert np.abs(sc - ref["strain"]).max() < TOL * np.abs(ref["strain"]).max()
    return asm


@pytest.mark.parametrize("shape", [capi.TRI3_G1, capi.QUAD4_G2, capi.HEX8_G2])
@pytest.mark.parametrize("state2d", [capi.PLANE_STRAIN, capi.PLANE_STRESS])
def test_linear_elasticity(shape, state2d):
    asm = run_both(shape, capi.LINEAR_ELASTICITY, capi.HOOKE, state2d=state2d)
    assert asm.uses_row_kernel


@pytest.mark.parametrize("shape", [capi.TRI3_G1, capi.QUAD4_G2, capi.HEX8_G2])
@pytest.mark.parametrize("material", [capi.AMOR_PHASE, capi.BOURDIN_PHASE, capi.MIEHE_PHASE])
@pytest.mark.parametrize("keep", [0, 1])
def test_phase_fracture(shape, material, keep):
    run_both(shape, capi.PHASE_FRACTURE, material, keep=keep)


@pytest.mark.parametrize("shape", [capi.TRI3_G1, capi.QUAD4_G2, capi.HEX8_G2])
@pytest.mark.parametrize("model", [capi.PHASE_FRACTURE_EXT, capi.PHASE_FRACTURE_EXTIT])
def test_phase_fracture_ext(shape, model):
    run_both(shape, model, capi.AMOR_PHASE)


@pytest.mark.parametrize("shape", [capi.TRI3_G1, capi.QUAD4_G2, capi.HEX8_G2])
def test_miehe_variants(shape):
    """MiehePhase (spectral split) through the Ext / ExtIt models, the ordered row kernel, the generic kernel,
    the WRITTEN pattern on shuffled nodes and the force-only call."""
    run_both(shape, capi.PHASE_FRACTURE_EXT, capi.MIEHE_PHASE)
    run_both(shape, capi.PHASE_FRACTURE_EXTIT, capi.MIEHE_PHASE, flags=capi.OVERWRITE | capi.DETERMINISTIC)
    run_both(shape, capi.PHASE_FRACTURE, capi.MIEHE_PHASE, keep=1, flags=capi.OVERWRITE | capi.FORCE_GENERIC)
    run_both(shape, capi.PHASE_FRACTURE, capi.MIEHE_PHASE, kind=capi.PATTERN_WRITTEN, flags=capi.ACCUMULATE, shuffle=True)
    run_both(shape, capi.PHASE_FRACTURE, capi.MIEHE_PHASE, want_K=False)


@pytest.mark.parametrize("shape", [capi.QUAD4_G2, capi.HEX8_G2])
@pytest.mark.parametrize("field", ["zero", "dilation", "compression", "uniaxial", "shear"])
def test_miehe_degenerate_strains(shape, field):
    """Repeated / zero principal strains (theta_ab branch of MiehePhaseMaterial.cpp:394-401) on an undistorted grid.
    The reference's projector is discontinuous where a principal strain crosses zero (diag[i] = eigVal > 0,
    :372-375), so fields whose principal strains are zero only up to rounding are not comparable between two
    implementations; the 3D fields below keep every principal strain either exactly repeated or away from zero."""
    import torch
    coords, conn = MESH[shape]()
    rank = coords.shape[1]; nd = rank + 1
    st = np.zeros(coords.shape[0] * nd)
    x = coords
    if field == "dilation":
        for i in range(rank): st[i::nd] = 1e-3 * x[:, i]
    elif field == "compression":
        for i in range(rank): st[i::nd] = -1e-3 * x[:, i]
    elif field == "uniaxial":
        st[0::nd] = 2e-3 * x[:, 0]
        for i in range(1, rank): st[i::nd] = (-0.6e-3 if rank == 3 else 0.0) * x[:, i]
    elif field == "shear":
        st[0::nd] = 1e-3 * x[:, 1] + 0.2e-3 * x[:, 0]; st[1::nd] = 1e-3 * x[:, 0]     # tr != 0: heavisideP(tr) is a step too
        if rank == 3: st[2::nd] = 0.3e-3 * x[:, 2]
    st[rank::nd] = 0.3 * np.exp(-np.abs(x[:, 1] - 0.5) / 0.3)
    kw = dict(state=capi.PLANE_STRAIN, young=210.0, poisson=0.3, frac_type=capi.AT2, gc=2.7e-3, l0=0.2, sigma_t=0.0,
              beta=6500.0, keep_off_diags=1)
    cfg = po.make_config(shape, capi.PHASE_FRACTURE, capi.MIEHE_PHASE, **kw)
    asm = capi.Assembler(shape, capi.PHASE_FRACTURE, capi.MIEHE_PHASE, coords, conn, **kw)
    rp, ci = asm.pattern()
    dev = torch.device("cuda", 0)
    K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    asm.assemble(torch.from_numpy(st).to(dev), K, f, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    ref = po.assemble(cfg, coords, conn, rp, ci, st)
    assert rel_rowmax_err(rp, K.cpu().numpy(), ref["K"]) < TOL
    assert np.abs(f.cpu().numpy() - ref["f"]).max() < TOL * max(np.abs(ref["f"]).max(), 1e-300)
    assert vec_err(asm.get_history(1), ref["hist_new"]) < TOL if ref["hist_new"].max() > 0 else True


@pytest.mark.parametrize("frac", [capi.AT1, capi.CZM_LINEAR, capi.CZM_EXPONENTIAL, capi.CZM_CORNELISSEN,
                                  capi.CZM_HYPERBOLIC])
def test_fracture_laws(frac):
    run_both(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, keep=1, frac=frac, sigma_t=3.0, young=3e4,
             gc=0.12, l0=0.25)


@pytest.mark.parametrize("shape", [capi.TRI3_G1, capi.QUAD4_G2, capi.HEX8_G2])
def test_written_pattern_accumulate_shuffled(shape):
    run_both(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, keep=0, kind=capi.PATTERN_WRITTEN, flags=capi.ACCUMULATE,
             shuffle=True)
    run_both(shape, capi.PHASE_FRACTURE_EXT, capi.BOURDIN_PHASE, kind=capi.PATTERN_WRITTEN, shuffle=True)


@pytest.mark.parametrize("shape", [capi.TRI3_G1, capi.QUAD4_G2, capi.HEX8_G2])
@pytest.mark.parametrize("flags", [capi.OVERWRITE, capi.ACCUMULATE])
def test_deterministic_row_kernel(shape, flags):
    """FA_DETERMINISTIC: phased row kernel, every entry summed in ascending element order."""
    run_both(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, keep=1, flags=flags | capi.DETERMINISTIC)
    run_both(shape, capi.PHASE_FRACTURE_EXTIT, capi.BOURDIN_PHASE, flags=flags | capi.DETERMINISTIC, shuffle=True)
    run_both(shape, capi.LINEAR_ELASTICITY, capi.HOOKE, flags=flags | capi.DETERMINISTIC)


@pytest.mark.parametrize("use_patch", [False, True])
def test_many_workgroups(use_patch, monkeypatch):
    """A mesh large enough for hundreds of workgroups with ragged boundaries (shuffled numbering); also the
    opt-in experimental patch kernel (FA_USE_PATCH)."""
    if use_patch:
        monkeypatch.setenv("FA_USE_PATCH", "1")
    coords, conn = meshgen.grid_quad4(150, 131)
    coords = meshgen.distort(coords, conn, 0.2, 3)
    coords, conn, _ = meshgen.shuffle_nodes(coords, conn, 3)
    kw = dict(young=210.0, poisson=0.3, gc=2.7e-3, l0=0.05, keep_off_diags=True)
    asm = capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, **kw)
    if use_patch:
        assert asm.npatch > 100 and 1.0 < asm.redundancy < 2.0
    st = meshgen.smooth_state(coords, 3, l0=0.05)
    cfg = po.make_config(po.QUAD4_G2, po.PHASE_FRACTURE, po.AMOR, **kw)
    rp, ci = asm.pattern()
    K, f, _ = asm.assemble_host(st, np.zeros(asm.nnz), flags=capi.OVERWRITE)
    ref = po.assemble(cfg, coords, conn, rp, ci, st)
    assert rel_rowmax_err(rp, K, ref["K"]) < TOL and vec_err(f, ref["f"]) < TOL
    assert vec_err(asm.get_history(1), ref["hist_new"]) < TOL
    K2, f2, _ = asm.assemble_host(st, np.zeros(asm.nnz), flags=capi.OVERWRITE | capi.DETERMINISTIC)
    K3, f3, _ = asm.assemble_host(st, np.zeros(asm.nnz), flags=capi.OVERWRITE | capi.DETERMINISTIC)
    assert np.array_equal(K2, K3) and np.array_equal(f2, f3)
    assert rel_rowmax_err(rp, K2, ref["K"]) < TOL


@pytest.mark.parametrize("material,nwg", [(capi.AMOR_PHASE, 8), (capi.BOURDIN_PHASE, 24), (capi.AMOR_PHASE, 64)])
def test_persistent_lane_pair_kernel(material, nwg, monkeypatch):
    """FA_PAIR_PERSIST=n: the overwrite GET_MATRIX0 pass of the lane-pair kernel as n workgroups that each walk several
    tiles (the default on meshes of >= 32 k tiles, which only the full-size tests reach): every workgroup count -- more
    tiles than workgroups with a ragged last round, and more workgroups than a round needs -- against the oracle at 1e-12,
    shuffled numbering, and bitwise equal to the one-workgroup-per-tile launch."""
    import torch
    coords, conn = meshgen.grid_quad4(97, 83)                               # 8232 nodes: ~130 tiles
    coords = meshgen.distort(coords, conn, 0.2, 5)
    coords, conn, _ = meshgen.shuffle_nodes(coords, conn, 5)
    kw = dict(young=210.0, poisson=0.3, gc=2.7e-3, l0=0.05)
    st = meshgen.smooth_state(coords, 3, l0=0.05)
    cfg = po.make_config(po.QUAD4_G2, po.PHASE_FRACTURE, material, **kw)
    dev = torch.device("cuda", 0)
    out = []
    for pers in (str(nwg), "0"):
        monkeypatch.setenv("FA_PAIR_PERSIST", pers)
        asm = capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, material, coords, conn, **kw)
        K = torch.full((asm.nnz,), 3.0, dtype=torch.float64, device=dev); f = torch.full((asm.ndof,), 3.0, dtype=torch.float64, device=dev)
        asm.assemble(torch.from_numpy(st).to(dev), K, f, flags=capi.OVERWRITE)
        torch.cuda.synchronize()
        out.append((K.cpu().numpy(), f.cpu().numpy(), asm.get_history(1)))
        rp, ci = asm.pattern()
        asm.close()
    ref = po.assemble(cfg, coords, conn, rp, ci, st)
    K, f, h = out[0]
    assert rel_rowmax_err(rp, K, ref["K"]) < TOL and vec_err(f, ref["f"]) < TOL and vec_err(h, ref["hist_new"]) < TOL
    assert np.array_equal(h, out[1][2])                                     # same arithmetic per Gauss point
    assert rel_rowmax_err(rp, K, out[1][0]) < 1e-14                         # the order of the LDS adds is the waves' own


@pytest.mark.parametrize("shape", [capi.TRI3_G1, capi.QUAD4_G2, capi.HEX8_G2])
def test_force_only(shape):
    run_both(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, want_K=False)
    run_both(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, want_K=False, flags=capi.ACCUMULATE)


@pytest.mark.parametrize("shape", [capi.TRI3_G1, capi.QUAD4_G2, capi.HEX8_G2])
@pytest.mark.parametrize("mode", ["node_tiles", "deterministic"])
def test_force_only_on_node_tiles(shape, mode, monkeypatch):
    """GET_INT_VECTOR passes run on element tiles by default (atomic f rows); FA_ELEM_TILES=0 and FA_DETERMINISTIC keep the
    force-vector kernel on node tiles (ordered sums) -- what FA_PART_* passes run on as well."""
    if mode == "node_tiles":
        monkeypatch.setenv("FA_ELEM_TILES", "0")
    extra = capi.DETERMINISTIC if mode == "deterministic" else 0
    run_both(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, want_K=False, flags=capi.OVERWRITE | extra)
    run_both(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, want_K=False, flags=capi.ACCUMULATE | extra)


@pytest.mark.parametrize("shape", [capi.TRI3_G1, capi.QUAD4_G2, capi.HEX8_G2])
@pytest.mark.parametrize("flags", [capi.OVERWRITE, capi.ACCUMULATE])
def test_generic_kernel_on_canonical_pattern(shape, flags):
    run_both(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, keep=1, flags=flags | capi.FORCE_GENERIC)


def test_generic_kernel_custom_dofs_and_unsorted_pattern():
    """A caller-supplied (Jive-like) dof numbering and a CSR whose rows are not column-sorted."""
    import torch
    shape, model, material = capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE
    coords, conn, st, kw, cfg = make_case(shape, model, material, 1)
    nn = coords.shape[0]; nd = 3
    rp, ci = po.pattern(nn, conn, nd, 2)
    ref = po.assemble(cfg, coords, conn, rp, ci, st)
    rng = np.random.default_rng(11)
    perm = rng.permutation(nn * nd)                  # canonical dof k -> caller's dof perm[k]
    ndof = nn * nd
    # caller's CSR: row perm[r] holds columns perm[ci] in shuffled order
    inv = np.empty(ndof, dtype=np.int64); inv[perm] = np.arange(ndof)
    lens = np.diff(rp)
    rp2 = np.zeros(ndof + 1, dtype=np.int64); rp2[1:] = np.cumsum(lens[inv])
    ci2 = np.empty_like(ci); src = np.empty(ci.size, dtype=np.int64)
    for r2 in range(ndof):
        r = inv[r2]
        idx = np.arange(rp[r], rp[r + 1]); rng.shuffle(idx)
        ci2[rp2[r2]:rp2[r2 + 1]] = perm[ci[idx]]; src[rp2[r2]:rp2[r2 + 1]] = idx
    asm = capi.Assembler(shape, model, material, coords, conn, row_ptr=rp2, col_idx=ci2,
                         node_dofs=perm.astype(np.int64), ndof=ndof, **kw)
    assert not asm.uses_row_kernel
    st2 = np.empty_like(st); st2[perm] = st
    K, f, _ = asm.assemble_host(st2, np.zeros(asm.nnz), flags=capi.OVERWRITE)
    assert rel_rowmax_err(rp2, K, ref["K"][src]) < TOL
    assert vec_err(f[perm], ref["f"]) < TOL


def test_history_commit_cycle_and_host_path():
    """COMMIT swaps newHist_/preHist_ (PhaseFractureModel.cpp:468-473); a second load step sees the first
    step's Psi as H0; loading flag only matters for K3."""
    shape, model, material = capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.BOURDIN_PHASE
    coords, conn, st, kw, cfg = make_case(shape, model, material, 1)
    asm = capi.Assembler(shape, model, material, coords, conn, **kw)
    rp, ci = asm.pattern()
    K1, f1, _ = asm.assemble_host(st, np.zeros(asm.nnz), flags=capi.OVERWRITE)
    r1 = po.assemble(cfg, coords, conn, rp, ci, st)
    assert rel_rowmax_err(rp, K1, r1["K"]) < TOL
    asm.commit()
    assert vec_err(asm.get_history(0), r1["hist_new"]) < TOL
    st2 = st.copy(); st2[0::3] *= 0.5; st2[1::3] *= 1.5          # part unloads, part loads further
    r2 = po.assemble(cfg, coords, conn, rp, ci, st2, hist_pre=r1["hist_new"])
    K20 = 0.5 * row_scale(rp, r2["K"]); f20 = np.full(asm.ndof, 0.5 * np.abs(r2["f"]).max())
    K2 = K20.copy(); f2 = f20.copy()
    asm.assemble_host(st2, K2, f2, flags=capi.ACCUMULATE)
    assert rel_rowmax_err(rp, K2 - K20, r2["K"]) < TOL
    assert vec_err(f2 - f20, r2["f"]) < TOL
    assert vec_err(asm.get_history(1), r2["hist_new"]) < TOL
    # force-only call rewrites newHist_ too (gotcha 9)
    asm.set_history(1, np.zeros(asm.nip_total))
    asm.assemble_host(st2, None, np.zeros(asm.ndof), flags=capi.OVERWRITE)
    assert vec_err(asm.get_history(1), r2["hist_new"]) < TOL


def test_sm01_as_shipped():
    """Config 1: tests/sm01 (2 Triangle3, Hooke plane stress) -- closed form of SURVEY.md 8c KAT-1."""
    coords = np.array([[0., 0.], [1., 0.], [1., 1.], [0., 1.]])
    conn = np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32)
    asm = capi.Assembler(capi.TRI3_G1, capi.LINEAR_ELASTICITY, capi.HOOKE, coords, conn, state=capi.PLANE_STRESS,
                         young=1e4, poisson=0.3)
    assert asm.nnz == 56
    u = np.zeros(8); u[5] = u[7] = 1e-4
    K, f, _ = asm.assemble_host(u, np.zeros(56), flags=capi.OVERWRITE)
    rp, ci = asm.pattern()
    a = 1e4 / (1 - 0.09); G = a * 0.35
    row0 = dict(zip(ci[rp[0]:rp[1]], K[rp[0]:rp[1]]))
    exp = {0: (a + G) / 2, 1: 0, 2: -a / 2, 3: a * 0.3 / 2, 4: 0, 5: -(a * 0.3 + G) / 2, 6: -G / 2, 7: G / 2}
    for c, v in exp.items():
        assert abs(row0[c] - v) < 1e-12 * a
    assert abs(f[5] + f[7] - 1.0989010989010989) < 1e-13


def test_empty_and_single_element():
    coords = np.array([[0., 0.], [2., 0.], [0., 1.]])
    conn = np.array([[0, 1, 2]], dtype=np.int32)
    asm = capi.Assembler(capi.TRI3_G1, capi.LINEAR_ELASTICITY, capi.HOOKE, coords, conn, young=10.0)
    cfg = po.make_config(po.TRI3_G1, po.LINEAR_ELASTICITY, po.HOOKE, young=10.0)
    rp, ci = asm.pattern()
    u = np.arange(6) * 1e-3
    K, f, _ = asm.assemble_host(u, np.zeros(asm.nnz), flags=capi.OVERWRITE)
    ref = po.assemble(cfg, coords, conn, rp, ci, u)
    assert rel_rowmax_err(rp, K, ref["K"]) < TOL
    asm0 = capi.Assembler(capi.TRI3_G1, capi.LINEAR_ELASTICITY, capi.HOOKE, coords, np.zeros((0, 3), np.int32))
    assert asm0.nnz == 0
    K, f, _ = asm0.assemble_host(u, np.zeros(0), np.ones(6), flags=capi.OVERWRITE)
    assert np.all(f == 0)


# ---- higher-order shapes (cooperative element kernel) ----
@pytest.mark.parametrize("shape", HIGH_ORDER)
def test_high_order_linear_elasticity(shape):
    """BASELINE config 5 family: LinearElasticity + Hooke on Hex20 (Gauss3^3 / Gauss2^3), and the 2D shapes."""
    asm = run_both(shape, capi.LINEAR_ELASTICITY, capi.HOOKE)
    assert not asm.uses_row_kernel
    run_both(shape, capi.LINEAR_ELASTICITY, capi.HOOKE, state2d=capi.PLANE_STRESS, flags=capi.ACCUMULATE, shuffle=True)


@pytest.mark.parametrize("shape", HIGH_ORDER)
@pytest.mark.parametrize("material", [capi.AMOR_PHASE, capi.BOURDIN_PHASE, capi.MIEHE_PHASE])
def test_high_order_phase_fracture(shape, material):
    run_both(shape, capi.PHASE_FRACTURE, material, keep=0)
    run_both(shape, capi.PHASE_FRACTURE, material, keep=1, kind=capi.PATTERN_WRITTEN, shuffle=True)
    run_both(shape, capi.PHASE_FRACTURE, material, want_K=False)


@pytest.mark.parametrize("shape", HIGH_ORDER)
def test_high_order_ext_models_and_column_search(shape):
    run_both(shape, capi.PHASE_FRACTURE_EXT, capi.AMOR_PHASE)
    run_both(shape, capi.PHASE_FRACTURE_EXTIT, capi.MIEHE_PHASE, flags=capi.ACCUMULATE)
    run_both(shape, capi.PHASE_FRACTURE, capi.AMOR_PHASE, keep=1, flags=capi.OVERWRITE | capi.FORCE_GENERIC)


def test_full_size_invariants_cfg3():
    """BASELINE config 3 at its full size (4000 x 4000 Q4, u-phi, AT2, Amor): too large for the oracle, so the result
    is checked through size-independent properties of the discretisation:
      * sum_a B_a = 0 (partition of unity)  =>  the internal forces of every displacement component sum to zero,
        and a rigid translation is in the null space of the u-u block: every u row of K sums to zero over the
        u columns of the same component pair;
      * K4 row sums: sum_b K_phi_phi[a,b] = int N_a (gc/l0 * ddw/cw... + g'' Psi) dV > 0 and symmetric u-u block;
      * the force-only call returns the same f_int as the matrix call, and the history is written identically;
      * FA_ACCUMULATE on a zeroed K equals FA_OVERWRITE."""
    import torch
    n = 4000
    coords, conn = meshgen.grid_quad4(n, n)
    kw = dict(young=210e3, poisson=0.3, gc=2.7, l0=0.015, frac_type=capi.AT2, keep_off_diags=False)
    asm = capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, **kw)
    dev = torch.device("cuda", 0)
    st = torch.from_numpy(meshgen.smooth_state(coords, 3)).to(dev)
    del coords
    K = torch.empty(asm.nnz, dtype=torch.float64, device=dev).fill_(float("nan"))
    f = torch.empty(asm.ndof, dtype=torch.float64, device=dev).fill_(float("nan"))
    asm.assemble(st, K, f, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    assert not torch.isnan(K).any() and not torch.isnan(f).any()                 # every entry of the pattern was written
    fmax = f.abs().max().item()
    for c in range(2):
        assert abs(f[c::3].sum().item()) < 1e-9 * fmax * np.sqrt(asm.ndof)
    # row sums per column type: the pattern is blocked [neighbour][type], 3 entries per neighbour in every row
    Kb = K.view(-1, 3)                                                           # (row-entry groups, column type)
    rp = torch.from_numpy(asm.pattern()[0]).to(dev)
    grp = rp // 3
    cs = torch.zeros(Kb.shape[0] + 1, 3, dtype=torch.float64, device=dev)
    torch.cumsum(Kb, 0, out=cs[1:])
    rowsum = cs[grp[1:]] - cs[grp[:-1]]                                          # (ndof, 3): sum over neighbours per column type
    kmax = K.abs().max().item()
    urows = torch.arange(asm.ndof, device=dev) % 3 < 2
    assert rowsum[urows][:, :2].abs().max().item() < 1e-7 * kmax                 # translation null space (cumsum rounding)
    assert rowsum[urows][:, 2].abs().max().item() == 0.0                         # u-phi block not written: exact zeros
    assert rowsum[~urows][:, :2].abs().max().item() == 0.0                       # phi-u block not written
    assert rowsum[~urows][:, 2].min().item() > 0.0                               # K4 row sums are positive
    del cs, rowsum, Kb
    h1 = asm.get_history(1).copy()
    f2 = torch.zeros_like(f)
    asm.assemble(st, None, f2, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    assert torch.equal(f, f2) or (f - f2).abs().max().item() < 1e-13 * fmax
    assert np.array_equal(asm.get_history(1), h1)
    K2 = torch.zeros_like(K); f3 = torch.zeros_like(f)
    asm.assemble(st, K2, f3, flags=capi.ACCUMULATE)
    torch.cuda.synchronize()
    assert (K - K2).abs().max().item() < 1e-12 * kmax and (f - f3).abs().max().item() < 1e-12 * fmax


# ---- micromorphic variant MicroPhaseFracture ----
ALL_SHAPES = [capi.TRI3_G1, capi.QUAD4_G2, capi.HEX8_G2] + HIGH_ORDER


def run_micro(shape, material, keep, *, flags=capi.OVERWRITE, want_K=True, commit=False, frac=capi.AT2, sigma_t=0.0,
              young=210.0, gc=2.7e-3, l0=0.2, tol=TOL, model=capi.MICRO_PHASE_FRACTURE, beta=250.0, state_mod=None,
              tol_K=None):
    import torch
    coords, conn, st, kw, cfg = make_case(shape, model, material, keep, frac=frac, sigma_t=sigma_t,
                                          young=young, gc=gc, l0=l0)
    kw["beta"] = beta
    if state_mod is not None:
        st = state_mod(coords, conn, st.copy())
    cfg = po.make_config(shape, model, material, **kw)
    asm = capi.Assembler(shape, model, material, coords, conn, **kw)
    rp, ci = asm.pattern()
    rpo, cio = po.pattern(coords.shape[0], conn, asm.ndofpn, coords.shape[1], po.PATTERN_FULL, po.keep_code(cfg))
    assert np.array_equal(rp, rpo) and np.array_equal(ci, cio)
    rng = np.random.default_rng(5)
    d0 = rng.uniform(0.0, 0.4, conn.shape[0] * asm.nip)           # committed local damage: some points load, some do not
    asm.set_history(0, d0)
    dev = torch.device("cuda", 0)
    K = torch.full((asm.nnz,), 3.0, dtype=torch.float64, device=dev) if want_K else None
    f = torch.full((asm.ndof,), -2.0, dtype=torch.float64, device=dev)
    ext = model != capi.MICRO_PHASE_FRACTURE
    stx = None
    if ext:                                                       # stateExt_: the caller's extrapolation, some of it outside [0,1]
        stx = st.copy()
        stx[asm.ndofpn - 1::asm.ndofpn] += rng.uniform(-0.05, 0.05, coords.shape[0])
        stx[asm.ndofpn - 1::asm.ndofpn][::7] = -0.01
        stx[asm.ndofpn - 1::asm.ndofpn][3::11] = 1.02
    err = torch.full((1,), 7.0, dtype=torch.float64, device=dev) if model == capi.MICRO_PHASE_FRACTURE_EXTIT else None
    asm.assemble(torch.from_numpy(st).to(dev), K, f, state00=None if stx is None else torch.from_numpy(stx).to(dev),
                 err_ext=err, flags=flags | capi.STORE_STRAIN)
    torch.cuda.synchronize()
    ref = po.assemble(cfg, coords, conn, rp, ci, st, state00=stx, hist_pre=d0, want_K=want_K, want_strain=True)
    assert ref["micro_stale_hits"] == 0 and asm.micro_stale_hits() == 0
    loading = (ref["hist_new"] > d0).mean()
    assert 0.05 < loading < 0.95                                  # both branches of :738-782 are exercised
    if want_K:
        assert rel_rowmax_err(rp, K.cpu().numpy(), ref["K"]) < (tol if tol_K is None else tol_K)
    assert vec_err(f.cpu().numpy(), ref["f"]) < tol
    assert np.abs(asm.get_history(1) - ref["hist_new"]).max() < tol
    assert np.abs(asm.get_strain_cache() - ref["strain"]).max() < tol * np.abs(ref["strain"]).max()
    if err is not None:
        assert ref["err_ext"] > 0 and abs(err.item() - ref["err_ext"]) < tol * ref["err_ext"]
    if commit:                                                    # COMMIT: newHist_.phasef_.swap(preHist_.phasef_) (:458-463)
        asm.commit()
        assert np.abs(asm.get_history(0) - ref["hist_new"]).max() < tol
    return asm


@pytest.mark.parametrize("shape", ALL_SHAPES)
@pytest.mark.parametrize("material", [capi.BOURDIN_PHASE, capi.MIEHE_PHASE])
def test_micro_phase_fracture(shape, material):
    run_micro(shape, material, keep=0)
    run_micro(shape, material, keep=1, commit=True)
    run_micro(shape, material, keep=0, want_K=False)


# czmHyperbolic (p = 4) at this state is nearly fully degraded: the consistent tangent g DP + DN - g'^2 / J sigma+ (x) sigma+
# (MicroPhaseFractureModel.cpp:804-816) is the difference of two terms that almost cancel, so K1 is ill-conditioned with
# respect to one-ulp changes upstream (Psi from MiehePhase's projector).  Measured, not guessed: the checker compiled with
# and without FMA contraction differs from ITSELF by 2.2e-12 on exactly this case and by < 6e-13 on the two others
# (tests/test_oracle_kat.py::test_micro_law_conditioning pins those numbers on the CPU), so 1e-12 is not a bar any two
# compilations of the reference's own source would meet there; f_int, the damage and the strain cache are held to 1e-12.
MICRO_LAW_TOL_K = {capi.AT1: TOL, capi.CZM_EXPONENTIAL: TOL, capi.CZM_HYPERBOLIC: 5e-12}


@pytest.mark.parametrize("frac", [capi.AT1, capi.CZM_EXPONENTIAL, capi.CZM_HYPERBOLIC])
def test_micro_fracture_laws(frac):
    """The local Newton iteration runs through (1-d)^p with p = 2.5 / 4: ::pow in the reference
    (MicroPhaseFractureModel.cpp:708-715), correctly rounded double-double powers on the device -- with x*x*sqrt(x) the
    czmExponential law missed 1e-12 (round 2 asserted 1e-11 for all three laws)."""
    run_micro(capi.QUAD4_G2, capi.MIEHE_PHASE, keep=1, frac=frac, sigma_t=3.0, young=3e4, gc=0.12, l0=0.25,
              tol_K=MICRO_LAW_TOL_K[frac])


def test_micro_stale_branch_is_counted():
    """A committed local damage d0 > 1 (unphysical) sends points into the `d > 1` branch of MicroPhaseFractureModel.cpp:772-776,
    where the reference keeps a dphidStrain from an earlier Gauss point and the kernels use 0: the divergence must not be
    silent.  The device counter equals the checker's count of such points; f_int, history and the strain cache (which do
    not depend on dphidStrain) still agree."""
    import torch
    shape, material = capi.QUAD4_G2, capi.BOURDIN_PHASE
    coords, conn, st, kw, cfg = make_case(shape, capi.MICRO_PHASE_FRACTURE, material, 1)
    kw["beta"] = 250.0
    cfg = po.make_config(shape, capi.MICRO_PHASE_FRACTURE, material, **kw)
    asm = capi.Assembler(shape, capi.MICRO_PHASE_FRACTURE, material, coords, conn, **kw)
    rp, ci = asm.pattern()
    rng = np.random.default_rng(9)
    d0 = rng.uniform(0.0, 0.4, conn.shape[0] * asm.nip)
    st = st.copy(); st[2::3] = 1.3                                  # micromorphic variable above 1 drives d above 1 ...
    d0[::5] = 2.0                                                   # ... and a committed d0 = 2 keeps `d > d0` false there
    asm.set_history(0, d0)
    dev = torch.device("cuda", 0)
    K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    asm.assemble(torch.from_numpy(st).to(dev), K, f, flags=capi.OVERWRITE)
    ref = po.assemble(cfg, coords, conn, rp, ci, st, hist_pre=d0)
    assert ref["micro_stale_hits"] > 0
    assert asm.micro_stale_hits() == ref["micro_stale_hits"]
    assert vec_err(f.cpu().numpy(), ref["f"]) < TOL
    assert np.abs(asm.get_history(1) - ref["hist_new"]).max() < TOL
    asm.assemble(torch.from_numpy(st).to(dev), None, f, flags=capi.OVERWRITE)     # the force-only pass counts too
    assert asm.micro_stale_hits() == ref["micro_stale_hits"]


def _expand_leading_elements(frac, amount):
    """state_mod: a uniform expansion added to the displacements of the nodes of the first `frac` of the elements, so that
    the serial loop meets its first compression point late (or never, frac = 1)."""
    def mod(coords, conn, st):
        rank = coords.shape[1]; nd = rank + 1
        nodes = np.unique(conn[:max(1, int(frac * conn.shape[0]))])
        S = st.reshape(-1, nd)
        if frac >= 1.0:
            S[:, :rank] = 0.0
        S[nodes, :rank] += amount * coords[nodes]
        return S.ravel()
    return mod


@pytest.mark.parametrize("shape", ALL_SHAPES)
def test_micro_amor(shape):
    """AmorPhase's split update (AmorPhaseMaterial.cpp:239-271) writes stiffN = K PVol at compression points only, into an
    array the model never resets (MicroPhaseFractureModel.cpp:535): a point in tension sees K PVol once the serial loop has
    passed a compression point, the array's initial contents (zero assumed by the checker and the library) before.  The
    library finds the first compression point of the call with a pre-pass; three states: mixed from the start, compression
    only after the first 60 % of the elements, none at all."""
    run_micro(shape, capi.AMOR_PHASE, keep=1, commit=True)
    run_micro(shape, capi.AMOR_PHASE, keep=0, want_K=False)
    run_micro(shape, capi.AMOR_PHASE, keep=0, state_mod=_expand_leading_elements(0.6, 0.02))
    run_micro(shape, capi.AMOR_PHASE, keep=1, state_mod=_expand_leading_elements(1.0, 0.01))


def _shear_state(trace):
    """state_mod: simple shear u_x = gamma y plus a volumetric part of size `trace` whose sign changes across the mesh
    (a kink in the last displacement component): trace(strain) is +2 trace/rank ... -trace/rank at the Gauss points --
    or zero up to rounding everywhere for trace = 0."""
    def mod(coords, conn, st):
        rank = coords.shape[1]; nd = rank + 1
        S = st.reshape(-1, nd)
        S[:, :rank] = 0.0
        S[:, 0] = 2e-3 * coords[:, 1]
        for i in range(rank - 1):
            S[:, i] += (trace / rank) * coords[:, i]
        S[:, rank - 1] += (3.0 * trace / rank) * np.abs(coords[:, rank - 1] - 0.45)
        return S.ravel()
    return mod


@pytest.mark.parametrize("shape", [capi.QUAD4_G2, capi.HEX8_G2])
def test_micro_amor_near_isochoric(shape):
    """the call-wide decision -- which Gauss point is the first in compression -- is made ONCE, by the
    pre-pass, for every kernel path.
    (i) |tr| = 1e-9 -- tiny against the shear 2e-3 but far above rounding, so the checker classifies alike: every path
    against the checker at 1e-12.  (ii) tr = 0 up to rounding (simple shear on a distorted mesh): the sign of tr is
    noise there and no two implementations need agree.  By default the device paths agree with each other on f_int and
    the history (the stress does not see the sign of a vanishing trace); their tangents may differ at such points, as
    those of the plain AmorPhase kernels do -- the tangent jumps by O(K) across tr = 0 in the reference itself.  With
    FA_MICRO_AMOR_MASK=1 the pre-pass classifies every point and all paths agree on K as well."""
    import os
    import torch
    run_micro(shape, capi.AMOR_PHASE, keep=1, state_mod=_shear_state(1e-9))
    run_micro(shape, capi.AMOR_PHASE, keep=0, want_K=False, state_mod=_shear_state(1e-9))
    run_micro(shape, capi.AMOR_PHASE, keep=0, flags=capi.OVERWRITE | capi.DETERMINISTIC, state_mod=_shear_state(1e-9))
    run_micro(shape, capi.AMOR_PHASE, keep=0, flags=capi.OVERWRITE | capi.FORCE_GENERIC, state_mod=_shear_state(-1e-9))
    coords, conn, st, kw, cfg = make_case(shape, capi.MICRO_PHASE_FRACTURE, capi.AMOR_PHASE, 0)
    kw["beta"] = 250.0
    st = _shear_state(0.0)(coords, conn, st.copy())
    dev = torch.device("cuda", 0)
    sd = torch.from_numpy(st).to(dev)
    for strict in (False, True):
        if strict:
            os.environ["FA_MICRO_AMOR_MASK"] = "1"
        try:
            asm = capi.Assembler(shape, capi.MICRO_PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, **kw)
        finally:
            os.environ.pop("FA_MICRO_AMOR_MASK", None)
        rp, _ = asm.pattern()
        out = {}
        for name, fl, wk in (("default", capi.OVERWRITE, True), ("ordered", capi.OVERWRITE | capi.DETERMINISTIC, True),
                             ("generic", capi.OVERWRITE | capi.FORCE_GENERIC, True), ("force", capi.OVERWRITE, False)):
            K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev) if wk else None
            f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
            asm.assemble(sd, K, f, flags=fl)
            torch.cuda.synchronize()
            out[name] = (None if K is None else K.cpu().numpy(), f.cpu().numpy(), asm.get_history(1))
        K0, f0, h0 = out["default"]
        for name in ("ordered", "generic", "force"):
            K, f, h = out[name]
            if K is not None and strict:
                assert rel_rowmax_err(rp, K, K0) < TOL, name
            assert vec_err(f, f0) < TOL and np.abs(h - h0).max() < TOL, (name, strict)
        asm.close()


@pytest.mark.parametrize("model", [capi.MICRO_PHASE_FRACTURE_EXT, capi.MICRO_PHASE_FRACTURE_EXTIT])
def test_micro_ext_amor(model):
    for shape in (capi.QUAD4_G2, capi.HEX8_G2, capi.TRI6_G3):
        run_micro(shape, capi.AMOR_PHASE, keep=0, model=model, commit=True, state_mod=_expand_leading_elements(0.3, 0.02))
    run_micro(capi.QUAD4_G2, capi.AMOR_PHASE, keep=0, model=model, want_K=False, flags=capi.OVERWRITE | capi.DETERMINISTIC)
    run_micro(capi.TRI3_G1, capi.AMOR_PHASE, keep=0, model=model, flags=capi.OVERWRITE | capi.FORCE_GENERIC)


# ---- MicroPhaseFractureExt / MicroPhaseFractureExtIt ----
@pytest.mark.parametrize("shape", ALL_SHAPES)
@pytest.mark.parametrize("model", [capi.MICRO_PHASE_FRACTURE_EXT, capi.MICRO_PHASE_FRACTURE_EXTIT])
def test_micro_ext_variants(shape, model):
    """Two local Newton solves per Gauss point (MicroPhaseFractureExtModel.cpp:728-850): the first, for the
    extrapolated micromorphic variable (stateExt_, handed over as state00), gives the degradation and the history;
    the second, for the current one, gives the K3 / K4 / f2 coefficients.  No K2 block (pattern keep code 2)."""
    material = capi.MIEHE_PHASE if shape in (capi.QUAD4_G2, capi.TRI6_G3, capi.HEX20_G2) else capi.BOURDIN_PHASE
    run_micro(shape, material, keep=0, model=model, commit=True)
    run_micro(shape, material, keep=0, model=model, want_K=False, beta=6500.0)


@pytest.mark.parametrize("material", [capi.BOURDIN_PHASE, capi.MIEHE_PHASE])
def test_micro_ext_quad4_both_materials(material):
    run_micro(capi.QUAD4_G2, material, keep=0, model=capi.MICRO_PHASE_FRACTURE_EXTIT, beta=6500.0)
    run_micro(capi.HEX8_G2, material, keep=0, model=capi.MICRO_PHASE_FRACTURE_EXT, flags=capi.OVERWRITE | capi.DETERMINISTIC)


def test_micro_ext_needs_state_ext():
    import torch
    coords, conn = meshgen.grid_quad4(3, 3)
    asm = capi.Assembler(capi.QUAD4_G2, capi.MICRO_PHASE_FRACTURE_EXT, capi.MIEHE_PHASE, coords, conn)
    dev = torch.device("cuda", 0)
    st = torch.zeros(asm.ndof, dtype=torch.float64, device=dev); f = torch.zeros_like(st)
    with pytest.raises(capi.FaError, match="stateExt_"):
        asm.assemble(st, None, f, flags=capi.OVERWRITE)


# ---- boundary behaviour ----
def test_pattern_with_a_missing_entry_is_refused():
    """A caller-supplied pattern that lacks an entry the model writes is FA_ERR_ARG at fa_create (the reference's
    MatrixBuilder::addBlock would add it or throw); the same pattern with every entry assembles on the generic kernel."""
    shape, model, material = capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE
    coords, conn, st, kw, cfg = make_case(shape, model, material, 1)
    rp, ci = po.pattern(coords.shape[0], conn, 3, 2)
    rows = [list(ci[rp[r]:rp[r + 1]][::-1]) for r in range(len(rp) - 1)]          # reversed rows: not a canonical layout

    def build(rows):
        rp2 = np.zeros(len(rows) + 1, dtype=np.int64); rp2[1:] = np.cumsum([len(x) for x in rows])
        return rp2, np.array([c for row in rows for c in row], dtype=np.int32)
    rp2, ci2 = build(rows)
    asm = capi.Assembler(shape, model, material, coords, conn, row_ptr=rp2, col_idx=ci2, ndof=len(rp2) - 1, **kw)
    assert not asm.uses_row_kernel
    K, f, _ = asm.assemble_host(st, np.zeros(asm.nnz), flags=capi.OVERWRITE)
    ref = po.assemble(cfg, coords, conn, rp, ci, st)
    src = np.concatenate([np.arange(rp[r], rp[r + 1])[::-1] for r in range(len(rp) - 1)])
    assert rel_rowmax_err(rp2, K, ref["K"][src]) < TOL and vec_err(f, ref["f"]) < TOL
    del rows[40][3]
    rp3, ci3 = build(rows)
    with pytest.raises(capi.FaError, match="has no entry"):
        capi.Assembler(shape, model, material, coords, conn, row_ptr=rp3, col_idx=ci3, ndof=len(rp3) - 1, **kw)


@pytest.mark.parametrize("shape", [capi.QUAD8_G2, capi.HEX20_G2])
def test_deterministic_is_refused_without_a_row_kernel(shape):
    """FA_DETERMINISTIC is only honoured by the ordered row kernel; the other paths say so instead of ignoring it."""
    import torch
    coords, conn, st, kw, cfg = make_case(shape, capi.LINEAR_ELASTICITY, capi.HOOKE)
    asm = capi.Assembler(shape, capi.LINEAR_ELASTICITY, capi.HOOKE, coords, conn, **kw)
    dev = torch.device("cuda", 0)
    K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    with pytest.raises(capi.FaError, match="FA_DETERMINISTIC"):
        asm.assemble(torch.from_numpy(st).to(dev), K, f, flags=capi.OVERWRITE | capi.DETERMINISTIC)


@pytest.mark.parametrize("shape,model,material,keep", [
    (capi.HEX8_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, 1),          # Hex8 row kernel (default: tile kernel)
    (capi.HEX8_G2, capi.PHASE_FRACTURE_EXTIT, capi.BOURDIN_PHASE, 0),
    (capi.QUAD8_G3, capi.PHASE_FRACTURE, capi.AMOR_PHASE, 1),         # cooperative kernel (default: tile kernel)
    (capi.TRI6_G3, capi.PHASE_FRACTURE_EXTIT, capi.MIEHE_PHASE, 0),
    (capi.HEX20_G2, capi.LINEAR_ELASTICITY, capi.HOOKE, 0),
    (capi.TRI3_G3, capi.PHASE_FRACTURE, capi.BOURDIN_PHASE, 0)])
def test_non_default_kernel_paths(shape, model, material, keep, monkeypatch):
    """FA_TILES=0 (read per fa_create): the kernels that serve a shape when the tile kernel is off -- what a caller's
    non-canonical pattern gets on the higher-order shapes -- run inside the default test invocation."""
    monkeypatch.setenv("FA_TILES", "0")
    run_both(shape, model, material, keep=keep)
    run_both(shape, model, material, keep=keep, flags=capi.ACCUMULATE, kind=capi.PATTERN_WRITTEN, shuffle=True)
    run_both(shape, model, material, keep=keep, want_K=False)


@pytest.mark.parametrize("shape,model,material", [
    (capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE),
    (capi.QUAD4_G2, capi.LINEAR_ELASTICITY, capi.HOOKE),
    (capi.TRI3_G1, capi.MICRO_PHASE_FRACTURE, capi.BOURDIN_PHASE)])
def test_persistent_force_vector_kernel(shape, model, material, monkeypatch):
    """FA_PERSIST=1 (opt-in): the persistent LDS-DMA form of the force-vector kernel; FA_FVEC=0: the row kernel's
    GET_INT_VECTOR pass that it replaced."""
    monkeypatch.setenv("FA_PERSIST", "1")
    run_both(shape, model, material, want_K=False)
    run_both(shape, model, material, want_K=False, flags=capi.ACCUMULATE, shuffle=True)
    monkeypatch.delenv("FA_PERSIST")
    monkeypatch.setenv("FA_FVEC", "0")
    run_both(shape, model, material, want_K=False)


def test_cooperative_kernel_on_a_callers_pattern():
    """Quad8 with a caller-supplied (reversed-row) pattern: the cooperative kernel with per-entry column search."""
    shape, model, material = capi.QUAD8_G3, capi.PHASE_FRACTURE, capi.AMOR_PHASE
    coords, conn, st, kw, cfg = make_case(shape, model, material, 1)
    rp, ci = po.pattern(coords.shape[0], conn, 3, 2)
    ci2 = np.concatenate([ci[rp[r]:rp[r + 1]][::-1] for r in range(len(rp) - 1)]).astype(np.int32)
    src = np.concatenate([np.arange(rp[r], rp[r + 1])[::-1] for r in range(len(rp) - 1)])
    asm = capi.Assembler(shape, model, material, coords, conn, row_ptr=rp, col_idx=ci2, ndof=len(rp) - 1, **kw)
    K, f, _ = asm.assemble_host(st, np.zeros(asm.nnz), flags=capi.OVERWRITE)
    ref = po.assemble(cfg, coords, conn, rp, ci, st)
    assert rel_rowmax_err(rp, K, ref["K"][src]) < TOL and vec_err(f, ref["f"]) < TOL


def test_host_getters_wait_for_a_side_stream():
    """fa_assemble on a non-blocking side stream followed at once by the blocking host getters."""
    import torch
    shape, model, material = capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE
    coords, conn = meshgen.grid_quad4(300, 300)
    kw = dict(young=210.0, poisson=0.3, gc=2.7e-3, l0=0.05)
    asm = capi.Assembler(shape, model, material, coords, conn, **kw)
    dev = torch.device("cuda", 0)
    st = torch.from_numpy(meshgen.smooth_state(coords, 3, l0=0.05)).to(dev)
    K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    asm.assemble(st, K, f, flags=capi.OVERWRITE | capi.STORE_STRAIN, stream=side.cuda_stream)
    h1 = asm.get_history(1); s1 = asm.get_strain_cache()
    torch.cuda.synchronize()
    assert np.array_equal(h1, asm.get_history(1)) and np.array_equal(s1, asm.get_strain_cache())
    assert h1.max() > 0
