"""GPU: hand-off of the device-resident K to a solver and the residual evaluation (N3):
fa_pattern_device32, fa_cons_eliminate / eval_masters / eval_slaves, fa_residual -- against the oracle's restatement
(oracle/falcon_oracle_cons.c), through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from falcon_amd import capi, meshgen
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

KW = dict(young=210.0, poisson=0.3, gc=2.7e-3, l0=0.2)


def edge_nodes(coords, axis, side):
    v = coords[:, axis]
    return np.nonzero(np.abs(v - (v.max() if side else v.min())) < 1e-12)[0]


def dirichlet_case(coords, nd):
    bot, top = edge_nodes(coords, 1, 0), edge_nodes(coords, 1, 1)
    slave = np.concatenate([bot * nd, bot * nd + 1, top * nd + 1])
    rval = np.concatenate([np.zeros(2 * bot.size), np.full(top.size, 0.01)])
    return dict(slave=slave, rval=rval)


def periodic_case(coords, nd):
    """right edge = left edge (+ 0.02 on dx), node by node: PeriodicModel.cpp:287 addConstraint(idof, rval, jdof, coef)"""
    left, right = edge_nodes(coords, 0, 0), edge_nodes(coords, 0, 1)
    left = left[np.argsort(coords[left, 1])]; right = right[np.argsort(coords[right, 1])]
    slave = np.concatenate([right * nd + t for t in range(nd)])
    mdof = np.concatenate([left * nd + t for t in range(nd)])
    rval = np.concatenate([np.full(right.size, 0.02)] + [np.zeros(right.size)] * (nd - 1))
    return dict(slave=slave, rval=rval, mptr=np.arange(slave.size + 1), mdof=mdof, mcoef=np.ones(slave.size))


def tied_case(coords, nd):
    """a few interior nodes tied to the mean of two others, all top-edge dy tied to one master dof (DispArclenBCs.cpp:440)"""
    top = edge_nodes(coords, 1, 1)
    master = top[0] * nd + 1
    slave = list(top[1:] * nd + 1); mptr = [0]; mdof = []; mcoef = []
    for _ in slave:
        mdof.append(master); mcoef.append(1.0); mptr.append(len(mdof))
    n = coords.shape[0]
    for node, (a, b) in [(n // 2, (n // 2 - 1, n // 2 + 1)), (n // 3, (n // 3 - 1, n // 3 + 1))]:
        for t in range(nd):
            slave.append(node * nd + t); mdof += [a * nd + t, b * nd + t]; mcoef += [0.5, 0.5]; mptr.append(len(mdof))
    return dict(slave=np.array(slave), rval=None, mptr=np.array(mptr), mdof=np.array(mdof), mcoef=np.array(mcoef))


def extended_pattern(rp, ci, cs):
    """pattern of T^t K T + I_S united with the one of K"""
    n = rp.size - 1
    P = sp.csr_matrix((np.ones(ci.size), ci, rp), shape=(n, n))
    rows = list(range(n)); cols = list(range(n))
    for k, s in enumerate(cs["slave"]):
        for q in range(cs["mptr"][k], cs["mptr"][k + 1]):
            rows.append(s); cols.append(cs["mdof"][q])
    Ts = sp.csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, n))
    E = (P + P @ Ts + Ts.T @ P + Ts.T @ P @ Ts).tocsr()
    E.sort_indices()
    return E.indptr.astype(np.int64), E.indices.astype(np.int32)


def setup(case, need_fill):
    import torch
    coords, conn = meshgen.grid_quad4(14, 9)
    coords = meshgen.distort(coords, conn, 0.05, 3)
    nd = 3
    cs = case(coords, nd)
    kw = {}
    if need_fill:
        rp0, ci0 = po.pattern(coords.shape[0], conn, nd, 2)
        rp, ci = extended_pattern(rp0, ci0, cs)
        kw = dict(row_ptr=rp, col_idx=ci)
    asm = capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, keep_off_diags=True, **KW, **kw)
    dev = torch.device("cuda", 0)
    st = meshgen.smooth_state(coords, nd, seed=2, amp=5e-3, noise=1e-4, l0=0.3)
    K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    asm.assemble(torch.from_numpy(st).to(dev), K, f, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    return asm, cs, K, f, dev


@pytest.mark.parametrize("case,need_fill", [(dirichlet_case, False), (periodic_case, True), (tied_case, True)])
def test_eliminate_matches_oracle(case, need_fill):
    import torch
    asm, cs, K, f, dev = setup(case, need_fill)
    rp, ci = asm.pattern()
    cons = asm.constraints(cs["slave"], cs.get("rval"), cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
    ref = po.Constraints(cs["slave"], cs.get("rval"), cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
    Kh = K.cpu().numpy().copy(); rh = -f.cpu().numpy().copy()
    rhs = (-f).clone()
    K2 = K.clone(); rhs2 = rhs.clone()
    cons.eliminate(K, rhs)
    cons.eliminate(K2, rhs2)
    torch.cuda.synchronize()
    ref.eliminate(rp, ci, Kh, rh)
    Kd = K.cpu().numpy(); rd = rhs.cpu().numpy()
    assert np.abs(Kd - Kh).max() <= 1e-12 * np.abs(Kh).max()
    assert np.abs(rd - rh).max() <= 1e-12 * np.abs(rh).max()
    assert torch.equal(K, K2) and torch.equal(rhs, rhs2)          # bitwise repeatable
    # the constrained matrix: unit rows / columns on the slaves, zero right-hand side there
    M = sp.csr_matrix((Kd, ci, rp), shape=(asm.ndof, asm.ndof))
    S = np.asarray(cs["slave"])
    assert np.abs(M[S][:, S] - sp.identity(S.size)).max() == 0.0
    mask = np.ones(asm.ndof, bool); mask[S] = False
    assert abs(M[S][:, mask]).max() == 0.0 and abs(M[mask][:, S]).max() == 0.0 and np.all(rd[S] == 0.0)
    # evalSlaveDofs on a solution vector (getLhs, cuDSSSolver.cpp:416)
    du = np.random.default_rng(1).normal(size=asm.ndof)
    dud = torch.from_numpy(du.copy()).to(dev)
    cons.eval_slaves(dud)
    exp = ref.eval_slaves(du.copy())
    torch.cuda.synchronize()
    assert np.abs(dud.cpu().numpy() - exp).max() <= 1e-13 * np.abs(exp).max()
    if cs.get("rval") is not None and cs.get("mptr") is None:
        assert np.abs(exp[S] - cs["rval"]).max() == 0.0
    # new right-hand values (next load step)
    if cs.get("rval") is not None:
        rv2 = 2.0 * np.asarray(cs["rval"])
        cons.set_rvalues(rv2)
        ref2 = po.Constraints(cs["slave"], rv2, cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
        asmK = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); f2 = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
        st = meshgen.smooth_state(asm._keep[0], 3, seed=2, amp=5e-3, noise=1e-4, l0=0.3)
        asm.assemble(torch.from_numpy(st).to(dev), asmK, f2, flags=capi.OVERWRITE)
        Kh2 = asmK.cpu().numpy().copy(); rh2 = -f2.cpu().numpy()
        rhsd = -f2
        cons.eliminate(asmK, rhsd)
        torch.cuda.synchronize()
        ref2.eliminate(rp, ci, Kh2, rh2)
        assert np.abs(rhsd.cpu().numpy() - rh2).max() <= 1e-12 * np.abs(rh2).max()
    cons.close(); asm.close()


def scatter_into(rp, ci, K, rpo, cio):
    """values of a CSR matrix placed into a pattern that contains its own; the new entries zero"""
    n = rp.size - 1
    out = np.zeros(cio.size)
    for r in range(n):
        pos = {int(c): q for q, c in zip(range(rpo[r], rpo[r + 1]), cio[rpo[r]:rpo[r + 1]])}
        for p in range(rp[r], rp[r + 1]):
            out[pos[int(ci[p])]] = K[p]
    return out


@pytest.mark.parametrize("case", [periodic_case, tied_case])
@pytest.mark.parametrize("shuffled_rows", [False, True])
def test_fill_in_is_built_by_the_constraint_set(case, shuffled_rows):
    """Constraints with masters on the assembler's own pattern: what StdConstrainer's new matrix structure holds
    (cuDSSSolver.cpp:108-121) is laid out by fa_cons_create -- the canonical pattern plus the fill-in, rows kept sorted by
    column (or, for a caller's pattern with unsorted rows, the fill-in appended to the row) -- and fa_cons_eliminate_out
    fills it: against the oracle's elimination on that pattern, and against the assembler given the pre-extended pattern."""
    import torch
    kw = {}
    if shuffled_rows:                                             # a caller's pattern whose rows are not sorted by column
        coords, conn = meshgen.grid_quad4(14, 9)
        rp0, ci0 = po.pattern(coords.shape[0], conn, 3, 2)
        ci0 = ci0.copy(); rng = np.random.default_rng(3)
        for r in range(rp0.size - 1):
            ci0[rp0[r]:rp0[r + 1]] = rng.permutation(ci0[rp0[r]:rp0[r + 1]])
        kw = dict(row_ptr=rp0, col_idx=ci0)
    coords, conn = meshgen.grid_quad4(14, 9)
    coords = meshgen.distort(coords, conn, 0.05, 3)
    nd = 3
    cs = case(coords, nd)
    asm = capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, coords, conn, keep_off_diags=True, **KW, **kw)
    dev = torch.device("cuda", 0)
    st = meshgen.smooth_state(coords, nd, seed=2, amp=5e-3, noise=1e-4, l0=0.3)
    K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    asm.assemble(torch.from_numpy(st).to(dev), K, f, flags=capi.OVERWRITE)
    rp, ci = asm.pattern()
    cons = asm.constraints(cs["slave"], cs.get("rval"), cs["mptr"], cs["mdof"], cs["mcoef"])
    assert cons.fill_in > 0 and cons.nnz == asm.nnz + cons.fill_in
    with pytest.raises(capi.FaError, match="fa_cons_eliminate_out"):
        cons.eliminate(K.clone(), (-f).clone())                   # in place: no room for the fill-in
    rpo, cio = cons.pattern()
    n = asm.ndof
    assert rpo[0] == 0 and rpo[-1] == cons.nnz and np.all(np.diff(rpo) >= np.diff(rp))
    rpe, cie = extended_pattern(rp, ci, cs)
    for r in range(n):
        old = ci[rp[r]:rp[r + 1]]; new = cio[rpo[r]:rpo[r + 1]]
        assert np.unique(new).size == new.size and set(old) <= set(new) <= set(cie[rpe[r]:rpe[r + 1]])
        if shuffled_rows:
            assert np.array_equal(new[:old.size], old)            # the caller's order kept, fill-in behind it
        else:
            assert np.all(np.diff(new) > 0)                       # still sorted by column
    Kh_in = K.cpu().numpy(); rh = -f.cpu().numpy().copy()
    rhs = (-f).clone(); rhs2 = (-f).clone()
    Ko = cons.eliminate_out(K, rhs)
    Ko2 = cons.eliminate_out(K, rhs2)
    torch.cuda.synchronize()
    assert np.array_equal(K.cpu().numpy(), Kh_in)                 # the input matrix is left alone
    assert torch.equal(Ko, Ko2) and torch.equal(rhs, rhs2)
    Kh = scatter_into(rp, ci, Kh_in, rpo, cio)
    ref = po.Constraints(cs["slave"], cs.get("rval"), cs["mptr"], cs["mdof"], cs["mcoef"])
    ref.eliminate(rpo, cio, Kh, rh)                               # raises if the pattern lacked an entry the elimination writes
    assert np.abs(Ko.cpu().numpy() - Kh).max() <= 1e-12 * np.abs(Kh).max()
    assert np.abs(rhs.cpu().numpy() - rh).max() <= 1e-12 * np.abs(rh).max()
    # 32-bit offsets of the output pattern for the solver libraries (cuDSSSolver.cpp:655-663)
    p_rp, p_ci = cons.pattern_device32()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rp32 = np.empty(n + 1, dtype=np.int32); ci32 = np.empty(cons.nnz, dtype=np.int32)
    assert hip.hipMemcpy(rp32.ctypes.data, p_rp, rp32.nbytes, 2) == 0 and hip.hipMemcpy(ci32.ctypes.data, p_ci, ci32.nbytes, 2) == 0
    assert np.array_equal(rp32.astype(np.int64), rpo) and np.array_equal(ci32, cio)
    if not shuffled_rows:
        # the same system as from an assembler that was handed the pre-extended pattern
        asm2, cs2, K2, f2, _ = setup(case, True)
        cons2 = asm2.constraints(cs2["slave"], cs2.get("rval"), cs2["mptr"], cs2["mdof"], cs2["mcoef"])
        assert cons2.fill_in == 0
        rhs3 = (-f2).clone()
        cons2.eliminate(K2, rhs3)
        torch.cuda.synchronize()
        rp2, ci2 = asm2.pattern()
        A = sp.csr_matrix((Ko.cpu().numpy(), cio, rpo), shape=(n, n)); B = sp.csr_matrix((K2.cpu().numpy(), ci2, rp2), shape=(n, n))
        assert abs(A - B).max() <= 1e-12 * abs(B).max()
        assert np.abs(rhs.cpu().numpy() - rhs3.cpu().numpy()).max() <= 1e-12 * np.abs(rh).max()
        cons2.close(); asm2.close()
    # without fill-in eliminate_out is the in-place elimination into a copy
    d = dirichlet_case(coords, nd)
    cd = asm.constraints(d["slave"], d["rval"])
    assert cd.fill_in == 0 and cd.nnz == asm.nnz
    Ka = K.clone(); ra = (-f).clone(); rb = (-f).clone()
    cd.eliminate(Ka, ra)
    Kb = cd.eliminate_out(K, rb)
    assert torch.equal(Ka, Kb) and torch.equal(ra, rb)
    cd.close(); cons.close(); asm.close()


def test_bad_constraints_are_refused():
    asm, cs, K, f, dev = setup(periodic_case, False)
    with pytest.raises(capi.FaError, match="constrained twice"):
        asm.constraints([3, 4, 3])
    with pytest.raises(capi.FaError, match="out of range"):
        asm.constraints([asm.ndof])
    with pytest.raises(capi.FaError, match="itself a slave"):
        asm.constraints([3, 4], None, [0, 1, 1], [4], [1.0])
    empty = asm.constraints(np.zeros(0, dtype=np.int64))
    Kc = K.clone()
    empty.eliminate(K, f)
    assert (K == Kc).all()
    asm.close()


@pytest.mark.parametrize("case,need_fill", [(dirichlet_case, False), (tied_case, True)])
@pytest.mark.parametrize("bounded", [False, True])
def test_residual_and_master_evaluation(case, need_fill, bounded):
    """BFGSModule::Work_::updateResidual (BFGSModule.cpp:438-489) on device vectors."""
    import torch
    asm, cs, K, f, dev = setup(case, need_fill)
    cons = asm.constraints(cs["slave"], cs.get("rval"), cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
    ref = po.Constraints(cs["slave"], cs.get("rval"), cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
    n = asm.ndof
    rng = np.random.default_rng(5)
    fint = f.cpu().numpy(); fext = 0.1 * rng.normal(size=n) * np.abs(fint).max(); du = rng.normal(size=n); u = rng.normal(size=n)
    lower = np.where(rng.random(n) < 0.4, u, u - 1.0); upper = np.where(rng.random(n) < 0.4, u, u + 1.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    r = torch.empty(n, dtype=torch.float64, device=dev); r2 = torch.empty(n, dtype=torch.float64, device=dev)
    if bounded:
        out = cons.residual(t(fext), f, t(du), r, u=t(u), lower=t(lower), upper=t(upper), r2=r2)
        out_b = cons.residual(t(fext), f, t(du), r, u=t(u), lower=t(lower), upper=t(upper), r2=r2)
        re, r2e, oe = ref.residual(fext, fint, du, u, lower, upper)
        assert np.array_equal(r2.cpu().numpy(), r2e)
    else:
        out = cons.residual(t(fext), f, t(du), r)
        out_b = cons.residual(t(fext), f, t(du), r)
        re, _, oe = ref.residual(fext, fint, du)
    torch.cuda.synchronize()
    assert np.abs(r.cpu().numpy() - re).max() <= 1e-14 * np.abs(re).max()
    o = out.cpu().numpy()
    assert abs(o[0] - oe[0]) <= 1e-12 * oe[0] and abs(o[2] - oe[2]) <= 1e-12 * oe[2]
    assert abs(o[1] - oe[1]) <= 1e-12 * np.abs(re * du).sum()
    assert torch.equal(out, out_b)
    v = rng.normal(size=n)
    vd = t(v)
    cons.eval_masters(vd)
    torch.cuda.synchronize()
    assert np.abs(vd.cpu().numpy() - ref.eval_masters(v.copy())).max() <= 1e-14 * np.abs(v).max()
    cons.close(); asm.close()


def test_pattern_with_32_bit_offsets():
    """cuDSSSolver.cpp:655-663: the solver libraries take int offsets / indices."""
    import torch
    asm, cs, K, f, dev = setup(dirichlet_case, False)
    rp, ci = asm.pattern()
    p_rp, p_ci = asm.pattern_device32()
    assert asm.pattern_device32() == (p_rp, p_ci)                  # made once, kept
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rp32 = np.empty(asm.ndof + 1, dtype=np.int32); ci32 = np.empty(asm.nnz, dtype=np.int32)
    assert hip.hipMemcpy(rp32.ctypes.data, p_rp, rp32.nbytes, 2) == 0
    assert hip.hipMemcpy(ci32.ctypes.data, p_ci, ci32.nbytes, 2) == 0
    assert np.array_equal(rp32.astype(np.int64), rp) and np.array_equal(ci32, ci)
    assert capi.lib().fa_device(asm.h) == 0
    asm.close()


@pytest.mark.parametrize("shape,gen", [(capi.QUAD8_G3, "grid_quad8"), (capi.TRI6_G3, "grid_tri6")])
def test_gradient_damage_mid_node_constraints_need_no_fill_in(shape, gen):
    """The constraints GradientDamageModel::setConstraints_ adds under INIT (GradientDamageModel.cpp:682-846: eqv dof of a
    mid-side node = mean of its edge's corner dofs) eliminated on the device from a GradientDamage matrix in its canonical
    pattern: every fold destination lies inside an element that holds the edge, so no extra entries are needed."""
    import torch
    coords, conn = getattr(meshgen, gen)(6, 5)
    coords = meshgen.distort(coords, conn, 0.05, 3)
    nd = 3
    slave, mdof = [], []
    for c in conn:
        for j in range(len(c) // 2):
            s = int(c[2 * j + 1]) * nd + 2
            if s not in slave:
                slave.append(s); mdof += [int(c[2 * j]) * nd + 2, int(c[(2 * j + 2) % len(c)]) * nd + 2]
    slave = np.array(slave); mdof = np.array(mdof); mptr = 2 * np.arange(slave.size + 1); mcoef = np.full(mdof.size, 0.5)
    asm = capi.Assembler(shape, capi.GRADIENT_DAMAGE, capi.DAMAGE, coords, conn, young=210.0, poisson=0.3,
                         softening=capi.SOFT_EXPONENTIAL2, kappa_i=1e-3, kappa_c=6e-3, length_scale=0.02)
    dev = torch.device("cuda", 0)
    st = meshgen.smooth_state(coords, nd, seed=2, amp=5e-3, noise=1e-4, l0=0.3).reshape(-1, nd)
    st[:, 2] = 3e-3 + 2e-3 * np.sin(3 * coords[:, 0])
    K = torch.zeros(asm.nnz, dtype=torch.float64, device=dev); f = torch.zeros(asm.ndof, dtype=torch.float64, device=dev)
    asm.assemble(torch.from_numpy(st.ravel().copy()).to(dev), K, f, flags=capi.OVERWRITE)
    torch.cuda.synchronize()
    rp, ci = asm.pattern()
    cons = asm.constraints(slave, None, mptr, mdof, mcoef)                # canonical pattern: accepted as it is
    ref = po.Constraints(slave, None, mptr, mdof, mcoef)
    Kh = K.cpu().numpy().copy(); rh = -f.cpu().numpy().copy()
    rhs = (-f).clone()
    cons.eliminate(K, rhs)
    torch.cuda.synchronize()
    ref.eliminate(rp, ci, Kh, rh)
    assert np.abs(K.cpu().numpy() - Kh).max() <= 1e-12 * np.abs(Kh).max()
    assert np.abs(rhs.cpu().numpy() - rh).max() <= 1e-12 * np.abs(rh).max()
    u = torch.from_numpy(st.ravel().copy()).to(dev)
    cons.eval_slaves(u)                                                   # the nonlocal strain becomes edge-wise linear
    uh = u.cpu().numpy()
    assert np.abs(uh[slave] - 0.5 * (uh[mdof[0::2]] + uh[mdof[1::2]])).max() == 0.0
    cons.close(); asm.close()
