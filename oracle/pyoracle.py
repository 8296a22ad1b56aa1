"""ctypes binding of the CPU oracle (oracle/falcon_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  PARITY UNPINNED
(see falcon_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfalcon_oracle.so")

# shape ids
TRI3_G1, TRI3_G3, QUAD4_G2, QUAD8_G3, QUAD8_G2, TRI6_G3, HEX8_G2, HEX20_G3, HEX20_G2 = range(9)
# models
(LINEAR_ELASTICITY, PHASE_FRACTURE, PHASE_FRACTURE_EXT, PHASE_FRACTURE_EXTIT, MICRO_PHASE_FRACTURE, MICRO_EXT,
 MICRO_EXTIT, GRADIENT_DAMAGE) = range(8)
HISTORY_MODELS = (PHASE_FRACTURE, MICRO_PHASE_FRACTURE, MICRO_EXT, MICRO_EXTIT, GRADIENT_DAMAGE)
# materials
HOOKE, AMOR, BOURDIN, MIEHE, DAMAGE = range(5)
(SOFT_PERFECT, SOFT_LINEAR1, SOFT_LINEAR2, SOFT_EXPONENTIAL1, SOFT_EXPONENTIAL2, SOFT_POWER, SOFT_HYPERBOLIC,
 SOFT_EXPONENTIAL2REG, SOFT_EXPOENERGY, SOFT_POLYNOMIAL, SOFT_MULTILINEAR) = range(11)
EQV_ENERGY, EQV_MAZARS, EQV_VONMISES = 0, 1, 2
PLANE_STRAIN, PLANE_STRESS = 0, 1
AT1, AT2, CZM_LINEAR, CZM_EXPONENTIAL, CZM_CORNELISSEN, CZM_HYPERBOLIC = range(6)
PATTERN_FULL, PATTERN_WRITTEN = 0, 1


class Config(C.Structure):
    _fields_ = [("shape", C.c_int), ("model", C.c_int), ("material", C.c_int), ("state", C.c_int),
                ("young", C.c_double), ("poisson", C.c_double),
                ("frac_type", C.c_int),
                ("gc", C.c_double), ("l0", C.c_double), ("sigma_t", C.c_double), ("beta", C.c_double),
                ("keep_off_diags", C.c_int),
                ("softening", C.c_int), ("kappa_i", C.c_double), ("kappa_c", C.c_double), ("soft_alpha", C.c_double),
                ("soft_beta", C.c_double), ("length_scale", C.c_double), ("eqv_strain", C.c_int),
                ("soft_b", C.c_double), ("soft_s", C.c_double), ("kappa_p", C.c_double), ("eta", C.c_double)]


class ShapeInfo(C.Structure):
    _fields_ = [("rank", C.c_int), ("npe", C.c_int), ("nip", C.c_int), ("str", C.c_int)]


class Cons(C.Structure):
    _fields_ = [("nslave", C.c_int64), ("slave", C.POINTER(C.c_int64)), ("rval", C.POINTER(C.c_double)),
                ("mptr", C.POINTER(C.c_int64)), ("mdof", C.POINTER(C.c_int64)), ("mcoef", C.POINTER(C.c_double))]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        dp, ip32, ip64 = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        L.fo_shape_info_get.argtypes = [C.c_int, C.POINTER(ShapeInfo)]
        L.fo_shape_tables.argtypes = [C.c_int, dp, dp, dp]
        L.fo_ndof_per_node.argtypes = [C.c_int, C.c_int]
        L.fo_micro_stale_hits.restype = C.c_int64
        L.fo_psi0.argtypes = [C.POINTER(Config)]
        L.fo_psi0.restype = C.c_double
        L.fo_degradation.argtypes = [C.POINTER(Config), C.c_double, dp, dp, dp]
        L.fo_degradation.restype = None
        L.fo_material_update.argtypes = [C.POINTER(Config), C.c_int, dp, C.c_double, dp, dp, dp, dp]
        L.fo_pattern.argtypes = [C.c_int, C.c_int64, C.c_int, ip32, C.c_int, C.c_int, C.c_int, C.c_int, ip64, ip32]
        L.fo_pattern.restype = C.c_int64
        L.fo_assemble.argtypes = [C.POINTER(Config), C.c_int, dp, C.c_int64, ip32, C.c_int64, C.c_int64,
                                  ip64, ip32, dp, dp, dp, C.c_double, C.c_double, dp, dp, dp, dp, dp, dp]
        L.fo_strain_table.argtypes = [C.c_int, C.c_int64, ip32, dp, C.c_int, C.c_int, dp, dp]
        L.fo_stress_table.argtypes = [C.POINTER(Config), C.c_int64, ip32, dp, C.c_int, dp, dp]
        L.fo_gd_stress_table.argtypes = [C.POINTER(Config), C.c_int64, ip32, dp, dp, dp, dp, dp]
        L.fo_gd_strain_table.argtypes = [C.POINTER(Config), C.c_int64, ip32, dp, dp, dp]
        L.fo_gd_damage_table.argtypes = [C.POINTER(Config), C.c_int64, ip32, dp, dp, dp]
        L.fo_gd_output_data.argtypes = [C.POINTER(Config), C.c_int64, ip32, dp, dp, dp, C.c_int, ip32, dp, dp]
        L.fo_last_error.restype = C.c_char_p
        L.fo_mazars.argtypes = [C.POINTER(Config), C.c_int, dp, dp, dp, dp]
        L.fo_vonmises.argtypes = [C.POINTER(Config), dp, dp, dp]
        L.fo_arc_func.argtypes = [C.POINTER(Config), C.c_int, dp, C.c_int64, ip32, dp, dp, dp, dp]
        L.fo_eval_masters.argtypes = [dp, C.POINTER(Cons)]; L.fo_eval_masters.restype = None
        L.fo_eval_slaves.argtypes = [dp, C.POINTER(Cons)]; L.fo_eval_slaves.restype = None
        L.fo_cons_eliminate.argtypes = [C.c_int64, ip64, ip32, dp, dp, C.POINTER(Cons)]
        L.fo_residual.argtypes = [C.c_int64, C.POINTER(Cons), dp, dp, dp, dp, dp, dp, dp, dp, dp]
        L.fo_residual.restype = None
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def shape_info(shape):
    si = ShapeInfo()
    if lib().fo_shape_info_get(shape, C.byref(si)):
        raise ValueError(lib().fo_last_error().decode())
    return si.rank, si.npe, si.nip, si.str


def shape_tables(shape):
    rank, npe, nip, _ = shape_info(shape)
    N = np.zeros((nip, npe)); dN = np.zeros((nip, npe, rank)); w = np.zeros(nip)
    lib().fo_shape_tables(shape, _p(N, C.c_double), _p(dN, C.c_double), _p(w, C.c_double))
    return N, dN, w


def strain_table(shape, nnode, conn, strain):
    """getStrain_: (table[nnode][str], weights[nnode]) from the strain_ cache [nip_total][srows]."""
    rank = shape_info(shape)[0]
    str_ = 4 if rank == 2 else 6
    conn = np.ascontiguousarray(conn, dtype=np.int32); strain = np.ascontiguousarray(strain, dtype=np.float64)
    table = np.zeros((nnode, str_)); weights = np.zeros(nnode)
    rc = lib().fo_strain_table(shape, conn.shape[0], _p(conn, C.c_int32), _p(strain, C.c_double), strain.shape[1], str_,
                               _p(table, C.c_double), _p(weights, C.c_double))
    assert rc == 0
    return table, weights


def stress_table(cfg, nnode, conn, strain):
    """getStress_: (table[nnode][str], weights[nnode]) from the strain_ cache [nip_total][srows] (last row = cached g)."""
    rank = shape_info(cfg.shape)[0]
    str_ = 4 if rank == 2 else 6
    conn = np.ascontiguousarray(conn, dtype=np.int32); strain = np.ascontiguousarray(strain, dtype=np.float64)
    table = np.zeros((nnode, str_)); weights = np.zeros(nnode)
    rc = lib().fo_stress_table(C.byref(cfg), conn.shape[0], _p(conn, C.c_int32), _p(strain, C.c_double), strain.shape[1],
                               _p(table, C.c_double), _p(weights, C.c_double))
    if rc != 0:
        raise ValueError(lib().fo_last_error().decode())
    return table, weights


def _gd_common(cfg, conn, *arrs):
    rank = shape_info(cfg.shape)[0]
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    return (rank, 4 if rank == 2 else 6, conn) + tuple(np.ascontiguousarray(a, dtype=np.float64) for a in arrs)


def _gd_rc(rc):
    if rc != 0:
        raise ValueError(lib().fo_last_error().decode())


def gd_stress_table(cfg, nnode, conn, strain, hist_pre):
    """GradientDamageModel::getStress_: (table[nnode][str], weights[nnode], newHist_ as the update leaves it)."""
    rank, str_, conn, strain, hist_pre = _gd_common(cfg, conn, strain, hist_pre)
    table = np.zeros((nnode, str_)); weights = np.zeros(nnode); hist_new = np.zeros_like(hist_pre)
    _gd_rc(lib().fo_gd_stress_table(C.byref(cfg), conn.shape[0], _p(conn, C.c_int32), _p(strain, C.c_double), _p(hist_pre, C.c_double),
                                    _p(hist_new, C.c_double), _p(table, C.c_double), _p(weights, C.c_double)))
    return table, weights, hist_new


def gd_strain_table(cfg, nnode, conn, strain):
    """GradientDamageModel::getStrain_: (table[nnode][str + 1], weights); ValueError in 2-D like the reference's throw."""
    rank, str_, conn, strain = _gd_common(cfg, conn, strain)
    table = np.zeros((nnode, str_ + 1)); weights = np.zeros(nnode)
    _gd_rc(lib().fo_gd_strain_table(C.byref(cfg), conn.shape[0], _p(conn, C.c_int32), _p(strain, C.c_double),
                                    _p(table, C.c_double), _p(weights, C.c_double)))
    return table, weights


def gd_damage_table(cfg, nnode, conn, hist_new):
    """GradientDamageModel::getDamage_: (table[nnode][1], weights) from newHist_, |N| weights."""
    rank, str_, conn, hist_new = _gd_common(cfg, conn, hist_new)
    table = np.zeros((nnode, 1)); weights = np.zeros(nnode)
    _gd_rc(lib().fo_gd_damage_table(C.byref(cfg), conn.shape[0], _p(conn, C.c_int32), _p(hist_new, C.c_double),
                                    _p(table, C.c_double), _p(weights, C.c_double)))
    return table, weights


def gd_output_data(cfg, nnode, conn, strain, hist_pre, hist_new, cols):
    """GradientDamageModel::getOutputData_: columns `cols` of [strain.tensor | stress.tensor | damage];
    returns (table, weights, newHist_ as the local update leaves it)."""
    rank, str_, conn, strain, hist_pre = _gd_common(cfg, conn, strain, hist_pre)
    hist_new = np.array(hist_new, dtype=np.float64)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    table = np.zeros((nnode, cols.size)); weights = np.zeros(nnode)
    _gd_rc(lib().fo_gd_output_data(C.byref(cfg), conn.shape[0], _p(conn, C.c_int32), _p(strain, C.c_double), _p(hist_pre, C.c_double),
                                   _p(hist_new, C.c_double), cols.size, _p(cols, C.c_int32), _p(table, C.c_double), _p(weights, C.c_double)))
    return table, weights, hist_new


def ndof_per_node(model, rank):
    return lib().fo_ndof_per_node(model, rank)


def make_config(shape, model, material, state=PLANE_STRAIN, young=210e3, poisson=0.3, frac_type=AT2,
                gc=2.7, l0=0.015, sigma_t=0.0, beta=6500.0, keep_off_diags=0, softening=0, kappa_i=0.0, kappa_c=0.0,
                soft_alpha=0.0, soft_beta=0.0, length_scale=0.0, eqv_strain=0, soft_b=0.0, soft_s=0.0, kappa_p=0.0, eta=0.0):
    return Config(shape, model, material, state, young, poisson, frac_type, gc, l0, sigma_t, beta,
                  int(keep_off_diags), softening, kappa_i, kappa_c, soft_alpha, soft_beta, length_scale, eqv_strain,
                  soft_b, soft_s, kappa_p, eta)


def psi0(cfg):
    return lib().fo_psi0(C.byref(cfg))


def degradation(cfg, pf):
    g, dg, ddg = C.c_double(), C.c_double(), C.c_double()
    lib().fo_degradation(C.byref(cfg), pf, C.byref(g), C.byref(dg), C.byref(ddg))
    return g.value, dg.value, ddg.value


def material_update(cfg, rank, strain, gphi=1.0):
    n = {2: 4, 3: 6}[rank]
    strain = np.ascontiguousarray(strain, dtype=np.float64)
    stress = np.zeros(n); stiff = np.zeros((n, n)); sp = np.zeros(n); psi = C.c_double()
    if lib().fo_material_update(C.byref(cfg), rank, _p(strain, C.c_double), gphi, _p(stress, C.c_double),
                                _p(stiff, C.c_double), C.byref(psi), _p(sp, C.c_double)):
        raise ValueError(lib().fo_last_error().decode())
    return stress, stiff, psi.value, sp


def vonmises(cfg, strain6):
    """(eq, d eq / d strain) of the restated von Mises equivalent strain for a 6-component strain vector."""
    strain6 = np.ascontiguousarray(strain6, dtype=np.float64)
    eq = C.c_double(); d = np.zeros(6)
    lib().fo_vonmises(C.byref(cfg), _p(strain6, C.c_double), C.byref(eq), _p(d, C.c_double))
    return eq.value, d


def mazars(cfg, rank, strain):
    """(eq, d eq / d strain, principal values) as DamageMaterial computes them."""
    strain = np.ascontiguousarray(strain, dtype=np.float64)
    eq = C.c_double(); d = np.zeros(strain.size); pr = np.zeros(3)
    if lib().fo_mazars(C.byref(cfg), rank, _p(strain, C.c_double), C.byref(eq), _p(d, C.c_double), _p(pr, C.c_double)):
        raise ValueError(lib().fo_last_error().decode())
    return eq.value, d, pr


def keep_code(cfg):
    """keep_off_diags code of fo_pattern for the WRITTEN pattern of this model."""
    if cfg.model in (PHASE_FRACTURE, MICRO_PHASE_FRACTURE):
        return 1 if cfg.keep_off_diags else 0
    if cfg.model in (PHASE_FRACTURE_EXT, PHASE_FRACTURE_EXTIT, MICRO_EXT, MICRO_EXTIT):
        return 2
    if cfg.model == GRADIENT_DAMAGE:
        return 1
    return 0


def pattern(nnode, conn, ndofpn, rank, kind=PATTERN_FULL, keep=0):
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    nel, npe = conn.shape
    row_ptr = np.zeros(nnode * ndofpn + 1, dtype=np.int64)
    nnz = lib().fo_pattern(nnode, nel, npe, _p(conn, C.c_int32), ndofpn, rank, kind, keep,
                           _p(row_ptr, C.c_int64), None)
    if nnz < 0:
        raise ValueError(lib().fo_last_error().decode())
    col_idx = np.zeros(nnz, dtype=np.int32)
    lib().fo_pattern(nnode, nel, npe, _p(conn, C.c_int32), ndofpn, rank, kind, keep,
                     _p(row_ptr, C.c_int64), _p(col_idx, C.c_int32))
    return row_ptr, col_idx


def assemble(cfg, coords, conn, row_ptr, col_idx, state, state0=None, state00=None, dt=0.0, dt0=0.0,
             hist_pre=None, want_K=True, want_strain=False, K=None, f=None, el_range=None, hist_new=None):
    """One GET_MATRIX0 (want_K) / GET_INT_VECTOR pass.  Returns dict(K, f, hist_new, strain, err_ext).
    For ExtIt pass stateExt_ as state00."""
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    conn = np.ascontiguousarray(conn, dtype=np.int32)
    nnode, rank = coords.shape
    nel, npe = conn.shape
    r, npe_s, nip, strc = shape_info(cfg.shape)
    assert r == rank and npe_s == npe
    ndof = nnode * ndof_per_node(cfg.model, rank)
    state = np.ascontiguousarray(state, dtype=np.float64)
    assert state.size == ndof
    if K is None and want_K:
        K = np.zeros(row_ptr[-1])
    if f is None:
        f = np.zeros(ndof)
    if cfg.model in HISTORY_MODELS:
        if hist_pre is None:     # preHist_.H0_ = Psi0_ (PhaseFractureModel.cpp:359); preHist_.phasef_ = 0 (MicroPhaseFractureModel.cpp:359)
            hist_pre = np.full(nel * nip, psi0(cfg) if cfg.model == PHASE_FRACTURE else 0.0)
        hist_pre = np.ascontiguousarray(hist_pre, dtype=np.float64)
        if hist_new is None:
            hist_new = np.zeros(nel * nip)
    else:
        hist_new = None
    srows = strc + (0 if cfg.model == LINEAR_ELASTICITY else 1)
    strain = np.zeros((nel * nip, srows)) if want_strain else None
    err = C.c_double(0.0)
    b, e = (0, nel) if el_range is None else el_range
    if state0 is not None:
        state0 = np.ascontiguousarray(state0, dtype=np.float64)
    if state00 is not None:
        state00 = np.ascontiguousarray(state00, dtype=np.float64)
    rc = lib().fo_assemble(C.byref(cfg), nnode, _p(coords, C.c_double), nel, _p(conn, C.c_int32), b, e,
                           _p(row_ptr, C.c_int64), _p(col_idx, C.c_int32),
                           _p(state, C.c_double), _p(state0, C.c_double), _p(state00, C.c_double),
                           dt, dt0, _p(hist_pre, C.c_double), _p(hist_new, C.c_double),
                           _p(strain, C.c_double), _p(K if want_K else None, C.c_double), _p(f, C.c_double),
                           C.byref(err))
    if rc:
        raise ValueError(lib().fo_last_error().decode())
    return dict(K=K if want_K else None, f=f, hist_new=hist_new, strain=strain, err_ext=err.value,
                micro_stale_hits=int(lib().fo_micro_stale_hits()))


def arc_func(cfg, coords, conn, state, state0):
    """getArcFunc_: (jac10[ndof], fe_dissipation)"""
    coords = np.ascontiguousarray(coords, dtype=np.float64); conn = np.ascontiguousarray(conn, dtype=np.int32)
    state = np.ascontiguousarray(state, dtype=np.float64); state0 = np.ascontiguousarray(state0, dtype=np.float64)
    jac10 = np.zeros(state.size); fv = C.c_double(0.0)
    rc = lib().fo_arc_func(C.byref(cfg), coords.shape[0], _p(coords, C.c_double), conn.shape[0], _p(conn, C.c_int32),
                           _p(state, C.c_double), _p(state0, C.c_double), _p(jac10, C.c_double), C.byref(fv))
    if rc:
        raise ValueError(lib().fo_last_error().decode())
    return jac10, fv.value


class Constraints:
    """u[slave] = rval + sum coef * u[master] (falcon_oracle_cons.c)."""

    def __init__(self, slave, rval=None, mptr=None, mdof=None, mcoef=None):
        self.slave = np.ascontiguousarray(slave, dtype=np.int64)
        self.rval = None if rval is None else np.ascontiguousarray(rval, dtype=np.float64)
        self.mptr = None if mptr is None else np.ascontiguousarray(mptr, dtype=np.int64)
        self.mdof = None if mdof is None else np.ascontiguousarray(mdof, dtype=np.int64)
        self.mcoef = None if mcoef is None else np.ascontiguousarray(mcoef, dtype=np.float64)
        self.c = Cons(self.slave.size, _p(self.slave, C.c_int64), _p(self.rval, C.c_double), _p(self.mptr, C.c_int64),
                      _p(self.mdof, C.c_int64), _p(self.mcoef, C.c_double))

    def eval_masters(self, vec):
        assert vec.dtype == np.float64 and vec.flags.c_contiguous
        lib().fo_eval_masters(_p(vec, C.c_double), C.byref(self.c))
        return vec

    def eval_slaves(self, vec):
        assert vec.dtype == np.float64 and vec.flags.c_contiguous
        lib().fo_eval_slaves(_p(vec, C.c_double), C.byref(self.c))
        return vec

    def eliminate(self, row_ptr, col_idx, K, rhs=None):
        """in place; raises when the pattern lacks a fill-in entry"""
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64); col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        assert K.dtype == np.float64 and K.flags.c_contiguous
        rc = lib().fo_cons_eliminate(row_ptr.size - 1, _p(row_ptr, C.c_int64), _p(col_idx, C.c_int32), _p(K, C.c_double),
                                     _p(rhs, C.c_double), C.byref(self.c))
        if rc:
            raise ValueError("the pattern lacks an entry the elimination writes")
        return K, rhs

    def residual(self, fext, fint, du, u=None, lower=None, upper=None):
        """(r, r2 or None, [rnorm, enorm, rnorm^2]) of BFGSModule::Work_::updateResidual"""
        n = fext.size
        r = np.zeros(n); r2 = np.zeros(n) if lower is not None else None; out = np.zeros(3)
        a = [None if v is None else np.ascontiguousarray(v, dtype=np.float64) for v in (fext, fint, du, u, lower, upper)]
        lib().fo_residual(n, C.byref(self.c), *[_p(v, C.c_double) for v in a], _p(r, C.c_double), _p(r2, C.c_double),
                          _p(out, C.c_double))
        return r, r2, out
