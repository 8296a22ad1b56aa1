"""ctypes binding of libfalcon_assemble.so (include/falcon_assemble.h) + a thin torch-backed helper.

PyTorch is used only for device memory and streams; the arithmetic is in the HIP library.  There is
no CPU fallback: if the library is missing or no GPU is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FA_LIB") or os.path.join(_HERE, "libfalcon_assemble.so")

# enums of include/falcon_assemble.h
TRI3_G1, TRI3_G3, QUAD4_G2, QUAD8_G3, QUAD8_G2, TRI6_G3, HEX8_G2, HEX20_G3, HEX20_G2 = range(9)
(LINEAR_ELASTICITY, PHASE_FRACTURE, PHASE_FRACTURE_EXT, PHASE_FRACTURE_EXTIT, MICRO_PHASE_FRACTURE,
 MICRO_PHASE_FRACTURE_EXT, MICRO_PHASE_FRACTURE_EXTIT, GRADIENT_DAMAGE) = range(8)
HOOKE, AMOR_PHASE, BOURDIN_PHASE, MIEHE_PHASE, DAMAGE = range(5)
(SOFT_PERFECT, SOFT_LINEAR1, SOFT_LINEAR2, SOFT_EXPONENTIAL1, SOFT_EXPONENTIAL2, SOFT_POWER, SOFT_HYPERBOLIC,
 SOFT_EXPONENTIAL2REG, SOFT_EXPOENERGY, SOFT_POLYNOMIAL, SOFT_MULTILINEAR) = range(11)
EQV_ENERGY, EQV_MAZARS, EQV_VONMISES = 0, 1, 2
PLANE_STRAIN, PLANE_STRESS = 0, 1
AT1, AT2, CZM_LINEAR, CZM_EXPONENTIAL, CZM_CORNELISSEN, CZM_HYPERBOLIC = range(6)
PATTERN_FULL, PATTERN_WRITTEN = 0, 1
ACCUMULATE, OVERWRITE, STORE_STRAIN, FORCE_GENERIC, DETERMINISTIC = 0, 1, 2, 4, 8
PART_INTERFACE, PART_INTERIOR = 16, 32

SHAPE_INFO = {  # shape id -> (rank, npe, nip)
    TRI3_G1: (2, 3, 1), TRI3_G3: (2, 3, 3), QUAD4_G2: (2, 4, 4), QUAD8_G3: (2, 8, 9), QUAD8_G2: (2, 8, 4),
    TRI6_G3: (2, 6, 3), HEX8_G2: (3, 8, 8), HEX20_G3: (3, 20, 27), HEX20_G2: (3, 20, 8)}


class FaError(RuntimeError):
    pass


class Problem(C.Structure):
    _fields_ = [("shape", C.c_int32), ("model", C.c_int32), ("material", C.c_int32), ("state", C.c_int32),
                ("young", C.c_double), ("poisson", C.c_double),
                ("frac_type", C.c_int32),
                ("gc", C.c_double), ("l0", C.c_double), ("sigma_t", C.c_double), ("beta", C.c_double),
                ("keep_off_diags", C.c_int32),
                ("nnode", C.c_int64), ("coords", C.c_void_p),
                ("nel", C.c_int64), ("conn", C.c_void_p),
                ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p), ("node_dofs", C.c_void_p),
                ("ndof", C.c_int64), ("pattern_kind", C.c_int32), ("device", C.c_int32),
                ("softening", C.c_int32), ("kappa_i", C.c_double), ("kappa_c", C.c_double), ("soft_alpha", C.c_double),
                ("soft_beta", C.c_double), ("length_scale", C.c_double), ("eqv_strain", C.c_int32),
                ("soft_b", C.c_double), ("soft_s", C.c_double), ("kappa_p", C.c_double), ("eta", C.c_double)]


class AssembleArgs(C.Structure):
    _fields_ = [("state", C.c_void_p), ("state0", C.c_void_p), ("state00", C.c_void_p),
                ("dt", C.c_double), ("dt0", C.c_double),
                ("K_values", C.c_void_p), ("f_int", C.c_void_p), ("err_ext", C.c_void_p),
                ("flags", C.c_int32), ("stream", C.c_void_p)]


EXPORTS = ["fa_create", "fa_destroy", "fa_last_error", "fa_ndof", "fa_nnz", "fa_nip_total", "fa_ndof_per_node",
           "fa_strain_rows", "fa_uses_row_kernel", "fa_psi0", "fa_patch_stats", "fa_pattern_host", "fa_pattern_device",
           "fa_assemble", "fa_assemble_noalloc", "fa_assemble_host", "fa_commit", "fa_get_history", "fa_set_history",
           "fa_history_device", "fa_get_strain_cache", "fa_shape_info", "fa_shape_tables", "fa_gather_sum", "fa_strain_table", "fa_stress_table", "fa_damage_table",
           "fa_pattern_device32", "fa_device", "fa_cons_create", "fa_cons_destroy", "fa_cons_set_rvalues", "fa_cons_eliminate", "fa_cons_eliminate_out", "fa_cons_fill_in", "fa_cons_nnz",
           "fa_cons_pattern_device", "fa_cons_pattern_device32", "fa_cons_pattern_host",
           "fa_cons_eval_masters", "fa_cons_eval_slaves", "fa_residual", "fa_arc_func", "fa_output_data", "fa_micro_stale_hits", "fa_set_interface"]

_lib = None


def _torch_first():
    """PyTorch's wheel carries a HIP runtime of its own (torch/lib/libamdhip64.so) next to the /opt/rocm one this library
    links: the two share a process only when torch's has initialised the GPU first (the other order leaves torch with
    "No HIP GPUs are available").  Callers that hand over torch tensors therefore get torch initialised before the
    library's first HIP call; a process without torch is not affected."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FaError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _torch_first()
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.fa_create.argtypes = [C.POINTER(Problem), C.POINTER(vp)]
        L.fa_destroy.argtypes = [vp]; L.fa_destroy.restype = None
        L.fa_last_error.restype = C.c_char_p
        for name in ("fa_ndof", "fa_nnz", "fa_nip_total"):
            getattr(L, name).argtypes = [vp]; getattr(L, name).restype = C.c_int64
        for name in ("fa_ndof_per_node", "fa_strain_rows", "fa_uses_row_kernel"):
            getattr(L, name).argtypes = [vp]; getattr(L, name).restype = C.c_int32
        L.fa_psi0.argtypes = [vp]; L.fa_psi0.restype = C.c_double
        L.fa_patch_stats.argtypes = [vp, C.POINTER(C.c_double)]; L.fa_patch_stats.restype = C.c_int32
        L.fa_pattern_host.argtypes = [vp, vp, vp]
        L.fa_pattern_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.fa_assemble.argtypes = [vp, C.POINTER(AssembleArgs)]
        L.fa_assemble_noalloc.argtypes = [vp, C.POINTER(AssembleArgs)]
        L.fa_assemble_host.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp, C.c_int32]
        L.fa_commit.argtypes = [vp]
        L.fa_get_history.argtypes = [vp, C.c_int32, vp]
        L.fa_set_history.argtypes = [vp, C.c_int32, vp]
        L.fa_history_device.argtypes = [vp, C.c_int32]; L.fa_history_device.restype = vp
        L.fa_get_strain_cache.argtypes = [vp, vp]
        L.fa_gather_sum.argtypes = [vp, vp, vp, C.c_int64, vp, vp]
        L.fa_strain_table.argtypes = [vp, vp, vp, vp]
        L.fa_stress_table.argtypes = [vp, vp, vp, vp]
        L.fa_damage_table.argtypes = [vp, vp, vp, vp]
        L.fa_pattern_device32.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.fa_device.argtypes = [vp]; L.fa_device.restype = C.c_int32
        L.fa_cons_create.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, C.POINTER(vp)]
        L.fa_cons_destroy.argtypes = [vp]; L.fa_cons_destroy.restype = None
        L.fa_cons_set_rvalues.argtypes = [vp, vp, vp]
        L.fa_cons_eliminate.argtypes = [vp, vp, vp, vp]
        L.fa_cons_eliminate_out.argtypes = [vp, vp, vp, vp, vp]
        L.fa_cons_fill_in.argtypes = [vp]; L.fa_cons_fill_in.restype = C.c_int64
        L.fa_cons_nnz.argtypes = [vp]; L.fa_cons_nnz.restype = C.c_int64
        L.fa_cons_pattern_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.fa_cons_pattern_device32.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        L.fa_cons_pattern_host.argtypes = [vp, vp, vp]
        L.fa_cons_eval_masters.argtypes = [vp, vp, vp]
        L.fa_cons_eval_slaves.argtypes = [vp, vp, vp]
        L.fa_residual.argtypes = [vp] * 11
        L.fa_arc_func.argtypes = [vp] * 6
        L.fa_output_data.argtypes = [vp, C.c_int32, vp, vp, vp, vp]
        L.fa_micro_stale_hits.argtypes = [vp, C.POINTER(C.c_int64)]
        L.fa_set_interface.argtypes = [vp, C.c_int32, vp]
        ip = C.POINTER(C.c_int32)
        L.fa_shape_info.argtypes = [C.c_int32, ip, ip, ip]
        L.fa_shape_tables.argtypes = [C.c_int32, vp, vp, vp]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise FaError(f"falcon_assemble error {rc}: {lib().fa_last_error().decode()}")


def _np_ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def shape_tables(shape):
    """(rank, npe, nip, N[nip,npe], dN[nip,npe,rank], w[nip]) of a shape id as the kernels use them (host only)."""
    r, n, q = C.c_int32(), C.c_int32(), C.c_int32()
    _check(lib().fa_shape_info(shape, C.byref(r), C.byref(n), C.byref(q)))
    N = np.empty((q.value, n.value)); dN = np.empty((q.value, n.value, r.value)); w = np.empty(q.value)
    _check(lib().fa_shape_tables(shape, _np_ptr(N), _np_ptr(dN), _np_ptr(w)))
    return r.value, n.value, q.value, N, dN, w


class Assembler:
    """One model instance on one GPU (the object a Jive shim would own)."""

    def __init__(self, shape, model, material, coords, conn, *, state=PLANE_STRAIN, young=210e3, poisson=0.3,
                 frac_type=AT2, gc=2.7, l0=0.015, sigma_t=0.0, beta=6500.0, keep_off_diags=False,
                 row_ptr=None, col_idx=None, node_dofs=None, ndof=0, pattern_kind=PATTERN_FULL, device=0,
                 softening=0, kappa_i=0.0, kappa_c=0.0, soft_alpha=0.0, soft_beta=0.0, length_scale=0.0, eqv_strain=0,
                 soft_b=0.0, soft_s=0.0, kappa_p=0.0, eta=0.0):
        coords = np.ascontiguousarray(coords, dtype=np.float64)
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        rank, npe, nip = SHAPE_INFO[shape]
        if coords.ndim != 2 or coords.shape[1] != rank:
            raise FaError(f"shape has invalid rank: {coords.shape[1] if coords.ndim == 2 else '?'} (should be {rank})")
        if conn.ndim != 2 or (conn.size and conn.shape[1] != npe):
            raise FaError(f"elements must have {npe} nodes")
        self._keep = [coords, conn]
        if row_ptr is not None:
            row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
            col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
        if node_dofs is not None:
            node_dofs = np.ascontiguousarray(node_dofs, dtype=np.int64)
        self._keep += [row_ptr, col_idx, node_dofs]
        p = Problem(shape, model, material, state, young, poisson, frac_type, gc, l0, sigma_t, beta,
                    int(bool(keep_off_diags)), coords.shape[0], _np_ptr(coords), conn.shape[0], _np_ptr(conn),
                    _np_ptr(row_ptr), _np_ptr(col_idx), _np_ptr(node_dofs), ndof, pattern_kind, device,
                    softening, kappa_i, kappa_c, soft_alpha, soft_beta, length_scale, eqv_strain,
                    soft_b, soft_s, kappa_p, eta)
        self.h = C.c_void_p()
        _check(lib().fa_create(C.byref(p), C.byref(self.h)))
        L = lib()
        self.shape, self.model, self.material, self.device = shape, model, material, device
        self.rank, self.npe, self.nip = rank, npe, nip
        self.nnode, self.nel = coords.shape[0], conn.shape[0]
        self.ndof, self.nnz = L.fa_ndof(self.h), L.fa_nnz(self.h)
        self.nip_total, self.ndofpn = L.fa_nip_total(self.h), L.fa_ndof_per_node(self.h)
        self.strain_rows = L.fa_strain_rows(self.h)
        self.uses_row_kernel = bool(L.fa_uses_row_kernel(self.h))
        self.psi0 = L.fa_psi0(self.h)
        red = C.c_double(1.0)
        self.npatch = L.fa_patch_stats(self.h, C.byref(red))
        self.redundancy = red.value

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            lib().fa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except TypeError:      # interpreter shutdown: module globals are already gone
            pass

    def pattern(self):
        rp = np.empty(self.ndof + 1, dtype=np.int64); ci = np.empty(self.nnz, dtype=np.int32)
        _check(lib().fa_pattern_host(self.h, _np_ptr(rp), _np_ptr(ci)))
        return rp, ci

    def row_ptr_device(self):
        """The 64-bit row offsets as a torch tensor on the context's device, without a copy (fa_pattern_device): the view is
        valid while the assembler lives; treat it as read-only."""
        import torch
        rp, ci = C.c_void_p(), C.c_void_p()
        _check(lib().fa_pattern_device(self.h, C.byref(rp), C.byref(ci)))

        class _View:
            __cuda_array_interface__ = {"shape": (self.ndof + 1,), "typestr": "<i8", "data": (rp.value, False), "version": 2}
        return torch.as_tensor(_View(), device=torch.device("cuda", self.device))

    def pattern_device32(self):
        """(row_ptr, col_idx) device addresses of the pattern with 32-bit offsets (cuDSSSolver.cpp:655-663)."""
        rp, ci = C.c_void_p(), C.c_void_p()
        _check(lib().fa_pattern_device32(self.h, C.byref(rp), C.byref(ci)))
        return rp.value, ci.value

    def constraints(self, slave_dofs, rvalues=None, master_ptr=None, master_dofs=None, master_coefs=None):
        return Constraints(self, slave_dofs, rvalues, master_ptr, master_dofs, master_coefs)

    # ---- device path (torch tensors on cuda:device, float64) ----
    def assemble(self, state, K=None, f=None, *, state0=None, state00=None, dt=0.0, dt0=0.0, err_ext=None,
                 flags=ACCUMULATE, stream=None, noalloc=False):
        """noalloc=True goes through fa_assemble_noalloc (static device scratch, no allocation inside the call)."""
        import torch
        def ptr(t, n):
            if t is None:
                return None
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() >= n, "bad device vector"
            return t.data_ptr()
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        a = AssembleArgs(ptr(state, self.ndof), ptr(state0, self.ndof), ptr(state00, self.ndof), dt, dt0,
                         ptr(K, self.nnz), ptr(f, self.ndof), ptr(err_ext, 1), flags, stream)
        _check((lib().fa_assemble_noalloc if noalloc else lib().fa_assemble)(self.h, C.byref(a)))

    # ---- host path (numpy vectors; H2D + assemble + D2H inside the library) ----
    def assemble_host(self, state, K=None, f=None, *, state0=None, state00=None, dt=0.0, dt0=0.0,
                      want_err=False, flags=ACCUMULATE):
        state = np.ascontiguousarray(state, dtype=np.float64)
        assert state.size == self.ndof
        if f is None:
            f = np.zeros(self.ndof)
        s0 = None if state0 is None else np.ascontiguousarray(state0, dtype=np.float64)
        s00 = None if state00 is None else np.ascontiguousarray(state00, dtype=np.float64)
        err = np.zeros(1)
        _check(lib().fa_assemble_host(self.h, _np_ptr(state), _np_ptr(s0), _np_ptr(s00), dt, dt0, _np_ptr(K),
                                      _np_ptr(f), _np_ptr(err) if want_err else None, flags))
        return K, f, float(err[0])

    def strain_table(self, table, weights, stream=None):
        """getStrain_: table[nnode][str] and weights[nnode] (device tensors) are added to from the strain cache."""
        import torch
        assert table.is_cuda and weights.is_cuda and table.dtype == torch.float64 and weights.dtype == torch.float64
        # GradientDamage: str + 1 columns (e_eq last); the others: str
        assert table.is_contiguous() and table.numel() == self.nnode * (self.strain_rows - (0 if self.model in (LINEAR_ELASTICITY, GRADIENT_DAMAGE) else 1))
        assert weights.numel() == self.nnode
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().fa_strain_table(self.h, table.data_ptr(), weights.data_ptr(), stream))

    def stress_table(self, table, weights, stream=None):
        """getStress_: like strain_table, with the stresses of the material update from the cached strain / degradation."""
        import torch
        assert table.is_cuda and weights.is_cuda and table.dtype == torch.float64 and weights.dtype == torch.float64
        assert table.is_contiguous() and table.numel() == self.nnode * (4 if self.rank == 2 else 6) and weights.numel() == self.nnode
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().fa_stress_table(self.h, table.data_ptr(), weights.data_ptr(), stream))

    def damage_table(self, table, weights, stream=None):
        """GradientDamageModel::getDamage_ (GradientDamageModel.cpp:1184-1241): table[nnode] += sum |N| omega(newHist_),
        weights[nnode] += sum |N|."""
        import torch
        assert table.is_cuda and weights.is_cuda and table.dtype == torch.float64 and weights.dtype == torch.float64
        assert table.is_contiguous() and table.numel() == self.nnode and weights.numel() == self.nnode
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().fa_damage_table(self.h, table.data_ptr(), weights.data_ptr(), stream))

    def output_data(self, cols, table, weights, stream=None):
        """getOutputData_ (PhaseFractureModel.cpp:1408-1508): columns `cols` of [strain.tensor | stress.tensor]."""
        import torch
        cols = np.ascontiguousarray(cols, dtype=np.int32)
        assert table.is_cuda and weights.is_cuda and table.dtype == torch.float64 and weights.dtype == torch.float64
        assert table.is_contiguous() and table.numel() == self.nnode * cols.size and weights.numel() == self.nnode
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().fa_output_data(self.h, cols.size, _np_ptr(cols), table.data_ptr(), weights.data_ptr(), stream))

    def arc_func(self, state, state0, jac10, fe_dissipation=None, stream=None):
        """getArcFunc_ (PhaseFractureModel.cpp:844-1063): fills jac10, returns the device scalar FE_DISSIPATION."""
        import torch
        for t in (state, state0, jac10):
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() >= self.ndof
        if fe_dissipation is None:
            fe_dissipation = torch.empty(1, dtype=torch.float64, device=state.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().fa_arc_func(self.h, state.data_ptr(), state0.data_ptr(), jac10.data_ptr(), fe_dissipation.data_ptr(), stream))
        return fe_dissipation

    def gather_sum(self, vec, idx, out=None, stream=None):
        """sum of vec[idx] (idx < 0 skipped) on the device: LodiModel's load / disp sums (LodiModel.cpp:263-283)."""
        import torch
        assert vec.is_cuda and vec.dtype == torch.float64 and idx.is_cuda and idx.dtype == torch.int64 and idx.is_contiguous()
        if out is None:
            out = torch.empty(1, dtype=torch.float64, device=vec.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(lib().fa_gather_sum(self.h, vec.data_ptr(), idx.data_ptr(), idx.numel(), out.data_ptr(), stream))
        return out

    def commit(self):
        _check(lib().fa_commit(self.h))

    def get_history(self, which):
        out = np.empty(self.nip_total)
        _check(lib().fa_get_history(self.h, which, _np_ptr(out)))
        return out

    def set_history(self, which, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert v.size == self.nip_total
        _check(lib().fa_set_history(self.h, which, _np_ptr(v)))

    def set_interface(self, node_ranges):
        """node_ranges: [(begin, end), ...] of interface nodes; afterwards assemble(..., flags | PART_INTERFACE / PART_INTERIOR)
        issues a pass in two launches (fa_set_interface)."""
        r = np.ascontiguousarray(np.asarray(node_ranges, dtype=np.int64).reshape(-1, 2))
        _check(lib().fa_set_interface(self.h, r.shape[0], _np_ptr(r)))

    def micro_stale_hits(self):
        """Gauss points of the last assemble call in the `d > 1` branch of MicroPhaseFractureModel.cpp:772-776 (0 for every
        other model): where the reference keeps a stale dphidStrain and the kernels use 0."""
        n = C.c_int64(0)
        _check(lib().fa_micro_stale_hits(self.h, C.byref(n)))
        return n.value

    def get_strain_cache(self):
        out = np.empty((self.nip_total, self.strain_rows))
        _check(lib().fa_get_strain_cache(self.h, _np_ptr(out)))
        return out


class Constraints:
    """A constraint set on an Assembler's pattern (jive::util::Constraints as the solver wrappers' StdConstrainer and
    BFGSModule::updateResidual use it): u[slave] = rvalue + sum coef * u[master]."""

    def __init__(self, asm, slave_dofs, rvalues=None, master_ptr=None, master_dofs=None, master_coefs=None):
        self.asm = asm
        sd = np.ascontiguousarray(slave_dofs, dtype=np.int64)
        rv = None if rvalues is None else np.ascontiguousarray(rvalues, dtype=np.float64)
        mp = None if master_ptr is None else np.ascontiguousarray(master_ptr, dtype=np.int64)
        md = None if master_dofs is None else np.ascontiguousarray(master_dofs, dtype=np.int64)
        mc = None if master_coefs is None else np.ascontiguousarray(master_coefs, dtype=np.float64)
        assert rv is None or rv.size == sd.size
        assert mp is None or mp.size == sd.size + 1
        self.nslave = sd.size
        self.h = C.c_void_p()
        _check(lib().fa_cons_create(asm.h, sd.size, _np_ptr(sd), _np_ptr(rv), _np_ptr(mp), _np_ptr(md), _np_ptr(mc),
                                    C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            lib().fa_cons_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except TypeError:
            pass

    def _stream(self, stream):
        import torch
        return torch.cuda.current_stream(self.asm.device).cuda_stream if stream is None else stream

    def _ptr(self, t, n):
        import torch
        if t is None:
            return None
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() >= n, "bad device vector"
        return t.data_ptr()

    def set_rvalues(self, rvalues, stream=None):
        rv = np.ascontiguousarray(rvalues, dtype=np.float64)
        assert rv.size == self.nslave
        _check(lib().fa_cons_set_rvalues(self.h, _np_ptr(rv), self._stream(stream)))

    def eliminate(self, K, rhs=None, stream=None):
        """K := T^t K T + I_S, rhs := T^t (rhs - K c), in place (StdConstrainer, cuDSSSolver.cpp:108-121,293)."""
        _check(lib().fa_cons_eliminate(self.h, self._ptr(K, self.asm.nnz), self._ptr(rhs, self.asm.ndof), self._stream(stream)))

    @property
    def fill_in(self):
        """entries of the constrained matrix outside the assembler's pattern (0: eliminate() works in place)"""
        return int(lib().fa_cons_fill_in(self.h))

    @property
    def nnz(self):
        return int(lib().fa_cons_nnz(self.h))

    def pattern(self):
        """host copy (row_ptr int64, col_idx int32) of the output pattern: the assembler's plus the fill-in"""
        rp = np.empty(self.asm.ndof + 1, dtype=np.int64); ci = np.empty(max(1, self.nnz), dtype=np.int32)
        _check(lib().fa_cons_pattern_host(self.h, _np_ptr(rp), _np_ptr(ci)))
        return rp, ci[:self.nnz]

    def pattern_device32(self):
        rp = C.c_void_p(); ci = C.c_void_p()
        _check(lib().fa_cons_pattern_device32(self.h, C.byref(rp), C.byref(ci)))
        return rp.value, ci.value

    def eliminate_out(self, K, rhs=None, out=None, stream=None):
        """K_out := T^t K T + I_S on the output pattern (fa_cons_nnz entries), rhs := T^t (rhs - K c) in place; returns K_out."""
        import torch
        if out is None:
            out = torch.empty(self.nnz, dtype=torch.float64, device=K.device)
        _check(lib().fa_cons_eliminate_out(self.h, self._ptr(K, self.asm.nnz), self._ptr(out, self.nnz), self._ptr(rhs, self.asm.ndof),
                                           self._stream(stream)))
        return out

    def eval_masters(self, vec, stream=None):
        _check(lib().fa_cons_eval_masters(self.h, self._ptr(vec, self.asm.ndof), self._stream(stream)))

    def eval_slaves(self, vec, stream=None):
        _check(lib().fa_cons_eval_slaves(self.h, self._ptr(vec, self.asm.ndof), self._stream(stream)))

    def residual(self, fext, fint, du, r, *, u=None, lower=None, upper=None, r2=None, out=None, stream=None):
        """BFGSModule::Work_::updateResidual (BFGSModule.cpp:438-489); returns the device tensor {rnorm, enorm, rnorm^2}."""
        import torch
        n = self.asm.ndof
        if out is None:
            out = torch.empty(3, dtype=torch.float64, device=r.device)
        _check(lib().fa_residual(self.h, self._ptr(fext, n), self._ptr(fint, n), self._ptr(du, n), self._ptr(u, n),
                                 self._ptr(lower, n), self._ptr(upper, n), self._ptr(r, n), self._ptr(r2, n),
                                 self._ptr(out, 3), self._stream(stream)))
        return out
