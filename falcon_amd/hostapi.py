"""ctypes access to libfalcon_host.so: the C++ mirror of the reference's plugin interface (falcon_amd/host).
A `Case` is what Jive's MatrixModel + solver see: a model tree built from a .pro / .mesh pair, driven only
through takeAction(action, params, globdat)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfalcon_host.so")
_lib = None


class HostError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        from falcon_amd.capi import _torch_first
        _torch_first()                       # see capi._torch_first: torch's HIP runtime has to come up first
        L = C.CDLL(LIB_PATH)
        vp, dp = C.c_void_p, C.POINTER(C.c_double)
        L.fh_create.argtypes = [C.c_char_p, C.c_char_p]; L.fh_create.restype = vp
        L.fh_destroy.argtypes = [vp]; L.fh_destroy.restype = None
        L.fh_last_error.restype = C.c_char_p
        L.fh_ndof.argtypes = [vp]; L.fh_ndof.restype = C.c_int64
        L.fh_nnz.argtypes = [vp]; L.fh_nnz.restype = C.c_int64
        L.fh_model_type.argtypes = [vp]; L.fh_model_type.restype = C.c_char_p
        L.fh_pattern.argtypes = [vp, vp, vp]; L.fh_pattern.restype = None
        L.fh_dof_index.argtypes = [vp, C.c_int64, C.c_char_p]; L.fh_dof_index.restype = C.c_int64
        L.fh_constraint_count.argtypes = [vp]; L.fh_constraint_count.restype = C.c_int64
        L.fh_constraints.argtypes = [vp, vp, vp, vp, vp, C.c_int64]; L.fh_constraints.restype = C.c_int64
        L.fh_take_action.argtypes = [vp, C.c_char_p, vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Case:
    def __init__(self, pro_text, mesh_text):
        self.h = lib().fh_create(pro_text.encode(), mesh_text.encode())
        if not self.h:
            raise HostError(lib().fh_last_error().decode())
        self.ndof, self.nnz = lib().fh_ndof(self.h), lib().fh_nnz(self.h)
        self.model_type = lib().fh_model_type(self.h).decode()

    def close(self):
        if getattr(self, "h", None):
            lib().fh_destroy(self.h)
            self.h = None

    __del__ = close

    def pattern(self):
        rp = np.empty(self.ndof + 1, dtype=np.int64); ci = np.empty(self.nnz, dtype=np.int32)
        lib().fh_pattern(self.h, _p(rp), _p(ci))
        return rp, ci

    def constraints(self):
        """(slaves, master_ptr, masters, coefs) of the Constraints the models have added (GradientDamage's INIT), in the
        layout fa_cons_create takes."""
        n = lib().fh_constraint_count(self.h)
        slaves = np.zeros(n, dtype=np.int64); mp = np.zeros(n + 1, dtype=np.int64)
        masters = np.zeros(2 * n + 1, dtype=np.int64); coefs = np.zeros(2 * n + 1)
        if n and lib().fh_constraints(self.h, _p(slaves), _p(mp), _p(masters), _p(coefs), 2 * n) != n:
            raise HostError("constraints with more than two masters")
        return slaves, mp, masters[:mp[-1]], coefs[:mp[-1]]

    def dof(self, node_id, dof_type):
        return lib().fh_dof_index(self.h, node_id, dof_type.encode())

    def take_action(self, action, state=None, state0=None, state00=None, step=0.0, step0=0.0, iter_count=0):
        """-> (handled, K_values, f_int); K and f are zeroed before the action like Jive's solver does."""
        arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64) for a in (state, state0, state00)]
        K = np.zeros(self.nnz); f = np.zeros(self.ndof)
        rc = lib().fh_take_action(self.h, action.encode(), _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), step, step0,
                                  iter_count, _p(K), _p(f))
        if rc < 0:
            raise HostError(lib().fh_last_error().decode())
        return bool(rc), K, f
