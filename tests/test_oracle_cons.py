"""CPU: the oracle's constraint elimination / residual restatement (oracle/falcon_oracle_cons.c) against dense numpy
algebra -- (T^t K T + I_S), T^t (r - K c) formed with explicit matrices -- on small random systems.  PARITY UNPINNED
against Jive (the Constraints / StdConstrainer classes are not in the reference checkout): what is pinned here is that
the CSR restatement implements the documented algebra."""
import numpy as np
import pytest

from oracle import pyoracle as po


def random_system(n, seed, dense=True):
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, n))
    if not dense:
        A[np.abs(A) < 0.8] = 0.0
        A += np.diag(rng.uniform(1, 2, n))
    rp = np.zeros(n + 1, dtype=np.int64); ci = []; val = []
    for i in range(n):
        cols = np.arange(n) if dense else np.nonzero(A[i])[0]
        ci += list(cols); val += list(A[i, cols]); rp[i + 1] = len(ci)
    return A, rp, np.array(ci, dtype=np.int32), np.array(val), rng


def dense_T(n, slave, rval, mptr, mdof, mcoef):
    T = np.eye(n); c = np.zeros(n)
    for k, s in enumerate(slave):
        T[s, s] = 0.0
        c[s] = 0.0 if rval is None else rval[k]
        if mptr is not None:
            for q in range(mptr[k], mptr[k + 1]):
                T[s, mdof[q]] += mcoef[q]
    return T, c


CASES = {
    "dirichlet": dict(slave=[0, 5, 6, 11], rval=[0.0, 0.3, -0.2, 1.5]),
    "single_master": dict(slave=[1, 4, 9], rval=[0.1, 0.0, -0.4], mptr=[0, 1, 2, 3], mdof=[2, 2, 7], mcoef=[1.0, -0.5, 2.0]),
    "multi_master": dict(slave=[3, 8, 10], rval=None, mptr=[0, 2, 2, 5], mdof=[0, 1, 0, 5, 6], mcoef=[0.5, 0.5, 0.25, 0.25, 0.5]),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_eliminate_matches_dense_algebra(name):
    n = 12
    cs = CASES[name]
    A, rp, ci, val, rng = random_system(n, 3)
    rhs = rng.normal(size=n)
    cons = po.Constraints(cs["slave"], cs.get("rval"), cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
    T, c = dense_T(n, cs["slave"], cs.get("rval"), cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
    Kc = T.T @ A @ T
    for s in cs["slave"]:
        Kc[s, s] += 1.0
    rc = T.T @ (rhs - A @ c)
    K = val.copy(); r = rhs.copy()
    cons.eliminate(rp, ci, K, r)
    assert np.abs(K.reshape(n, n) - Kc).max() < 1e-13 * np.abs(Kc).max()
    assert np.abs(r - rc).max() < 1e-13 * np.abs(rc).max()
    # the constrained solution satisfies the constraints and the reduced equations: solve, then evalSlaveDofs
    du = np.linalg.solve(Kc, rc)
    cons.eval_slaves(du)
    full = T @ du + c
    assert np.abs(full - du).max() < 1e-12 * np.abs(du).max()
    free = np.setdiff1d(np.arange(n), cs["slave"])
    res = T.T @ (rhs - A @ du)
    assert np.abs(res[free]).max() < 1e-10 * np.abs(rhs).max()


def test_eliminate_reports_missing_fill_in():
    n = 12
    A, rp, ci, val, rng = random_system(n, 5, dense=False)
    cs = CASES["single_master"]
    cons = po.Constraints(cs["slave"], cs.get("rval"), cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
    with pytest.raises(ValueError):
        cons.eliminate(rp, ci, val.copy(), np.zeros(n))
    # plain Dirichlet conditions never need fill-in
    d = CASES["dirichlet"]
    K = val.copy(); r = rng.normal(size=n); r0 = r.copy()
    po.Constraints(d["slave"], d["rval"]).eliminate(rp, ci, K, r)
    T, c = dense_T(n, d["slave"], d["rval"], None, None, None)
    Kc = T.T @ A @ T + np.diag(1.0 - np.diag(T))
    dense = np.zeros((n, n))
    for i in range(n):
        dense[i, ci[rp[i]:rp[i + 1]]] = K[rp[i]:rp[i + 1]]
    assert np.abs(dense - Kc).max() < 1e-13 * np.abs(Kc).max()
    assert np.abs(r - T.T @ (r0 - A @ c)).max() < 1e-13 * np.abs(r0).max()


@pytest.mark.parametrize("name", sorted(CASES))
def test_eval_masters_and_slaves(name):
    n = 12
    cs = CASES[name]
    rng = np.random.default_rng(8)
    cons = po.Constraints(cs["slave"], cs.get("rval"), cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
    T, c = dense_T(n, cs["slave"], cs.get("rval"), cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
    v = rng.normal(size=n)
    assert np.abs(cons.eval_masters(v.copy()) - T.T @ v).max() < 1e-14
    assert np.abs(cons.eval_slaves(v.copy()) - (T @ v + c)).max() < 1e-14


@pytest.mark.parametrize("bounded", [False, True])
def test_residual_matches_the_reference_loop(bounded):
    """BFGSModule.cpp:438-489 written out in numpy."""
    n = 40
    rng = np.random.default_rng(4)
    cs = CASES["single_master"]
    cons = po.Constraints(cs["slave"], cs.get("rval"), cs.get("mptr"), cs.get("mdof"), cs.get("mcoef"))
    T, _ = dense_T(n, cs["slave"], None, cs["mptr"], cs["mdof"], cs["mcoef"])
    fext, fint, du, u = (rng.normal(size=n) for _ in range(4))
    lower = np.where(rng.random(n) < 0.5, u, u - 1.0); upper = np.where(rng.random(n) < 0.5, u, u + 1.0)
    lower[7] = upper[7] = u[7]                          # lower == upper: both branches taken (:465-476)
    r_exp = T.T @ (fext - fint)
    if bounded:
        eps = np.sqrt(np.finfo(float).eps)
        f = -r_exp.copy()
        lb = lower + eps * np.abs(lower); ub = upper - eps * np.abs(upper)
        f = np.where(u <= lb, np.minimum(0.0, f), f)
        f = np.where(u >= ub, np.maximum(0.0, f), f)
        r, r2, out = cons.residual(fext, fint, du, u, lower, upper)
        assert np.abs(r2 - f).max() < 1e-14
        vec = f
    else:
        r, r2, out = cons.residual(fext, fint, du)
        vec = r_exp
    assert np.abs(r - r_exp).max() < 1e-14
    assert abs(out[0] - np.linalg.norm(vec)) < 1e-13 * np.linalg.norm(vec)
    assert abs(out[1] - vec @ du) < 1e-13 * np.abs(vec * du).sum()
    assert abs(out[2] - out[0] ** 2) < 1e-13 * out[2]
