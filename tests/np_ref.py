"""Independent numpy restatement (dense B, einsum) of the PhaseFracture / LinearElasticity element
matrices, used only to cross-check the C oracle on small meshes.  Follows
PhaseFractureModel.cpp:629-731 and LinearElasticityModel.cpp:389-422 of the reference."""
import numpy as np


def B_matrix(g):                      # g: (npe, rank) -> (str, rank*npe); ShapeUtils.cpp:73-145
    npe, rank = g.shape
    if rank == 2:
        b = np.zeros((4, 2 * npe))
        b[0, 0::2] = g[:, 0]; b[1, 1::2] = g[:, 1]; b[3, 0::2] = g[:, 1]; b[3, 1::2] = g[:, 0]
    else:
        b = np.zeros((6, 3 * npe))
        b[0, 0::3] = g[:, 0]; b[1, 1::3] = g[:, 1]; b[2, 2::3] = g[:, 2]
        b[3, 0::3] = g[:, 1]; b[3, 1::3] = g[:, 0]
        b[4, 1::3] = g[:, 2]; b[4, 2::3] = g[:, 1]
        b[5, 2::3] = g[:, 0]; b[5, 0::3] = g[:, 2]
    return b


def hooke(rank, state, E, nu):        # HookeMaterial.cpp:202-267
    n = 4 if rank == 2 else 6
    D = np.zeros((n, n))
    if rank == 3 or state == 0:
        a = E / ((1 + nu) * (1 - 2 * nu))
        D[:3, :3] = a * nu
        D[0, 0] = D[1, 1] = D[2, 2] = a * (1 - nu)
        for k in range(3, n):
            D[k, k] = a * 0.5 * (1 - 2 * nu)
    else:
        a = E / (1 - nu * nu)
        D[0, 0] = D[1, 1] = a; D[0, 1] = D[1, 0] = a * nu; D[3, 3] = a * 0.5 * (1 - nu)
    return D


def amor(rank, E, nu, eps, g):        # AmorPhaseMaterial.cpp:200-232
    n = eps.size
    K = E / (3 * (1 - 2 * nu)); G = E / (2 * (1 + nu))
    PV = np.zeros((n, n)); PV[:3, :3] = 1.0
    PD = np.zeros((n, n)); PD[:3, :3] = -1.0 / 3.0
    for i in range(3):
        PD[i, i] = 2.0 / 3.0
    for i in range(3, n):
        PD[i, i] = 0.5
    tr = eps[:3].sum()
    D = 2 * G * PD + (K * PV if tr > 0 else 0)
    sp = D @ eps
    psi = 0.5 * sp @ eps
    D = g * D + (K * PV if tr <= 0 else 0)
    return D @ eps, D, psi, sp


def bourdin(rank, state, E, nu, eps, g):   # BourdinPhaseMaterial.cpp:131-145
    D = hooke(rank, state, E, nu)
    sp = D @ eps
    return g * sp, g * D, 0.5 * sp @ eps, sp


def element_pf(x, N, dN, w, u, phi, H0, mat, frac, keep):
    """x (npe,rank); N (nip,npe); dN (nip,npe,rank); returns K1,K2,K3,K4,f1,f2,Hnew."""
    npe, rank = x.shape
    nd = rank * npe
    K1 = np.zeros((nd, nd)); K2 = np.zeros((nd, npe)); K3 = np.zeros((npe, nd)); K4 = np.zeros((npe, npe))
    f1 = np.zeros(nd); f2 = np.zeros(npe); Hn = np.zeros(len(w))
    gc, l0, cw, eta, p, a1, a2, a3 = frac
    for ip in range(len(w)):
        J = x.T @ dN[ip]
        g = dN[ip] @ np.linalg.inv(J)
        wip = w[ip] * np.linalg.det(J)
        B = B_matrix(g)
        eps = B @ u
        pf = N[ip] @ phi
        gn = (1 - pf) ** p; gd = gn + a1 * pf + a1 * a2 * pf ** 2 + a1 * a2 * a3 * pf ** 3
        gphi = gn / gd + 1e-7
        dgn = -p * (1 - pf) ** (p - 1); dgd = dgn + a1 + 2 * a1 * a2 * pf + 3 * a1 * a2 * a3 * pf ** 2
        ddgn = p * (p - 1) * (1 - pf) ** (p - 2); ddgd = ddgn + 2 * a1 * a2 + 6 * a1 * a2 * a3 * pf
        dg = (dgn * gd - gn * dgd) / gd ** 2
        ddg = ((ddgn * gd - gn * ddgd) * gd - 2 * (dgn * gd - gn * dgd) * dgd) / gd ** 3
        dw = eta + 2 * (1 - eta) * pf; ddw = 2 * (1 - eta)
        sig, D, psi, sp = mat(eps, gphi)
        loading = 0.0 if psi < H0[ip] else 1.0
        psi = max(psi, H0[ip]); Hn[ip] = psi
        Nip = N[ip]
        K1 += wip * B.T @ D @ B
        K4 += wip * ((gc / (cw * l0) * ddw + ddg * psi) * np.outer(Nip, Nip) + (2 * gc * l0 / cw) * g @ g.T)
        if keep:
            K2 += wip * dg * B.T @ np.outer(sp, Nip)
            K3 += wip * loading * dg * np.outer(Nip, sp @ B)
        f1 += wip * B.T @ sig
        f2 += wip * (Nip * (gc / (cw * l0) * dw + dg * psi) + (2 * gc * l0 / cw) * (g @ g.T) @ phi)
    return K1, K2, K3, K4, f1, f2, Hn


# ---- MiehePhaseMaterial with numpy.linalg.eigh (NOT the cyclic Jacobi the oracle and the kernels share) ----
# Full 3x3x3x3 tensors exactly as the reference writes them: getposProj_ (MiehePhaseMaterial.cpp:342-406),
# update (:200-258), otimes / otimesu / otimesl / doubleDot (TensorUtils.cpp:73-231), tensor2voigt (:304-398, with
# its duplicated-term shear entries restated verbatim), voigt2tensorStrain (:467-492), tensor2voigtStress (:436-460).
def _voigt_pairs(n):
    return [(0, 0), (1, 1), (2, 2), (0, 1)] + ([(1, 2), (2, 0)] if n == 6 else [])


def tensor2voigt(A, n):
    M = np.zeros((n, n))
    pr = _voigt_pairs(n)
    for I, (i, j) in enumerate(pr):
        for J, (k, l) in enumerate(pr):
            if I < 3 and J < 3:
                M[I, J] = A[i, j, k, l]
            elif I >= 3 and J < 3:
                M[I, J] = 0.5 * (A[i, j, k, l] + A[j, i, k, l])
            elif I < 3 and J >= 3:
                M[I, J] = 0.5 * (A[i, j, k, l] + A[i, j, l, k])
            elif n == 4:
                M[I, J] = 0.25 * (A[0, 1, 0, 1] + A[1, 0, 0, 1] + A[0, 1, 1, 0] + A[1, 0, 1, 0])
            else:
                # TensorUtils.cpp:366-397 as written: some entries repeat two terms instead of using all four
                dup = {(3, 3): [(0, 1, 0, 1), (1, 0, 1, 0), (0, 1, 0, 1), (1, 0, 1, 0)],
                       (4, 3): [(1, 2, 0, 1), (2, 1, 1, 0), (2, 1, 0, 1), (1, 2, 1, 0)],
                       (5, 3): [(0, 2, 0, 1), (2, 0, 1, 0), (2, 0, 0, 1), (0, 2, 1, 0)],
                       (3, 4): [(0, 1, 1, 2), (1, 0, 2, 1), (0, 1, 1, 2), (1, 0, 2, 1)],
                       (4, 4): [(1, 2, 1, 2), (2, 1, 2, 1), (2, 1, 1, 2), (1, 2, 2, 1)],
                       (5, 4): [(0, 2, 1, 2), (2, 0, 2, 1), (2, 0, 1, 2), (0, 2, 2, 1)],
                       (3, 5): [(0, 1, 0, 2), (1, 0, 2, 0), (0, 1, 0, 2), (1, 0, 2, 0)],
                       (4, 5): [(1, 2, 0, 2), (2, 1, 2, 0), (2, 1, 0, 2), (1, 2, 2, 0)],
                       (5, 5): [(0, 2, 0, 2), (2, 0, 2, 0), (2, 0, 0, 2), (0, 2, 2, 0)]}[(I, J)]
                M[I, J] = 0.25 * sum(A[t] for t in dup)
    return M


def miehe(rank, E, nu, eps, g, eigh=np.linalg.eigh):
    """returns sigma, D, Psi, sigma+ (Voigt, STRAIN_COUNTS[rank] entries) and the eigenvalues."""
    n = eps.size
    K = E / (3 * (1 - 2 * nu)); G = E / (2 * (1 + nu)); lam = K - 0.666666667 * G     # MiehePhaseMaterial.cpp:97-99
    T = np.zeros((3, 3))
    T[0, 0], T[1, 1], T[2, 2] = eps[0], eps[1], eps[2]
    T[0, 1] = T[1, 0] = 0.5 * eps[3]
    if n == 6:
        T[1, 2] = T[2, 1] = 0.5 * eps[4]; T[2, 0] = T[0, 2] = 0.5 * eps[5]
    val, vec = eigh(T)
    epos = 0.5 * (val + np.abs(val)); diag = (val > 0.0).astype(float)
    P = np.zeros((3, 3, 3, 3))
    M = [np.outer(vec[:, a], vec[:, a]) for a in range(3)]
    for a in range(3):
        P += np.einsum("ij,kl->ijkl", M[a], M[a]) * diag[a]
    for a in range(3):
        for b in range(a):
            Gab = np.einsum("ik,jl->ijkl", M[a], M[b]) + np.einsum("il,jk->ijkl", M[a], M[b])
            Gba = np.einsum("ik,jl->ijkl", M[b], M[a]) + np.einsum("il,jk->ijkl", M[b], M[a])
            if abs(val[a] - val[b]) <= 1e-13:
                th = 0.5 * (diag[a] + diag[b]) / 2.0
            else:
                th = 0.5 * (epos[a] - epos[b]) / (val[a] - val[b])
            P += th * (Gab + Gba)
    I2 = np.eye(3)
    I4 = 0.5 * (np.einsum("ik,jl->ijkl", I2, I2) + np.einsum("il,jk->ijkl", I2, I2))
    N4 = I4 - P
    Tp = np.einsum("ijkl,kl->ij", P, T); Tn = T - Tp
    tr = np.trace(T)
    trp = 0.5 * (tr + abs(tr)); trn = 0.5 * (tr - abs(tr))
    sp_t = lam * trp * I2 + 2 * G * Tp; sn_t = lam * trn * I2 + 2 * G * Tn
    tv = lambda A: np.array([A[0, 0], A[1, 1], A[2, 2], 0.5 * (A[0, 1] + A[1, 0])] +
                            ([0.5 * (A[1, 2] + A[2, 1]), 0.5 * (A[0, 2] + A[2, 0])] if n == 6 else []))
    sP, sN = tv(sp_t), tv(sn_t)
    psi = 0.5 * lam * trp ** 2 + G * np.einsum("ij,ij->", Tp, Tp)
    sgn = 0.0 if tr < 0.0 else 1.0                                # heavisideP(0) = 1 (MathUtils.cpp:33-36)
    II = np.einsum("ij,kl->ijkl", I2, I2)
    jac = g * (2 * G * P + lam * sgn * II) + (2 * G * N4 + lam * (1 - sgn) * II)
    return g * sP + sN, tensor2voigt(jac, n), psi, sP, val


def degrade(pf, p, a1, a2, a3, floor):
    gn = (1 - pf) ** p; gd = gn + a1 * pf + a1 * a2 * pf ** 2 + a1 * a2 * a3 * pf ** 3
    dgn = -p * (1 - pf) ** (p - 1); dgd = dgn + a1 + 2 * a1 * a2 * pf + 3 * a1 * a2 * a3 * pf ** 2
    ddgn = p * (p - 1) * (1 - pf) ** (p - 2); ddgd = ddgn + 2 * a1 * a2 + 6 * a1 * a2 * a3 * pf
    dg = (dgn * gd - gn * dgd) / gd ** 2
    ddg = ((ddgn * gd - gn * ddgd) * gd - 2 * (dgn * gd - gn * dgd) * dgd) / gd ** 3
    return gn / gd + floor, dg, ddg


def element_ext(x, N, dN, w, u, phi, phi0, phiX, dt, dt0, mat, frac, psi0, beta, extit):
    """PhaseFractureExtModel::getMatrix_ (PhaseFractureExtModel.cpp:640-774) / ...ExtItModel (:660-797): degradation
    at the extrapolated phase field, Psi = max(Psi, Psi0), K3 without the loading flag, no K2, penalty on phi0 - phi.
    phiX = state00 (Ext) or stateExt_ (ExtIt).  Returns K1, K3, K4, f1, f2, errExt."""
    npe, rank = x.shape
    nd = rank * npe
    K1 = np.zeros((nd, nd)); K3 = np.zeros((npe, nd)); K4 = np.zeros((npe, npe))
    f1 = np.zeros(nd); f2 = np.zeros(npe); err = 0.0
    gc, l0, cw, eta, p, a1, a2, a3 = frac
    for ip in range(len(w)):
        J = x.T @ dN[ip]
        g = dN[ip] @ np.linalg.inv(J)
        wip = w[ip] * np.linalg.det(J)
        B = B_matrix(g)
        eps = B @ u
        Nip = N[ip]
        pf = Nip @ phi; pf0 = Nip @ phi0
        if extit:
            pfEx = Nip @ phiX
            err += (pf - pfEx) ** 2
        elif dt0 < 1e-15:
            pfEx = pf0
        else:
            pfEx = -(Nip @ phiX) * dt / dt0 + pf0 * (dt + dt0) / dt0
        pfEx = min(max(pfEx, 0.0), 1.0)
        gEx = degrade(pfEx, p, a1, a2, a3, 1e-7)[0]
        _, dg, ddg = degrade(pf, p, a1, a2, a3, 1e-7)
        dw = eta + 2 * (1 - eta) * pf; ddw = 2 * (1 - eta)
        sig, D, psi, sp = mat(eps, gEx)
        psi = max(psi, psi0)
        K1 += wip * B.T @ D @ B
        K4 += wip * ((gc / (cw * l0) * ddw + ddg * psi) * np.outer(Nip, Nip) + (2 * gc * l0 / cw) * g @ g.T)
        K3 += wip * dg * np.outer(Nip, sp @ B)
        f1 += wip * B.T @ sig
        f2 += wip * (Nip * (gc / (cw * l0) * dw + dg * psi) + (2 * gc * l0 / cw) * (g @ g.T) @ phi)
        if pf0 - pf > 1e-15:
            K4 += wip * gc / (cw * l0) * beta * np.outer(Nip, Nip)
            f2 -= wip * gc / (cw * l0) * beta * (pf0 - pf) * Nip
    return K1, K3, K4, f1, f2, err


def micro_newton(psi, target, d, frac, alpha):
    """Local damage of the micromorphic model (MicroPhaseFractureModel.cpp:697-732): Newton on
    Res(d) = g'(d) Psi + gc/(cw l0) w'(d) + alpha (d - target), |Res| <= 1e-4 or 100 iterations."""
    gc, l0, cw, eta, p, a1, a2, a3 = frac
    for it in range(101):
        _, dg, ddg = degrade(d, p, a1, a2, a3, 0.0)
        res = dg * psi + gc / (cw * l0) * (eta + 2 * (1 - eta) * d) + alpha * (d - target)
        jac = ddg * psi + gc / (cw * l0) * 2 * (1 - eta) + alpha
        if abs(res) <= 1e-4 or it == 100:
            return d, dg, ddg
        d -= res / jac


# ---- GradientDamageModel::getMatrix_ (GradientDamageModel.cpp:368-588) with DamageMaterial (eqvStrain = Energy) ----
def damage_law(soft, Ki, Kc, al, be, K):
    """(omega, dOmega/dKappa): DamageMaterial.cpp:1044-1633, numbered like softeningLawDefs (:134-144)."""
    if soft == 0:
        return (0.0 if K < Ki else 1 - Ki / K), (0.0 if K < Ki else Ki / K ** 2)
    if soft == 1:
        om = 0.0 if K < Ki else (1.0 if K > Kc else Kc * (K - Ki) / (K * (Kc - Ki)))
        return (0.999 if om == 1.0 else om), (Ki * Kc / ((Kc - Ki) * K ** 2) if K < Kc else 0.0)
    if soft == 2:
        return (0.0 if K < Ki else (0.999 if K > Kc else (K - Ki) / (Kc - Ki))), (1 / (Kc - Ki) if K < Kc else 0.0)
    if soft == 3:
        ex = np.exp(-be * (K - Ki))
        return (0.0 if K < Ki else 1 - Ki / K * (1 - al + al * ex)), Ki / K ** 2 * (1 - al + al * ex) + be * Ki / K * al * ex
    if soft == 4:
        ex = np.exp((Ki - K) / (Kc - Ki))
        return (0.0 if K < Ki else 1 - Ki / K * ex), Ki / K ** 2 * ex + Ki / K * ex / (Kc - Ki)
    # the laws of :1313-1633; `al` carries the single parameter of hyperbolic (b), expoEnergy (s), multilinear (kappaP)
    if soft == 5:                                          # power, Kcc = Kc
        if K > Kc:
            return (1 - (Ki / Kc) ** be * 0.0 ** al), 0.0
        om = 1 - (Ki / K) ** be * ((Kc - K) / (Kc - Ki)) ** al
        c_ = (Kc - K) / (Kc - Ki)
        d = (Ki / K) ** (be - 1) * c_ ** (al - 1) * (Ki / K ** 2 * be * c_ + Ki / K * al / (Kc - Ki))
        return (0.0 if K < Ki else om), d
    if soft == 6:                                          # hyperbolic
        q = 1 + al * (K - Ki)
        return (0.0 if K < Ki else 1 - 1 / q), al / q ** 2
    if soft == 7:                                          # exponential2Reg: absent from the overloads in use
        return 0.0, 0.0
    if soft == 8:                                          # expoEnergy
        if K < Ki:
            return 0.0, 0.0
        t = (K - Ki) / Kc
        return 1 - np.exp(-t ** al), np.exp(-t ** al) * t ** (al - 1) * al / Kc
    if soft == 9:                                          # polynomial; the derivative's d uses a = Ki - K un-normalised
        if K < Ki:
            return 0.0, 0.0
        an = (K - Ki) / (Kc - Ki)
        a_ = Ki - K; b_ = Kc - Ki; c_ = a_ / b_; d = 1 + a_ ** 2 * (2 * a_ - 3)
        return 1 - Ki / K * (1 + an ** 2 * (2 * an - 3)), -Ki / K ** 2 * (2 * K * (c_ / b_ * (2 * c_ - 3) + c_ ** 2 / b_) - d)
    Kp = al                                                # multilinear
    if K < Ki:
        return 0.0, 0.0
    if K > Kp:
        return 1 - Ki / K, Ki / K ** 2
    return (0.0 if K < Kp else (0.999 if K > Kc else (K - Kp) / (Kc - Kp))), (1 / (Kc - Kp) if K < Kc else 0.0)


def element_gd(x, N, dN, w, u, eqv, kap0, load0, D0, law, c, want_K=True, mazars=None, vonmises=None):
    """returns K1, K2, K3, K4, f1, f2, kappa, loading.  law = (soft, Ki, Kc, alpha, beta); mazars = (nu, plane_stress)
    for eqvStrain = Mazars, vonmises = (eta, nu) for vonMises (rank 3), both None for Energy."""
    npe, rank = x.shape
    nd = rank * npe
    K1 = np.zeros((nd, nd)); K2 = np.zeros((nd, npe)); K3 = np.zeros((npe, nd)); K4 = np.zeros((npe, npe))
    f1 = np.zeros(nd); f2 = np.zeros(npe); kap = np.zeros(len(w)); ld = np.zeros(len(w), dtype=int)
    for ip in range(len(w)):
        J = x.T @ dN[ip]
        g = dN[ip] @ np.linalg.inv(J)
        wip = w[ip] * np.linalg.det(J)
        B = B_matrix(g)
        eps = B @ u
        Nip = N[ip]
        e_nl = Nip @ eqv
        f = e_nl - kap0[ip]
        if f == 0.0:
            kap[ip], ld[ip] = kap0[ip], load0[ip]
        elif f > 0.0:
            kap[ip], ld[ip] = e_nl, int(e_nl >= law[1])
        else:
            kap[ip], ld[ip] = kap0[ip], 0
        om, dom = damage_law(*law, kap[ip])
        D = (1 - om) * D0
        K1 += wip * B.T @ D @ B
        deq = vonmises_deriv(eps, *vonmises) if vonmises else D0 @ eps if mazars is None else mazars_deriv(eps, *mazars)
        K3 -= wip * np.outer(Nip, deq @ B)
        if ld[ip]:
            K2 -= wip * dom * B.T @ D0 @ np.outer(eps, Nip)
        K4 += wip * (np.outer(Nip, Nip) + c * g @ g.T)
        f1 += wip * B.T @ (D @ eps)
        f2 -= wip * (vonmises_eq(eps, *vonmises) if vonmises else 0.5 * eps @ D0 @ eps if mazars is None else mazars_eq(eps)) * Nip
    f2 += K4 @ eqv
    return K1, K2, K3, K4, f1, f2, kap, ld


# ---- Mazars equivalent strain as DamageMaterial computes it (DamageMaterial.cpp:618-642, :886-1032): principal values
# from the trigonometric solution of the characteristic cubic (MathUtils.cpp:112-182) with the reference's truncated
# PI = 3.14159265 (Constants.cpp:13) -- written independently of the C oracle, same published formulas ----
REF_PI = 3.14159265


def strain_tensor_v6(strain):
    v = np.zeros(6)
    v[:3] = strain[:3]; v[3] = 0.5 * strain[3]
    if strain.size == 6:
        v[4] = 0.5 * strain[4]; v[5] = 0.5 * strain[5]
    return v


def principal_values_ref(v):
    I1 = v[0] + v[1] + v[2]
    I2 = v[0] * v[1] + v[1] * v[2] + v[2] * v[0] - (v[3] ** 2 + v[4] ** 2 + v[5] ** 2)
    I3 = v[0] * v[1] * v[2] + 2 * v[3] * v[4] * v[5] - v[0] * v[4] ** 2 - v[1] * v[5] ** 2 - v[2] * v[3] ** 2
    aa, bb, cc = -I1, I2, -I3
    q = (aa * aa - 3 * bb) / 9.0
    r = (2 * aa ** 3 - 9 * aa * bb + 27 * cc) / 54.0
    with np.errstate(all="ignore"):
        h = r / np.sqrt(q * q * q)
        if abs(h) > 1.0:
            h = -1.0 if h < 0 else 1.0
        phi = np.arccos(h); p = np.sqrt(q)
        rad = 0.6666666666666667 * REF_PI
        roots = np.array([-2 * p * np.cos(phi / 3) - aa / 3, -2 * p * np.cos(phi / 3 + rad) - aa / 3,
                          -2 * p * np.cos(phi / 3 - rad) - aa / 3])
    return roots, I1, I2


def mazars_eq(strain):
    pr, _, _ = principal_values_ref(strain_tensor_v6(strain))
    return np.sqrt(sum(x * x for x in pr if x > 0))


def mazars_deriv(strain, nu, plane_stress):
    n = strain.size
    ret = np.zeros(n)
    if n == 6:
        v = strain_tensor_v6(strain)
        pr, I1, I2 = principal_values_ref(v)
        dI1 = np.array([1.0, 1, 1, 0, 0, 0])
        dI2 = np.array([v[1] + v[2], v[2] + v[0], v[0] + v[1], -2 * v[3], -2 * v[4], -2 * v[5]])
        dI3 = np.array([v[1] * v[2] - v[4] ** 2, v[2] * v[0] - v[5] ** 2, v[0] * v[1] - v[3] ** 2,
                        2 * (v[4] * v[5] - v[2] * v[3]), 2 * (v[5] * v[3] - v[0] * v[4]), 2 * (v[3] * v[4] - v[1] * v[5])])
        bar = np.sqrt(sum(x * x for x in pr if x > 0))
        if bar < 1e-16:
            return ret
        for x in pr:
            if x > 0:
                ret += x * (x * x * dI1 - x * dI2 + dI3) / (3 * x * x - 2 * I1 * x + I2)
        return ret / bar
    exx, eyy, exy = strain[0], strain[1], 0.5 * strain[3]
    d = np.sqrt((exx - eyy) ** 2 + 4 * exy * exy)
    poiD = nu / (nu - 1.0)
    p0 = 0.5 * (exx + eyy + d); p1 = exx + eyy - p0; p2 = poiD * (exx + eyy) if plane_stress else 0.0
    p0, p1, p2 = max(p0, 0.0), max(p1, 0.0), max(p2, 0.0)
    den = np.sqrt(p0 * p0 + p1 * p1 + p2 * p2)
    if den < 1e-8:
        return ret
    if d == 0.0:
        d = 1.0
    a = 0.5 + 0.5 / d * (exx - eyy); b = 1.0 - a; c = 2.0 * 0.5 / d * exy
    ret[0] = (p0 * a + p1 * b + p2 * poiD) / den
    ret[1] = (p0 * b + p1 * a + p2 * poiD) / den
    ret[3] = (p0 * c - p1 * c) / den
    return ret


# ---- von Mises equivalent strain (DamageMaterial.cpp:609-617, :667-684; VonMisesStrain.cpp): written from the tensor
# definition -- J2 = 1/2 dev : dev -- not from the reference's component formulas; its gradient is taken with respect
# to the tensor components for the shear entries too (twice the derivative with respect to gamma), as the reference does ----
def _vm_parts(strain, eta, nu):
    e = np.array([[strain[0], strain[3] / 2, strain[5] / 2], [strain[3] / 2, strain[1], strain[4] / 2],
                  [strain[5] / 2, strain[4] / 2, strain[2]]])
    I1 = np.trace(e); dev = e - I1 / 3 * np.eye(3)
    J2 = 0.5 * np.tensordot(dev, dev)
    a = (eta - 1) / (1 - 2 * nu); b = 0.5 / eta; c = 12 * eta / (1 + nu) ** 2
    return I1, J2, dev, a, b, c


def vonmises_eq(strain, eta, nu):
    I1, J2, _, a, b, c = _vm_parts(strain, eta, nu)
    return b * (a * I1 + np.sqrt(a * a * I1 * I1 + c * J2))


def vonmises_deriv(strain, eta, nu):
    I1, J2, dev, a, b, c = _vm_parts(strain, eta, nu)
    d = np.sqrt(a * a * I1 * I1 + c * J2)
    d = 1.0 if d == 0.0 else d
    dI1 = a * b + b / d * a * a * I1; dJ2 = b / d * c * 0.5
    gJ2 = np.array([dev[0, 0], dev[1, 1], dev[2, 2], 2 * dev[0, 1], 2 * dev[1, 2], 2 * dev[0, 2]])   # dJ2/de_ij, shear doubled
    return dI1 * np.array([1.0, 1, 1, 0, 0, 0]) + dJ2 * gJ2
