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
