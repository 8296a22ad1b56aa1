"""Known answers that do NOT come from the oracle (every golden vector is the oracle's own): one unit
square Quad4 element, PhaseFracture / AT2 / AmorPhase, plane strain, uniaxial stretch eps_xx = e with a uniform phase
field.  Everything is written out by hand from the reference's formulas:
  AmorPhaseMaterial.cpp:200-232   Psi+ = K/2 <tr>+^2 + G dev:dev,  sigma = g (K tr I + 2 G dev) for tr > 0
  PhaseFractureModel.cpp:660-676  g = (1 - phi)^2 + 1e-7, g' = -2 (1 - phi), g'' = 2          (AT2: p = 2, a1 = 2, a2 = -1/2, a3 = 0)
  PhaseFractureModel.cpp:296-355  AT2: cw = 2, eta = 0 (w = phi^2), Psi0 = 0
  PhaseFractureModel.cpp:713-730  K4 = w N (gc/(cw l0) w'' + g'' Psi) N^t + w (2 gc l0 / cw) Be^t Be
                                  f2 = w N (gc/(cw l0) w' + g' Psi) + w (2 gc l0 / cw) Be^t grad phi,   f1 = w Bd^t sigma
with the exact integrals of the bilinear shape functions on the unit square (2 x 2 Gauss is exact for them).
The CPU test holds the oracle to these numbers, the GPU test the HIP path (through the C ABI)."""
import numpy as np
import pytest

from falcon_amd import capi
from oracle import pyoracle as po

E, NU, GC, L0, EPS, PHI = 210.0, 0.3, 2.7e-3, 0.1, 2.0e-3, 0.3
COORDS = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0]])
CONN = np.array([[0, 1, 2, 3]], dtype=np.int32)


def closed_form():
    K = E / (3 * (1 - 2 * NU)); G = E / (2 * (1 + NU))
    g = (1 - PHI) ** 2 + 1e-7; dg = -2 * (1 - PHI); ddg = 2.0
    psi = 0.5 * K * EPS ** 2 + G * (2.0 / 3.0) * EPS ** 2            # tr = e > 0: all of it is Psi+
    sxx = g * (K * EPS + 2 * G * (2.0 / 3.0) * EPS); syy = g * (K * EPS - 2 * G * (1.0 / 3.0) * EPS)
    cw = 2.0
    f = np.zeros((4, 3))
    f[:, 0] = 0.5 * sxx * np.array([-1, 1, 1, -1])                    # int Bd^t sigma over the unit square
    f[:, 1] = 0.5 * syy * np.array([-1, -1, 1, 1])
    f[:, 2] = 0.25 * (GC / (cw * L0) * 2 * PHI + dg * psi)            # int N_a = 1/4, grad phi = 0
    M = np.array([[4, 2, 1, 2], [2, 4, 2, 1], [1, 2, 4, 2], [2, 1, 2, 4]]) / 36.0       # int N_a N_b
    L = np.array([[4, -1, -2, -1], [-1, 4, -1, -2], [-2, -1, 4, -1], [-1, -2, -1, 4]]) / 6.0   # int grad N_a . grad N_b
    K4 = (GC / (cw * L0) * 2.0 + ddg * psi) * M + (2 * GC * L0 / cw) * L
    return f, K4, psi


def state():
    st = np.zeros(12)
    st[0::3] = EPS * COORDS[:, 0]
    st[2::3] = PHI
    return st


def check(Kvals, rp, ci, f, hist_new):
    fe, K4, psi = closed_form()
    assert np.abs(f.reshape(4, 3) - fe).max() < 1e-13 * np.abs(fe).max()
    dense = np.zeros((12, 12))
    for r in range(12):
        dense[r, ci[rp[r]:rp[r + 1]]] = Kvals[rp[r]:rp[r + 1]]
    assert np.abs(dense[2::3, 2::3] - K4).max() < 1e-13 * np.abs(K4).max()
    assert np.abs(dense[0::3, 2::3]).max() == 0.0 and np.abs(dense[2::3, 0::3]).max() == 0.0   # keepOffDiags = false
    assert np.abs(hist_new - psi).max() < 1e-13 * psi                 # newHist_ = max(Psi, Psi0 = 0)
    # K1 is symmetric and annihilates the rigid translations (not closed-form here, but oracle-free)
    K1 = dense[np.ix_([0, 1, 3, 4, 6, 7, 9, 10], [0, 1, 3, 4, 6, 7, 9, 10])]
    assert np.abs(K1 - K1.T).max() < 1e-13 * np.abs(K1).max()
    assert np.abs(K1 @ np.tile([1.0, 0.0], 4)).max() < 1e-13 * np.abs(K1).max()


def test_oracle_against_the_hand_computed_element():
    cfg = po.make_config(po.QUAD4_G2, po.PHASE_FRACTURE, po.AMOR, young=E, poisson=NU, gc=GC, l0=L0)
    rp, ci = po.pattern(4, CONN, 3, 2)
    r = po.assemble(cfg, COORDS, CONN, rp, ci, state())
    check(r["K"], rp, ci, r["f"], r["hist_new"])


@pytest.mark.gpu
def test_hip_path_against_the_hand_computed_element():
    asm = capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE, capi.AMOR_PHASE, COORDS, CONN, young=E, poisson=NU, gc=GC, l0=L0)
    rp, ci = asm.pattern()
    K, f, _ = asm.assemble_host(state(), np.zeros(asm.nnz), flags=capi.OVERWRITE)
    check(K, rp, ci, f, asm.get_history(1))


# ---- PhaseFractureExt: extrapolated phase field in the degradation of the stress, penalty on phi < phi_old ----
PHI0, PHI00, DT, DT0, BETA = 0.35, 0.25, 0.7, 0.4, 6500.0


def closed_form_ext():
    """PhaseFractureExtModel.cpp:656-691 (pfEx = -phi00 dt/dt0 + phi0 (dt + dt0)/dt0, clipped to [0, 1], degrades the
    stress), :714-730 (K4 / f2 with g'(phi), g''(phi) of the CURRENT phase field and Psi = max(Psi+, Psi0)), :752-758
    (phi0 - phi > 1e-15: K4 += gc/(cw l0) beta N N^t, f2 -= gc/(cw l0) beta (phi0 - phi) N)."""
    K = E / (3 * (1 - 2 * NU)); G = E / (2 * (1 + NU))
    pfex = min(max(-PHI00 * DT / DT0 + PHI0 * (DT + DT0) / DT0, 0.0), 1.0)
    gex = (1 - pfex) ** 2 + 1e-7
    dg = -2 * (1 - PHI); ddg = 2.0
    psi = 0.5 * K * EPS ** 2 + G * (2.0 / 3.0) * EPS ** 2
    sxx = gex * (K * EPS + 2 * G * (2.0 / 3.0) * EPS); syy = gex * (K * EPS - 2 * G * (1.0 / 3.0) * EPS)
    cw = 2.0; cp = GC / (cw * L0) * BETA
    f = np.zeros((4, 3))
    f[:, 0] = 0.5 * sxx * np.array([-1, 1, 1, -1]); f[:, 1] = 0.5 * syy * np.array([-1, -1, 1, 1])
    f[:, 2] = 0.25 * (GC / (cw * L0) * 2 * PHI + dg * psi) - 0.25 * cp * (PHI0 - PHI)
    M = np.array([[4, 2, 1, 2], [2, 4, 2, 1], [1, 2, 4, 2], [2, 1, 2, 4]]) / 36.0
    L = np.array([[4, -1, -2, -1], [-1, 4, -1, -2], [-2, -1, 4, -1], [-1, -2, -1, 4]]) / 6.0
    K4 = (GC / (cw * L0) * 2.0 + ddg * psi + cp) * M + (2 * GC * L0 / cw) * L
    return f, K4


def ext_states():
    st0 = np.zeros(12); st00 = np.zeros(12)
    st0[2::3] = PHI0; st00[2::3] = PHI00
    return st0, st00


def check_ext(Kvals, rp, ci, f):
    fe, K4 = closed_form_ext()
    assert np.abs(f.reshape(4, 3) - fe).max() < 1e-13 * np.abs(fe).max()
    dense = np.zeros((12, 12))
    for r in range(12):
        dense[r, ci[rp[r]:rp[r + 1]]] = Kvals[rp[r]:rp[r + 1]]
    assert np.abs(dense[2::3, 2::3] - K4).max() < 1e-13 * np.abs(K4).max()


def test_oracle_against_the_hand_computed_ext_element():
    cfg = po.make_config(po.QUAD4_G2, po.PHASE_FRACTURE_EXT, po.AMOR, young=E, poisson=NU, gc=GC, l0=L0, beta=BETA)
    rp, ci = po.pattern(4, CONN, 3, 2)
    st0, st00 = ext_states()
    r = po.assemble(cfg, COORDS, CONN, rp, ci, state(), state0=st0, state00=st00, dt=DT, dt0=DT0)
    check_ext(r["K"], rp, ci, r["f"])


@pytest.mark.gpu
def test_hip_path_against_the_hand_computed_ext_element():
    asm = capi.Assembler(capi.QUAD4_G2, capi.PHASE_FRACTURE_EXT, capi.AMOR_PHASE, COORDS, CONN, young=E, poisson=NU, gc=GC, l0=L0,
                         beta=BETA)
    rp, ci = asm.pattern()
    st0, st00 = ext_states()
    K, f, _ = asm.assemble_host(state(), np.zeros(asm.nnz), state0=st0, state00=st00, dt=DT, dt0=DT0, flags=capi.OVERWRITE)
    check_ext(K, rp, ci, f)
