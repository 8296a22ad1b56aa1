// fa_kernel_impl.h -- gfx950 kernels for the element-level K / f_int assembly.
//
// Work decomposition ("owner computes", atomic-free):
//   * one lane per (node a, incident element e) pair, i.e. per entry of the inverse connectivity
//     n2e; lanes of one node are adjacent, a workgroup owns a contiguous node range [n0,n1);
//   * the lane evaluates the Gauss-point loop of e (shape gradients, B, material update,
//     degradation, history) and keeps only the rows of node a of the element blocks
//     K1..K4 / f1,f2 (PhaseFractureModel.cpp:713-730) in registers;
//   * the workgroup's slice of the CSR value array -- rows of nodes [n0,n1) are one contiguous
//     range because dofs are node-major and rows column-sorted -- is mirrored in LDS; lanes add
//     their row blocks into it in phases ordered by the lane's slot in the node's element list
//     (two lanes of one phase never share a node => no atomics, and every CSR entry is summed in
//     ascending element order exactly like the reference's serial element loop);
//   * the LDS image is then streamed to HBM with fully coalesced stores, each K entry written once.
// A generic fallback (any dof map / any CSR pattern) scatters the same row blocks with a per-entry
// column search and FP64 atomics, like Jive's addBlock.
#pragma once
#include <hip/hip_runtime.h>

#include "fa_internal.h"

namespace fa {

template <int RANK_, int NPE_, int NIP_>
struct Shape {
  static constexpr int RANK = RANK_, NPE = NPE_, NIP = NIP_;
  static constexpr int S   = (RANK_ == 2) ? 3 : 6;   // strain components that matter (2D: xx, yy, xy)
  static constexpr int STR = (RANK_ == 2) ? 4 : 6;   // STRAIN_COUNTS (util/Constants.cpp:9)
  static constexpr int ND  = RANK_ * NPE_;
  static constexpr int BLOCK = block_for(RANK_), LDS = lds_for(RANK_);
  static constexpr int MIN_WAVES = (RANK_ == 2) ? FA_WAVES2D : 1;   // waves per SIMD the register allocator must allow
};

template <class SH, bool WANT_K, bool PF, bool KEEP>
struct RowBlock {
  double fu[SH::RANK];
  double fp;
  double Kuu[WANT_K ? SH::RANK : 1][WANT_K ? SH::ND : 1];
  double Kpp[(WANT_K && PF) ? SH::NPE : 1];
  double Kup[(WANT_K && KEEP) ? SH::RANK : 1][(WANT_K && KEEP) ? SH::NPE : 1];
  double Kpu[(WANT_K && KEEP) ? SH::ND : 1];
};

// rational degradation function and its derivatives (PhaseFractureModel.cpp:660-676).
// ptype: 0 -> p = 2, 1 -> p = 2.5, 2 -> p = 4; pow(x,0) == 1 also at x == 0.
__device__ __forceinline__ void degrade(const Params& P, double pf, double& g, double& dg, double& ddg) {
  const double x = 1.0 - pf;
  double gn, dgn, ddgn;
  if (P.ptype == 0)      { gn = x * x;               dgn = -2.0 * x;            ddgn = 2.0; }
  else if (P.ptype == 1) { const double sq = sqrt(x);
                           gn = x * x * sq;          dgn = -2.5 * (x * sq);     ddgn = 2.5 * 1.5 * sq; }
  else                   { const double x2 = x * x;
                           gn = x2 * x2;             dgn = -4.0 * (x2 * x);     ddgn = 4.0 * 3.0 * x2; }
  const double a12 = P.a1 * P.a2, a123 = a12 * P.a3;
  const double gd   = gn + P.a1 * pf + a12 * pf * pf + a123 * pf * pf * pf;
  const double dgd  = dgn + P.a1 + 2.0 * a12 * pf + 3.0 * a123 * pf * pf;
  const double ddgd = ddgn + 2.0 * a12 + 6.0 * a123 * pf;
  const double inv  = 1.0 / gd;
  g   = gn * inv + 1.e-7;
  const double q = dgn * gd - gn * dgd;
  dg  = q * (inv * inv);
  ddg = ((ddgn * gd - gn * ddgd) * gd - 2.0 * q * dgd) * (inv * inv * inv);
}

// Isotropic-form materials: D = [d0 d1 d1; d1 d0 d1; d1 d1 d0] (+) d3 on the shear diagonal.
//   material 0: Hooke (HookeMaterial.cpp:152-166), 1: AmorPhase (AmorPhaseMaterial.cpp:200-232),
//   2: BourdinPhase (BourdinPhaseMaterial.cpp:131-145).
// e: strain with S components (2D: xx, yy, gamma_xy -- eps_zz is identically 0 because B row 2 is 0).
template <int RANK, int S>
__device__ __forceinline__ void iso_update(const Params& P, const double (&e)[S], double g,
                                           double& d0, double& d1, double& d3,
                                           double (&sig)[S], double (&sp)[S], double& Psi) {
  constexpr int NN = RANK;                                  // normal components carried
  const double tr = (RANK == 3) ? (e[0] + e[1] + e[2]) : (e[0] + e[1] + 0.0);
  if (P.material == 1) {
    double dd = 2.0 * P.G * (2.0 / 3.0), od = 2.0 * P.G * (-1.0 / 3.0);
    const double sh = 2.0 * P.G * 0.5;
    if (tr > 0.0) { dd += P.Kb; od += P.Kb; }
    double psi2 = 0.0;
#pragma unroll
    for (int i = 0; i < NN; i++) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NN; j++) s += ((i == j) ? dd : od) * e[j];
      sp[i] = s; psi2 += s * e[i];
    }
#pragma unroll
    for (int i = NN; i < S; i++) { sp[i] = sh * e[i]; psi2 += sp[i] * e[i]; }
    Psi = 0.5 * psi2;
    d0 = dd * g; d1 = od * g; d3 = sh * g;
    if (tr <= 0.0) { d0 += P.Kb; d1 += P.Kb; }
#pragma unroll
    for (int i = 0; i < NN; i++) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NN; j++) s += ((i == j) ? d0 : d1) * e[j];
      sig[i] = s;
    }
#pragma unroll
    for (int i = NN; i < S; i++) sig[i] = d3 * e[i];
  } else {
    double psi2 = 0.0;
#pragma unroll
    for (int i = 0; i < NN; i++) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < NN; j++) s += ((i == j) ? P.hd0 : P.hd1) * e[j];
      sp[i] = s; psi2 += s * e[i];
    }
#pragma unroll
    for (int i = NN; i < S; i++) { sp[i] = P.hd3 * e[i]; psi2 += sp[i] * e[i]; }
    Psi = 0.5 * psi2;
    d0 = g * P.hd0; d1 = g * P.hd1; d3 = g * P.hd3;
#pragma unroll
    for (int i = 0; i < S; i++) sig[i] = g * sp[i];
  }
}

// B_a^T v for a Voigt vector v (2D: xx,yy,xy; 3D: xx,yy,zz,xy,yz,zx), ShapeUtils.cpp:73-145
template <int RANK, int S>
__device__ __forceinline__ void Bt_vec(const double (&ga)[RANK], const double (&v)[S], double (&out)[RANK]) {
  if constexpr (RANK == 2) {
    out[0] = ga[0] * v[0] + ga[1] * v[2];
    out[1] = ga[1] * v[1] + ga[0] * v[2];
  } else {
    out[0] = ga[0] * v[0] + ga[1] * v[3] + ga[2] * v[5];
    out[1] = ga[1] * v[1] + ga[0] * v[3] + ga[2] * v[4];
    out[2] = ga[2] * v[2] + ga[1] * v[4] + ga[0] * v[5];
  }
}

// Gauss-point loop of element e, keeping the rows of local node la.
template <class SH, int FAM, bool WANT_K, bool KEEP>
__device__ __forceinline__ void compute_rowblock(const Params& P, int64_t e, int la, const int32_t (&cn)[SH::NPE],
                                                 RowBlock<SH, WANT_K, FAM != FAM_LE, KEEP>& R, double& err_acc) {
  // preHist_.H0_ of all Gauss points of e: one contiguous read, issued with the other gathers
  double H0v[SH::NIP];
  if constexpr (FAM != FAM_LE) {
    if (P.model == 1) {
#pragma unroll
      for (int ip = 0; ip < SH::NIP; ip++) H0v[ip] = P.hist_pre[e * SH::NIP + ip];
    }
  }
  constexpr int RANK = SH::RANK, NPE = SH::NPE, NIP = SH::NIP, S = SH::S, STR = SH::STR;
  constexpr bool PF = (FAM != FAM_LE);
  const int ndofpn = PF ? RANK + 1 : RANK;
  const double* __restrict__ tabN = P.tab;
  const double* __restrict__ tabD = P.tab + NIP * NPE;
  const double* __restrict__ tabW = P.tab + NIP * NPE * (1 + RANK);

  double x[NPE][RANK], u[NPE][RANK];
  double phi[PF ? NPE : 1], phi0[PF ? NPE : 1], phiX[PF ? NPE : 1];
#pragma unroll
  for (int a = 0; a < NPE; a++) {
    const int64_t n = cn[a];
    // XDofSpace::getDofIndices: node-major, type-minor unless the caller supplied its own dof map
    int64_t dofs[PF ? RANK + 1 : RANK];
#pragma unroll
    for (int i = 0; i < (PF ? RANK + 1 : RANK); i++)
      dofs[i] = P.node_dofs ? P.node_dofs[n * ndofpn + i] : n * ndofpn + i;
#pragma unroll
    for (int i = 0; i < RANK; i++) { x[a][i] = P.coords[n * RANK + i]; u[a][i] = P.state[dofs[i]]; }
    if constexpr (PF) {
      phi[a] = P.state[dofs[RANK]];
      if (P.model >= 2) { phi0[a] = P.state0[dofs[RANK]]; phiX[a] = P.stateX[dofs[RANK]]; }
      else { phi0[a] = 0.0; phiX[a] = 0.0; }
    }
  }

#pragma unroll
  for (int i = 0; i < RANK; i++) R.fu[i] = 0.0;
  R.fp = 0.0;
  if constexpr (WANT_K) {
#pragma unroll
    for (int i = 0; i < RANK; i++)
#pragma unroll
      for (int j = 0; j < SH::ND; j++) R.Kuu[i][j] = 0.0;
    if constexpr (PF) {
#pragma unroll
      for (int b = 0; b < NPE; b++) R.Kpp[b] = 0.0;
    }
    if constexpr (KEEP) {
#pragma unroll
      for (int i = 0; i < RANK; i++)
#pragma unroll
        for (int b = 0; b < NPE; b++) R.Kup[i][b] = 0.0;
#pragma unroll
      for (int j = 0; j < SH::ND; j++) R.Kpu[j] = 0.0;
    }
  }

#pragma unroll 1
  for (int ip = 0; ip < NIP; ip++) {
    // ---- IShape::getShapeGradients: J = sum_a x_a (x) dN_a/dxi, grads = dN J^{-1}, w = w_ref det J ----
    double J[RANK][RANK];
#pragma unroll
    for (int i = 0; i < RANK; i++)
#pragma unroll
      for (int k = 0; k < RANK; k++) J[i][k] = 0.0;
#pragma unroll
    for (int a = 0; a < NPE; a++)
#pragma unroll
      for (int i = 0; i < RANK; i++)
#pragma unroll
        for (int k = 0; k < RANK; k++) J[i][k] += x[a][i] * tabD[(ip * NPE + a) * RANK + k];
    double inv[RANK][RANK], det;
    if constexpr (RANK == 2) {
      det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      const double r = 1.0 / det;
      inv[0][0] = J[1][1] * r;  inv[0][1] = -J[0][1] * r;
      inv[1][0] = -J[1][0] * r; inv[1][1] = J[0][0] * r;
    } else {
      const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
      const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
      const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
      det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
      const double r = 1.0 / det;
      inv[0][0] = c00 * r;
      inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * r;
      inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * r;
      inv[1][0] = c01 * r;
      inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * r;
      inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * r;
      inv[2][0] = c02 * r;
      inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * r;
      inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * r;
    }
    const double w = tabW[ip] * det;
    double g[NPE][RANK];
#pragma unroll
    for (int a = 0; a < NPE; a++)
#pragma unroll
      for (int i = 0; i < RANK; i++) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < RANK; k++) s += tabD[(ip * NPE + a) * RANK + k] * inv[k][i];
        g[a][i] = s;
      }
    // ---- strain = B u (ShapeUtils.cpp:73-145); 2D keeps xx, yy, gamma_xy ----
    double eps[S];
#pragma unroll
    for (int c = 0; c < S; c++) eps[c] = 0.0;
#pragma unroll
    for (int a = 0; a < NPE; a++) {
      if constexpr (RANK == 2) {
        eps[0] += g[a][0] * u[a][0];
        eps[1] += g[a][1] * u[a][1];
        eps[2] += g[a][1] * u[a][0]; eps[2] += g[a][0] * u[a][1];
      } else {
        eps[0] += g[a][0] * u[a][0];
        eps[1] += g[a][1] * u[a][1];
        eps[2] += g[a][2] * u[a][2];
        eps[3] += g[a][1] * u[a][0]; eps[3] += g[a][0] * u[a][1];
        eps[4] += g[a][2] * u[a][1]; eps[4] += g[a][1] * u[a][2];
        eps[5] += g[a][0] * u[a][2]; eps[5] += g[a][2] * u[a][0];
      }
    }
    const bool owner = ((ip % NPE) == la);             // exactly one lane of the element owns each ip
    const int64_t ipoint = e * NIP + ip;               // initializeIPMPMap_, PhaseFractureModel.cpp:1070-1093
    const int srows = PF ? STR + 1 : STR;
    if (P.strain != nullptr && owner) {
      double* sc = P.strain + ipoint * srows;
      if constexpr (RANK == 2) { sc[0] = eps[0]; sc[1] = eps[1]; sc[2] = 0.0; sc[3] = eps[2]; }
      else {
#pragma unroll
        for (int c = 0; c < S; c++) sc[c] = eps[c];
      }
    }

    double d0, d1, d3, sig[S], sp[S], Psi = 0.0;
    double dgphi = 0.0, ddgphi = 0.0, pf = 0.0, pf0 = 0.0, loading = 1.0;
    double gradphi[RANK];
    if constexpr (!PF) {
      iso_update<RANK, S>(P, eps, 1.0, d0, d1, d3, sig, sp, Psi);
    } else {
#pragma unroll
      for (int i = 0; i < RANK; i++) gradphi[i] = 0.0;
#pragma unroll
      for (int a = 0; a < NPE; a++) {
        pf += tabN[ip * NPE + a] * phi[a];
#pragma unroll
        for (int i = 0; i < RANK; i++) gradphi[i] += g[a][i] * phi[a];
      }
      double gmat;
      if (P.model == 1) {
        degrade(P, pf, gmat, dgphi, ddgphi);
      } else {
        // extrapolated phase field (PhaseFractureExtModel.cpp:656-691, PhaseFractureExtItModel.cpp:693)
        double pfX = 0.0;
#pragma unroll
        for (int a = 0; a < NPE; a++) { pf0 += tabN[ip * NPE + a] * phi0[a]; pfX += tabN[ip * NPE + a] * phiX[a]; }
        double pfEx;
        if (P.model == 2) {
          if (P.dt0 < 1.e-15) pfEx = pf0;
          else                pfEx = -pfX * P.dt / P.dt0 + pf0 * (P.dt + P.dt0) / P.dt0;
        } else {
          pfEx = pfX;
          if (owner) { const double d = pf - pfEx; err_acc += d * d; }
        }
        if (pfEx < 0.0) pfEx = 0.0; else if (pfEx > 1.0) pfEx = 1.0;
        double t1, t2, t3;
        degrade(P, pfEx, gmat, t1, t2);
        degrade(P, pf, t3, dgphi, ddgphi);
      }
      if (P.strain != nullptr && owner) P.strain[ipoint * srows + STR] = gmat;
      iso_update<RANK, S>(P, eps, gmat, d0, d1, d3, sig, sp, Psi);
      if (P.model == 1) {
        const double H0 = H0v[ip];                      // PhaseFractureModel.cpp:696-703
        if (Psi < H0) loading = 0.0;
        Psi = (Psi > H0) ? Psi : H0;
        if (owner) P.hist_new[ipoint] = Psi;
      } else {
        Psi = (Psi > P.psi0) ? Psi : P.psi0;            // PhaseFractureExtModel.cpp:726
      }
    }

    // ---- rows of node la ----
    double wga[RANK];                                 // w * grad N_la (select, no runtime register indexing)
    double Nla = 0.0;
#pragma unroll
    for (int a = 0; a < NPE; a++)
      if (a == la) {
#pragma unroll
        for (int i = 0; i < RANK; i++) wga[i] = w * g[a][i];
        Nla = tabN[ip * NPE + a];
      }
    {
      double t[RANK];
      Bt_vec<RANK, S>(wga, sig, t);                     // f1 += w B^T sigma (:727)
#pragma unroll
      for (int i = 0; i < RANK; i++) R.fu[i] += t[i];
    }
    if constexpr (WANT_K) {
#pragma unroll
      for (int b = 0; b < NPE; b++) {                   // K1 += w B_a^T D B_b (:713)
        if constexpr (RANK == 2) {
          const double gbx = g[b][0], gby = g[b][1];
          R.Kuu[0][2 * b]     += (wga[0] * d0) * gbx + (wga[1] * d3) * gby;
          R.Kuu[0][2 * b + 1] += (wga[0] * d1) * gby + (wga[1] * d3) * gbx;
          R.Kuu[1][2 * b]     += (wga[1] * d1) * gbx + (wga[0] * d3) * gby;
          R.Kuu[1][2 * b + 1] += (wga[1] * d0) * gby + (wga[0] * d3) * gbx;
        } else {
          const double dotp = wga[0] * g[b][0] + wga[1] * g[b][1] + wga[2] * g[b][2];
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
              if (i == j) R.Kuu[i][3 * b + j] += (d0 - d3) * (wga[i] * g[b][i]) + d3 * dotp;
              else        R.Kuu[i][3 * b + j] += d1 * (wga[i] * g[b][j]) + d3 * (wga[j] * g[b][i]);
            }
        }
      }
    }
    if constexpr (PF) {
      const double wNa = w * Nla;
      double gg = 0.0;
#pragma unroll
      for (int i = 0; i < RANK; i++) gg += wga[i] * gradphi[i];
      const double dw = P.eta + 2.0 * (1.0 - P.eta) * pf;
      R.fp += wNa * (P.gcl * dw + dgphi * Psi) + P.c2 * gg;            // f2 (:728-730)
      const bool pen = (P.model >= 2) && (pf0 - pf > 1.e-15);           // PhaseFractureExtModel.cpp:752-758
      if (pen) R.fp -= wNa * (P.gcl * P.beta * (pf0 - pf));
      if constexpr (WANT_K) {
        const double ddw = 2.0 * (1.0 - P.eta);
        double c1 = P.gcl * ddw + ddgphi * Psi;
        if (pen) c1 += P.gcl * P.beta;
#pragma unroll
        for (int b = 0; b < NPE; b++) {                                 // K4 (:714-716)
          double gab = 0.0;
#pragma unroll
          for (int i = 0; i < RANK; i++) gab += wga[i] * g[b][i];
          R.Kpp[b] += (wNa * c1) * tabN[ip * NPE + b] + P.c2 * gab;
        }
        if constexpr (KEEP) {
          if (P.keep2) {                                                // K2 += w g' B^T (sigma+ (x) N) (:720)
            double t[RANK];
            Bt_vec<RANK, S>(wga, sp, t);
#pragma unroll
            for (int i = 0; i < RANK; i++)
#pragma unroll
              for (int b = 0; b < NPE; b++) R.Kup[i][b] += (dgphi * t[i]) * tabN[ip * NPE + b];
          }
          if (P.keep3) {                                                // K3 += w loading g' N (x) (sigma+^T B) (:721)
            const double c = wNa * loading * dgphi;
#pragma unroll
            for (int b = 0; b < NPE; b++) {
              double t[RANK], gb[RANK];
#pragma unroll
              for (int i = 0; i < RANK; i++) gb[i] = g[b][i];
              Bt_vec<RANK, S>(gb, sp, t);
#pragma unroll
              for (int j = 0; j < RANK; j++) R.Kpu[b * RANK + j] += c * t[j];
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Row-gather kernel (canonical node-major dofs, column-sorted blocked pattern)
// ------------------------------------------------------------------------------------------------
template <class SH, int FAM, bool WANT_K, bool KEEP>
__global__ __launch_bounds__(SH::BLOCK, SH::MIN_WAVES) void assemble_rows(const Params P) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, kBlock = SH::BLOCK, kLdsDoubles = SH::LDS;
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? RANK + 1 : RANK;
  __shared__ double sK[WANT_K ? kLdsDoubles : 1];
  __shared__ double sF[kBlock * NDOF];
  __shared__ double sErr[kBlock / 64];

  const int tid = threadIdx.x;
  const int n0 = P.wg_start[blockIdx.x], n1 = P.wg_start[blockIdx.x + 1];
  const int64_t base = P.row_ptr[(int64_t)n0 * NDOF];
  const int len = WANT_K ? (int)(P.row_ptr[(int64_t)n1 * NDOF] - base) : 0;
  const int nf = (n1 - n0) * NDOF;
  if constexpr (WANT_K)
    for (int i = tid; i < len; i += kBlock) sK[i] = 0.0;
  for (int i = tid; i < nf; i += kBlock) sF[i] = 0.0;

  const int t0 = P.n2e_ptr[n0], t1 = P.n2e_ptr[n1];
  const int t = t0 + tid;
  const bool active = t < t1;
  RowBlock<SH, WANT_K, PF, KEEP> R;
  int32_t cn[NPE];
  int la = 0, slot = -1, a = n0;
  int64_t e = 0;
  int rs[NDOF];
  uint8_t ps[NPE];
  double err_acc = 0.0;
  if (active) {
    const int ent = P.n2e_ent[t];
    e = ent / NPE; la = ent - (int)e * NPE;
#pragma unroll
    for (int b = 0; b < NPE; b++) cn[b] = P.conn[e * NPE + b];
#pragma unroll
    for (int b = 0; b < NPE; b++) if (b == la) a = cn[b];
    slot = t - P.n2e_ptr[a];
    if constexpr (WANT_K) {                        // row starts / column positions: issued before the compute
#pragma unroll
      for (int i = 0; i < NDOF; i++) rs[i] = (int)(P.row_ptr[(int64_t)a * NDOF + i] - base);
#pragma unroll
      for (int b = 0; b < NPE; b++) ps[b] = P.pos[(e * NPE + la) * NPE + b];
    }
#if defined(FA_ABLATE) && (FA_ABLATE & 2)
    {   // experiment: gathers only, no Gauss-point arithmetic
      double acc = 0.0;
      for (int b = 0; b < NPE; b++) { acc += P.coords[(int64_t)cn[b] * RANK] + P.coords[(int64_t)cn[b] * RANK + 1];
        for (int i = 0; i < NDOF; i++) acc += P.state[(int64_t)cn[b] * NDOF + i]; }
      for (int ip = 0; ip < SH::NIP; ip++) acc += P.hist_pre[e * SH::NIP + ip];
      for (int i = 0; i < RANK; i++) R.fu[i] = acc; R.fp = acc;
      if constexpr (WANT_K) { for (int i = 0; i < RANK; i++) for (int j = 0; j < SH::ND; j++) R.Kuu[i][j] = acc + j;
        if constexpr (PF) for (int b = 0; b < NPE; b++) R.Kpp[b] = acc - b; }
    }
#else
    compute_rowblock<SH, FAM, WANT_K, KEEP>(P, e, la, cn, R, err_acc);
#endif
  }
  __syncthreads();

  // ---- phased, conflict-free accumulation into the LDS image of the CSR rows ----
#ifdef FA_LDS_ATOMIC
#define FA_LDS_ADD(ref, v) __hip_atomic_fetch_add(&(ref), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#else
#define FA_LDS_ADD(ref, v) (ref) += (v)
#endif
#if defined(FA_ABLATE) && (FA_ABLATE & 4)
  if (active) { asm volatile("" :: "v"(R.fu[0]), "v"(R.fp)); if constexpr (WANT_K) { for (int j = 0; j < SH::ND; j++) asm volatile("" :: "v"(R.Kuu[0][j]), "v"(R.Kuu[1][j])); if constexpr (PF) for (int b = 0; b < NPE; b++) asm volatile("" :: "v"(R.Kpp[b])); } }
  for (int s = 0; s < 0; s++) {
#else
#ifdef FA_LDS_ATOMIC
  for (int s = 0; s < 1; s++) {
#else
  for (int s = 0; s < P.max_valence; s++) {
#endif
#endif
#ifdef FA_LDS_ATOMIC
    if (active) {
#else
    if (active && slot == s) {
#endif
      const int fo = (a - n0) * NDOF;
#pragma unroll
      for (int i = 0; i < RANK; i++) FA_LDS_ADD(sF[fo + i], R.fu[i]);
      if constexpr (PF) FA_LDS_ADD(sF[fo + RANK], R.fp);
      if constexpr (WANT_K) {
#pragma unroll
        for (int i = 0; i < RANK; i++) {
#pragma unroll
          for (int b = 0; b < NPE; b++) {
            const int gs = rs[i] + (int)ps[b] * P.cnt[i];
#pragma unroll
            for (int j = 0; j < RANK; j++) FA_LDS_ADD(sK[gs + P.off[i][j]], R.Kuu[i][b * RANK + j]);
            if constexpr (KEEP) { if (P.keep2) FA_LDS_ADD(sK[gs + P.off[i][RANK]], R.Kup[i][b]); }
          }
        }
        if constexpr (PF) {
#pragma unroll
          for (int b = 0; b < NPE; b++) {
            const int gs = rs[RANK] + (int)ps[b] * P.cnt[RANK];
            FA_LDS_ADD(sK[gs + P.off[RANK][RANK]], R.Kpp[b]);
            if constexpr (KEEP) {
              if (P.keep3) {
#pragma unroll
                for (int j = 0; j < RANK; j++) FA_LDS_ADD(sK[gs + P.off[RANK][j]], R.Kpu[b * RANK + j]);
              }
            }
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- stream the finished rows out, coalesced ----
#if defined(FA_ABLATE) && (FA_ABLATE & 1)
  if (len < 0)
#endif
  if constexpr (WANT_K) {
    double* __restrict__ Kg = P.K + base;
    if (P.overwrite) for (int i = tid; i < len; i += kBlock) Kg[i] = sK[i];
    else             for (int i = tid; i < len; i += kBlock) Kg[i] += sK[i];
  }
  {
    double* __restrict__ fg = P.f + (int64_t)n0 * NDOF;
    if (P.overwrite) for (int i = tid; i < nf; i += kBlock) fg[i] = sF[i];
    else             for (int i = tid; i < nf; i += kBlock) fg[i] += sF[i];
  }

  if (PF && P.err_part != nullptr) {               // ExtIt: errExt_ partial of this workgroup
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) err_acc += __shfl_down(err_acc, o, 64);
    if ((tid & 63) == 0) sErr[tid >> 6] = err_acc;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int k = 0; k < kBlock / 64; k++) s += sErr[k];
      P.err_part[blockIdx.x] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Generic fallback: any dof map, any CSR pattern (searched per entry like MatrixBuilder::addBlock)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t find_col(const Params& P, int64_t row, int64_t col) {
  int64_t lo = P.row_ptr[row], hi = P.row_ptr[row + 1];
  if (P.col_sorted) {
    const int64_t end = hi;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (P.col_idx[mid] < col) lo = mid + 1; else hi = mid; }
    return (lo < end && P.col_idx[lo] == col) ? lo : -1;
  }
  for (int64_t k = lo; k < hi; k++) if (P.col_idx[k] == col) return k;
  return -1;
}

template <class SH, int FAM, bool WANT_K, bool KEEP>
__global__ __launch_bounds__(SH::BLOCK) void assemble_generic(const Params P) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, kBlock = SH::BLOCK;
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? RANK + 1 : RANK;
  __shared__ double sErr[kBlock / 64];
  const int tid = threadIdx.x;
  const int64_t t = (int64_t)blockIdx.x * kBlock + tid;
  const bool active = t < P.nel * NPE;
  double err_acc = 0.0;
  if (active) {
    const int64_t e = t / NPE; const int la = (int)(t - e * NPE);
    int32_t cn[NPE];
#pragma unroll
    for (int b = 0; b < NPE; b++) cn[b] = P.conn[e * NPE + b];
    RowBlock<SH, WANT_K, PF, KEEP> R;
    compute_rowblock<SH, FAM, WANT_K, KEEP>(P, e, la, cn, R, err_acc);
    auto dof = [&](int node, int i) -> int64_t {
      return P.node_dofs ? P.node_dofs[(int64_t)node * NDOF + i] : (int64_t)node * NDOF + i;
    };
    int a = cn[0];
#pragma unroll
    for (int b = 0; b < NPE; b++) if (b == la) a = cn[b];
#pragma unroll
    for (int i = 0; i < RANK; i++) atomicAdd(&P.f[dof(a, i)], R.fu[i]);
    if constexpr (PF) atomicAdd(&P.f[dof(a, RANK)], R.fp);
    if constexpr (WANT_K) {
      for (int b = 0; b < NPE; b++) {
        for (int i = 0; i < RANK; i++) {
          const int64_t row = dof(a, i);
          for (int j = 0; j < RANK; j++) {
            const int64_t p = find_col(P, row, dof(cn[b], j));
            if (p >= 0) atomicAdd(&P.K[p], R.Kuu[i][b * RANK + j]);
          }
          if constexpr (KEEP) {
            if (P.keep2) {
              const int64_t p = find_col(P, row, dof(cn[b], RANK));
              if (p >= 0) atomicAdd(&P.K[p], R.Kup[i][b]);
            }
          }
        }
        if constexpr (PF) {
          const int64_t row = dof(a, RANK);
          const int64_t p = find_col(P, row, dof(cn[b], RANK));
          if (p >= 0) atomicAdd(&P.K[p], R.Kpp[b]);
          if constexpr (KEEP) {
            if (P.keep3) {
              for (int j = 0; j < RANK; j++) {
                const int64_t q = find_col(P, row, dof(cn[b], j));
                if (q >= 0) atomicAdd(&P.K[q], R.Kpu[b * RANK + j]);
              }
            }
          }
        }
      }
    }
  }
  if (PF && P.err_part != nullptr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) err_acc += __shfl_down(err_acc, o, 64);
    if ((tid & 63) == 0) sErr[tid >> 6] = err_acc;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int k = 0; k < kBlock / 64; k++) s += sErr[k];
      P.err_part[blockIdx.x] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
template <class SH, int FAM, bool WANT_K, bool KEEP>
static int launch_one(const Params& P, bool rows, int grid, hipStream_t st) {
  if (rows) hipLaunchKernelGGL((assemble_rows<SH, FAM, WANT_K, KEEP>), dim3(grid), dim3(SH::BLOCK), 0, st, P);
  else      hipLaunchKernelGGL((assemble_generic<SH, FAM, WANT_K, KEEP>), dim3(grid), dim3(SH::BLOCK), 0, st, P);
  return (int)hipGetLastError();
}

template <class SH>
static int launch_shape(const Params& P, int fam, bool want_K, bool keep, bool rows, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (fam == FAM_LE)
    return want_K ? launch_one<SH, FAM_LE, true, false>(P, rows, grid, st)
                  : launch_one<SH, FAM_LE, false, false>(P, rows, grid, st);
  if (fam == FAM_PF_ISO) {
    if (!want_K) return launch_one<SH, FAM_PF_ISO, false, false>(P, rows, grid, st);
    return keep ? launch_one<SH, FAM_PF_ISO, true, true>(P, rows, grid, st)
                : launch_one<SH, FAM_PF_ISO, true, false>(P, rows, grid, st);
  }
  return -1000;   // family not built for this shape
}

}  // namespace fa
