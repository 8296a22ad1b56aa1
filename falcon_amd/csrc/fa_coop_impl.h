// fa_coop_impl.h -- cooperative element kernel for the higher-order shapes (Triangle3/Gauss3, Triangle6, Quad8, Hex20).
//
// With 6..20 nodes and up to 27 Gauss points an element no longer fits the registers of one lane, so a workgroup
// takes EPB elements at a time and stages everything in LDS:
//   phase 1  gather       one thread per (element, node): connectivity, coordinates, state
//   phase 2  Gauss points one thread per (element, point): Jacobian, gradients, B u, degradation, material update,
//                         history -- the gradients and the constitutive record go to LDS, each point is evaluated once
//   phase 3  blocks       one thread per (element, node a) for f1 / f2 and per (element, node pair a,b) for the
//                         (rank+1)^2 block of K1..K4; the point records are broadcast reads from LDS
// and the blocks are added to HBM with FP64 atomics (MatrixBuilder::addBlock semantics).  For the canonical blocked
// CSR layout the entry index comes from the `pos` table, otherwise from a per-entry column search.
#pragma once
#include "fa_kernel_impl.h"

// the node / Gauss-point loops stay rolled: unrolled 20 x 27 they need > 256 VGPRs and spill
#ifndef FA_COOP_UNROLL_A
#define FA_COOP_UNROLL_A 1
#endif
#ifndef FA_COOP_UNROLL_IP
#define FA_COOP_UNROLL_IP 1
#endif

namespace fa {

template <class SH, int FAM, bool KEEP>
struct CoopSlot {
  static constexpr bool PF = (FAM != FAM_LE);
  int32_t cn[SH::NPE];
  double x[SH::NPE][SH::RANK], u[SH::NPE][SH::RANK];
  double phi[PF ? SH::NPE : 1], phi0[PF ? SH::NPE : 1], phiX[PF ? SH::NPE : 1];
  double w[SH::NIP];
  double g[SH::NIP][SH::NPE][SH::RANK];
  double gphi[SH::NIP][PF ? SH::RANK : 1];
  IpMat<SH, FAM, KEEP> m[SH::NIP];
};

template <class SH, int FAM, bool WANT_K, bool KEEP, class SP = SpecRuntime>
__global__ __launch_bounds__(256) void assemble_coop(const Params P) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, NIP = SH::NIP, S = SH::S, STR = SH::STR, kBlock = 256;
  constexpr int EPB = coop_epb(RANK, NPE, NIP);
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? RANK + 1 : RANK;
  using Slot = CoopSlot<SH, FAM, KEEP>;
  static_assert(sizeof(Slot) * EPB <= 65536, "LDS budget");
  __shared__ Slot sl[EPB];
  __shared__ double sErr[kBlock / 64];
  const double* __restrict__ tabN = P.tab;
  const double* __restrict__ tabD = P.tab + NIP * NPE;
  const double* __restrict__ tabW = P.tab + NIP * NPE * (1 + RANK);

  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * EPB;
  const int ne = (int)((P.nel - e0 < EPB) ? (P.nel - e0) : EPB);
  auto dof = [&](int64_t node, int i) -> int64_t { return P.node_dofs ? P.node_dofs[node * NDOF + i] : node * NDOF + i; };

  // ---- phase 1: gather ----
  for (int t = tid; t < ne * NPE; t += kBlock) {
    const int s = t / NPE, a = t - s * NPE;
    Slot& L = sl[s];
    const int64_t n = P.conn[(e0 + s) * NPE + a];
    L.cn[a] = (int32_t)n;
#pragma unroll
    for (int i = 0; i < RANK; i++) { L.x[a][i] = P.coords[n * RANK + i]; L.u[a][i] = P.state[dof(n, i)]; }
    if constexpr (PF) {
      const int64_t dp = dof(n, RANK);
      L.phi[a] = P.state[dp];
      if (P.model == 2 || P.model == 3) { L.phi0[a] = P.state0[dp]; L.phiX[a] = P.stateX[dp]; }
      else if ((P.model == 5 || P.model == 6)) L.phiX[a] = P.stateX[dp];
    }
  }
  __syncthreads();

  // ---- phase 2: Gauss points ----
  double err_acc = 0.0;
  for (int t = tid; t < ne * NIP; t += kBlock) {
    const int s = t / NIP, ip = t - s * NIP;
    Slot& L = sl[s];
    const int64_t e = e0 + s;
    double J[RANK][RANK];
#pragma unroll
    for (int i = 0; i < RANK; i++)
#pragma unroll
      for (int k = 0; k < RANK; k++) J[i][k] = 0.0;
#pragma unroll FA_COOP_UNROLL_A
    for (int a = 0; a < NPE; a++) {
#pragma unroll
      for (int k = 0; k < RANK; k++) {
        const double dn = tabD[(ip * NPE + a) * RANK + k];
#pragma unroll
        for (int i = 0; i < RANK; i++) J[i][k] += L.x[a][i] * dn;
      }
    }
    double inv[RANK][RANK], det;
    if constexpr (RANK == 2) {
      det = J[0][0] * J[1][1] - J[0][1] * J[1][0];
      const double rd = 1.0 / det;
      inv[0][0] = J[1][1] * rd;  inv[0][1] = -J[0][1] * rd;
      inv[1][0] = -J[1][0] * rd; inv[1][1] = J[0][0] * rd;
    } else {
      const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
      const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
      const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
      det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
      const double rd = 1.0 / det;
      inv[0][0] = c00 * rd;
      inv[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * rd;
      inv[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * rd;
      inv[1][0] = c01 * rd;
      inv[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * rd;
      inv[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * rd;
      inv[2][0] = c02 * rd;
      inv[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * rd;
      inv[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * rd;
    }
    L.w[ip] = tabW[ip] * det;
    double eps[S], pf = 0.0, pf0 = 0.0, pfX = 0.0, gphi[RANK];
#pragma unroll
    for (int c = 0; c < S; c++) eps[c] = 0.0;
#pragma unroll
    for (int i = 0; i < RANK; i++) gphi[i] = 0.0;
#pragma unroll FA_COOP_UNROLL_A
    for (int a = 0; a < NPE; a++) {
      double ga[RANK], ua[RANK];
#pragma unroll
      for (int i = 0; i < RANK; i++) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < RANK; k++) acc += tabD[(ip * NPE + a) * RANK + k] * inv[k][i];
        ga[i] = acc; L.g[ip][a][i] = acc; ua[i] = L.u[a][i];
      }
      if constexpr (RANK == 2) {                       // strain = B u (ShapeUtils.cpp:73-104)
        eps[0] += ga[0] * ua[0]; eps[1] += ga[1] * ua[1];
        eps[2] += ga[1] * ua[0]; eps[2] += ga[0] * ua[1];
      } else {                                         // ShapeUtils.cpp:114-145
        eps[0] += ga[0] * ua[0]; eps[1] += ga[1] * ua[1]; eps[2] += ga[2] * ua[2];
        eps[3] += ga[1] * ua[0]; eps[3] += ga[0] * ua[1];
        eps[4] += ga[2] * ua[1]; eps[4] += ga[1] * ua[2];
        eps[5] += ga[0] * ua[2]; eps[5] += ga[2] * ua[0];
      }
      if constexpr (PF) {
        const double Na = tabN[ip * NPE + a], ph = L.phi[a];
        pf += Na * ph;
#pragma unroll
        for (int i = 0; i < RANK; i++) gphi[i] += ga[i] * ph;
        if (P.model == 2 || P.model == 3) { pf0 += Na * L.phi0[a]; pfX += Na * L.phiX[a]; }
        else if ((P.model == 5 || P.model == 6)) pfX += Na * L.phiX[a];
      }
    }
    const int64_t ipoint = e * NIP + ip;               // initializeIPMPMap_, PhaseFractureModel.cpp:1070-1093
    if (P.strain != nullptr) {
      double* sc = P.strain + ipoint * (PF ? STR + 1 : STR);
      if constexpr (RANK == 2) { sc[0] = eps[0]; sc[1] = eps[1]; sc[2] = 0.0; sc[3] = eps[2]; }
      else {
#pragma unroll
        for (int c = 0; c < S; c++) sc[c] = eps[c];
      }
    }
    double H0 = 0.0;
    if constexpr (PF) {
      if (P.model == 1 || P.model >= 4) H0 = P.hist_pre[ipoint];
#pragma unroll
      for (int i = 0; i < RANK; i++) L.gphi[ip][i] = gphi[i];
    }
    IpMat<SH, FAM, KEEP> m;
    ip_material<SH, FAM, KEEP, SP>(P, ipoint, eps, pf, pf0, pfX, H0, true, m, err_acc);
    L.m[ip] = m;
  }
  __syncthreads();

  // ---- phase 3a: f1 += w B^T sigma, f2 (PhaseFractureModel.cpp:727-730) ----
  for (int t = tid; t < ne * NPE; t += kBlock) {
    const int s = t / NPE, a = t - s * NPE;
    const Slot& L = sl[s];
    double fu[RANK], fp = 0.0;
#pragma unroll
    for (int i = 0; i < RANK; i++) fu[i] = 0.0;
#pragma unroll FA_COOP_UNROLL_IP
    for (int ip = 0; ip < NIP; ip++) {
      const double w = L.w[ip];
      double wga[RANK], tt[RANK], sig[S];
#pragma unroll
      for (int i = 0; i < RANK; i++) wga[i] = w * L.g[ip][a][i];
#pragma unroll
      for (int c = 0; c < S; c++) sig[c] = L.m[ip].sig[c];
      Bt_vec<RANK, S>(wga, sig, tt);
#pragma unroll
      for (int i = 0; i < RANK; i++) fu[i] += tt[i];
      if constexpr (PF) {
        double gg = 0.0;
#pragma unroll
        for (int i = 0; i < RANK; i++) gg += wga[i] * L.gphi[ip][i];
        fp += (w * tabN[ip * NPE + a]) * L.m[ip].cf + P.c2 * gg;
      }
    }
    const int64_t n = L.cn[a];
#pragma unroll
    for (int i = 0; i < RANK; i++) atomicAdd(&P.f[dof(n, i)], fu[i]);
    if constexpr (PF) atomicAdd(&P.f[dof(n, RANK)], fp);
  }

  // ---- phase 3b: node-pair blocks of K1..K4 (PhaseFractureModel.cpp:713-722) ----
  if constexpr (WANT_K) {
    for (int t = tid; t < ne * NPE * NPE; t += kBlock) {
      const int s = t / (NPE * NPE), ab = t - s * (NPE * NPE);
      const int a = ab / NPE, b = ab - a * NPE;
      const Slot& L = sl[s];
      double Kuu[RANK][RANK], Kpp = 0.0, Kup[RANK], Kpu[RANK];
#pragma unroll
      for (int i = 0; i < RANK; i++) {
        Kup[i] = 0.0; Kpu[i] = 0.0;
#pragma unroll
        for (int j = 0; j < RANK; j++) Kuu[i][j] = 0.0;
      }
#pragma unroll FA_COOP_UNROLL_IP
      for (int ip = 0; ip < NIP; ip++) {
        const IpMat<SH, FAM, KEEP>& m = L.m[ip];
        const double w = L.w[ip];
        double wga[RANK], gb[RANK];
#pragma unroll
        for (int i = 0; i < RANK; i++) { wga[i] = w * L.g[ip][a][i]; gb[i] = L.g[ip][b][i]; }
        if constexpr (FAM == FAM_PF_MIEHE) {
#pragma unroll
          for (int i = 0; i < RANK; i++) {             // row i of w B_a^T D, then (.) B_b
            double Bi[S], row[S], tt[RANK];
#pragma unroll
            for (int c = 0; c < S; c++) Bi[c] = 0.0;
            if constexpr (RANK == 2) { if (i == 0) { Bi[0] = wga[0]; Bi[2] = wga[1]; } else { Bi[1] = wga[1]; Bi[2] = wga[0]; } }
            else {
              if (i == 0) { Bi[0] = wga[0]; Bi[3] = wga[1]; Bi[5] = wga[2]; }
              else if (i == 1) { Bi[1] = wga[1]; Bi[3] = wga[0]; Bi[4] = wga[2]; }
              else { Bi[2] = wga[2]; Bi[4] = wga[1]; Bi[5] = wga[0]; }
            }
#pragma unroll
            for (int c = 0; c < S; c++) {
              double acc = 0.0;
#pragma unroll
              for (int k = 0; k < S; k++) acc += Bi[k] * m.D[k][c];
              row[c] = acc;
            }
            Bt_vec<RANK, S>(gb, row, tt);
#pragma unroll
            for (int j = 0; j < RANK; j++) Kuu[i][j] += tt[j];
          }
        } else if constexpr (RANK == 2) {
          const double d0 = m.d0, d1 = m.d1, d3 = m.d3;
          Kuu[0][0] += (wga[0] * d0) * gb[0] + (wga[1] * d3) * gb[1];
          Kuu[0][1] += (wga[0] * d1) * gb[1] + (wga[1] * d3) * gb[0];
          Kuu[1][0] += (wga[1] * d1) * gb[0] + (wga[0] * d3) * gb[1];
          Kuu[1][1] += (wga[1] * d0) * gb[1] + (wga[0] * d3) * gb[0];
        } else {
          const double d0 = m.d0, d1 = m.d1, d3 = m.d3;
          const double dotp = wga[0] * gb[0] + wga[1] * gb[1] + wga[2] * gb[2];
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
              if (i == j) Kuu[i][j] += (d0 - d3) * (wga[i] * gb[i]) + d3 * dotp;
              else        Kuu[i][j] += d1 * (wga[i] * gb[j]) + d3 * (wga[j] * gb[i]);
            }
        }
        if constexpr (PF) {
          const double Na = tabN[ip * NPE + a], Nb = tabN[ip * NPE + b];
          const double wNa = w * Na;
          double gab = 0.0;
#pragma unroll
          for (int i = 0; i < RANK; i++) gab += wga[i] * gb[i];
          Kpp += (wNa * m.c1) * Nb + P.c2 * gab;                        // K4 (:714-716)
          if constexpr (KEEP) {
            double spv[S], tt[RANK];
#pragma unroll
            for (int c = 0; c < S; c++) spv[c] = m.sp[c];
            if (P.keep2) {                                              // K2 (:720)
              if (P.k2sig) {                                            // GradientDamage / Mazars: D0 eps travels as sig / (1 - omega)
                double sv[S];
#pragma unroll
                for (int c = 0; c < S; c++) sv[c] = m.sig[c];
                Bt_vec<RANK, S>(wga, sv, tt);
              } else Bt_vec<RANK, S>(wga, spv, tt);
#pragma unroll
              for (int i = 0; i < RANK; i++) Kup[i] += (m.dg * tt[i]) * Nb;
            }
            if (P.keep3) {                                              // K3 (:721)
              Bt_vec<RANK, S>(gb, spv, tt);
#pragma unroll
              for (int j = 0; j < RANK; j++) Kpu[j] += (wNa * m.dgl) * tt[j];
            }
          }
        }
      }
      // ---- addBlock ----
      const int64_t na = L.cn[a], nb = L.cn[b];
      const int64_t e = e0 + s;
      const int pab = P.canon ? (int)P.pos[(e * NPE + a) * NPE + b] : 0;
      // every entry written here exists: the canonical layouts hold the blocks of this model by construction
      // (off[i][j] >= 0 for them) and fa_create has verified a caller's pattern entry by entry -- nothing is dropped
      auto entry = [&](int i, int j) -> int64_t {
        if (P.canon) return P.row_ptr[na * NDOF + i] + (int64_t)pab * P.cnt[i] + P.off[i][j];
        return find_col(P, dof(na, i), dof(nb, j));
      };
#pragma unroll
      for (int i = 0; i < RANK; i++) {
#pragma unroll
        for (int j = 0; j < RANK; j++) atomicAdd(&P.K[entry(i, j)], Kuu[i][j]);
        if constexpr (KEEP) { if (P.keep2) atomicAdd(&P.K[entry(i, RANK)], Kup[i]); }
      }
      if constexpr (PF) {
        atomicAdd(&P.K[entry(RANK, RANK)], Kpp);
        if constexpr (KEEP) {
          if (P.keep3) {
#pragma unroll
            for (int j = 0; j < RANK; j++) atomicAdd(&P.K[entry(RANK, j)], Kpu[j]);
          }
        }
      }
    }
  }
  if (PF && P.err_part != nullptr) reduce_err<kBlock>(err_acc, sErr, &P.err_part[blockIdx.x]);
}

template <class SH, int FAM, bool WANT_K, bool KEEP>
static int launch_coop_one(const Params& P, int grid, hipStream_t st) {
  if constexpr (FAM == FAM_PF_ISO) {                   // GradientDamage: its own instantiation (SpecGD)
    if (P.model == 7) { hipLaunchKernelGGL((assemble_coop<SH, FAM, WANT_K, KEEP, SpecGD<SpecRuntime>>), dim3(grid), dim3(256), 0, st, P); return (int)hipGetLastError(); }
  }
  if constexpr (FAM == FAM_PF_MIEHE) {                 // MicroPhaseFracture* + AmorPhase (SpecMA)
    if (is_micro_amor(P)) { hipLaunchKernelGGL((assemble_coop<SH, FAM, WANT_K, KEEP, SpecMA<SpecRuntime>>), dim3(grid), dim3(256), 0, st, P); return (int)hipGetLastError(); }
  }
  hipLaunchKernelGGL((assemble_coop<SH, FAM, WANT_K, KEEP, SpecRuntime>), dim3(grid), dim3(256), 0, st, P);
  return (int)hipGetLastError();
}

// same signature as the row-kernel launchers; `path` is ignored (one kernel), grid = ceil(nel / coop_epb)
template <class SH>
static int launch_coop_shape(const Params& P, int fam, bool want_K, bool keep, int /*path*/, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (fam == FAM_LE)
    return want_K ? launch_coop_one<SH, FAM_LE, true, false>(P, grid, st) : launch_coop_one<SH, FAM_LE, false, false>(P, grid, st);
  if (fam == FAM_PF_ISO) {
    if (!want_K) return launch_coop_one<SH, FAM_PF_ISO, false, false>(P, grid, st);
    return keep ? launch_coop_one<SH, FAM_PF_ISO, true, true>(P, grid, st) : launch_coop_one<SH, FAM_PF_ISO, true, false>(P, grid, st);
  }
  if (fam == FAM_PF_MIEHE) {
    if (!want_K) return launch_coop_one<SH, FAM_PF_MIEHE, false, false>(P, grid, st);
    return keep ? launch_coop_one<SH, FAM_PF_MIEHE, true, true>(P, grid, st) : launch_coop_one<SH, FAM_PF_MIEHE, true, false>(P, grid, st);
  }
  return -1000;
}

}  // namespace fa
