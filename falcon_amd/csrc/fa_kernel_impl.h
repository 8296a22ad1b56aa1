// fa_kernel_impl.h -- gfx950 kernels for the element-level K / f_int assembly.
//
// The kernels share one Gauss-point core (ip_point / ip_material / accum_rows):
//
//  assemble_rows_pair  (default for Quad4)  "owner computes" on tiles of consecutive nodes: the slice of the CSR value
//      array that holds the tile's rows is mirrored in LDS.  Lanes come in aligned pairs working on one element;
//      each evaluates half of the Gauss points (Jacobian, gradients, strain, degradation, material update, history),
//      the two swap every Gauss-point record with a DPP quad permute and each keeps only the rows of its own node
//      of K1..K4 / f1,f2 (PhaseFractureModel.cpp:713-730).  Row blocks go into the LDS image with ds_add_f64; the
//      image is streamed to HBM with 16-byte coalesced stores: every K entry written once, no global atomics.
//  assemble_rows   (Triangle3, Hex8; FA_DETERMINISTIC)  same tiles, one lane per (node, incident element) pair
//      evaluating all Gauss points.  With FA_DETERMINISTIC the lanes add in phases ordered by the element's slot in
//      the node's element list, so every CSR entry is summed in ascending element order like the reference's
//      serial loop (bitwise reproducible).
//  assemble_generic  any dof map / any CSR pattern: per-entry column search + FP64 atomics, like
//      MatrixBuilder::addBlock.
//  assemble_patch  (opt-in, FA_USE_PATCH; experimental)  2-D/3-D node patches, NPE lanes per element sharing the
//      Gauss points; correct but slower than the row kernels.
//  (fa_coop_impl.h: assemble_coop for the higher-order shapes.)
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fa_internal.h"

namespace fa {

// The shape tables are never written: read through the constant address space their loads stay scalar (s_load)
// wherever the index is uniform, even after the kernel's own global stores (history, strain cache); through a plain
// global pointer they become vector loads with a memory round trip inside the Gauss-point loop.
typedef const __attribute__((address_space(4))) double* const_dptr;
typedef const __attribute__((address_space(4))) int32_t* const_iptr;

template <int RANK_, int NPE_, int NIP_>
struct Shape {
  static constexpr int RANK = RANK_, NPE = NPE_, NIP = NIP_;
  static constexpr int S   = (RANK_ == 2) ? 3 : 6;   // strain components that matter (2D: xx, yy, xy)
  static constexpr int STR = (RANK_ == 2) ? 4 : 6;   // STRAIN_COUNTS (util/Constants.cpp:9)
  static constexpr int ND  = RANK_ * NPE_;
  static constexpr int BLOCK = block_for(RANK_), LDS = lds_for(RANK_);
  static constexpr int PBLOCK = pblock_for(NPE_);
  static constexpr bool SHARE = (NIP_ == NPE_) && (64 % NPE_ == 0);   // group shares the Gauss points
};

// What a kernel instantiation knows at compile time about the model.  SpecRuntime reads everything from Params
// (uniform branches); SpecFixed folds model / degradation exponent / material into the code and promises the
// canonical dof numbering (the row kernels' precondition), which removes every uniform branch from the Gauss-point loop.
struct SpecRuntime {
  static constexpr bool ma = false;                    // MicroPhaseFracture* + AmorPhase: SpecMA
  static constexpr bool gd_fixed = false;              // GradientDamage with law and strain kind known at compile time
  static constexpr bool gd = false;                    // GradientDamage has kernels of its own (SpecGD): its constitutive
                                                       // branch (exp, acos, cos) costs every runtime-model kernel registers
  static __device__ __forceinline__ int model(const Params& P) { return P.model; }
  static __device__ __forceinline__ int ptype(const Params& P) { return P.ptype; }
  static __device__ __forceinline__ int material(const Params& P) { return P.material; }
  static __device__ __forceinline__ bool custom_dofs(const Params& P) { return P.node_dofs != nullptr; }
};
struct SpecCanon : SpecRuntime {                       // runtime model, canonical dofs
  static __device__ __forceinline__ bool custom_dofs(const Params&) { return false; }
};
template <int MODEL, int PTYPE, int MAT>
struct SpecFixed {
  static constexpr bool gd = false, ma = false, gd_fixed = false;
  static __device__ __forceinline__ constexpr int model(const Params&) { return MODEL; }
  static __device__ __forceinline__ constexpr int ptype(const Params&) { return PTYPE; }
  static __device__ __forceinline__ constexpr int material(const Params&) { return MAT; }
  static __device__ __forceinline__ constexpr bool custom_dofs(const Params&) { return false; }
};

// GradientDamage (model 7) on top of a runtime spec: the only instantiations that contain DamageMaterial's update
// SOFT / EQV >= 0: the softening law / equivalent-strain kind are compile-time constants (the eleven-way law switch and
// the two unused strain routines drop out: Quad4 force-vector kernel 216 -> 191 VGPRs with Mazars, 128 with Energy; Hex8
// 256 + 96 bytes of scratch -> 209).  Instantiated for exponential1 with Mazars (the material of every GradientDamage
// input of the reference, tests/sm03 included) and with Energy; every other combination runs the runtime form (-1, -1).
template <class BASE, int SOFT = -1, int EQV = -1>
struct SpecGD : BASE {
  static constexpr bool gd = true, gd_fixed = (SOFT >= 0 && EQV >= 0);
  static __device__ __forceinline__ constexpr int model(const Params&) { return 7; }
  static __device__ __forceinline__ int soft(const Params& P) { if constexpr (SOFT >= 0) return SOFT; else return P.soft; }
  static __device__ __forceinline__ int eqv(const Params& P) { if constexpr (EQV >= 0) return EQV; else return P.eqv; }
};
// launch helper: go(spec) with the compile-time form where one exists
template <class BASE, class F>
static inline auto gd_dispatch(const Params& P, F&& go) {
  if (P.soft == 3 && P.eqv == 1) return go(SpecGD<BASE, 3, 1>{});
  if (P.soft == 3 && P.eqv == 0) return go(SpecGD<BASE, 3, 0>{});
  return go(SpecGD<BASE>{});
}

// MicroPhaseFracture / Ext / ExtIt with AmorPhase (general-tangent family): the only instantiations with its split update
template <class BASE>
struct SpecMA : BASE {
  static constexpr bool ma = true;
  static __device__ __forceinline__ constexpr int material(const Params&) { return 1; }
};
static inline bool is_micro_amor(const Params& P) { return P.model >= 4 && P.model <= 6 && P.material == 1; }

template <int I, int N>
struct SFor {
  template <class F> static __device__ __forceinline__ void run(F& f) { f(std::integral_constant<int, I>{}); SFor<I + 1, N>::run(f); }
};
template <int N>
struct SFor<N, N> { template <class F> static __device__ __forceinline__ void run(F&) {} };

template <class SH, bool WANT_K, bool PF, bool KEEP>
struct RowBlock {
  double fu[SH::RANK];
  double fp;
  double Kuu[WANT_K ? SH::RANK : 1][WANT_K ? SH::ND : 1];
  double Kpp[(WANT_K && PF) ? SH::NPE : 1];
  double Kup[(WANT_K && KEEP) ? SH::RANK : 1][(WANT_K && KEEP) ? SH::NPE : 1];
  double Kpu[(WANT_K && KEEP) ? SH::ND : 1];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < SH::RANK; i++) fu[i] = 0.0;
    fp = 0.0;
    if constexpr (WANT_K) {
#pragma unroll
      for (int i = 0; i < SH::RANK; i++)
#pragma unroll
        for (int j = 0; j < SH::ND; j++) Kuu[i][j] = 0.0;
      if constexpr (PF) {
#pragma unroll
        for (int b = 0; b < SH::NPE; b++) Kpp[b] = 0.0;
      }
      if constexpr (KEEP) {
#pragma unroll
        for (int i = 0; i < SH::RANK; i++)
#pragma unroll
          for (int b = 0; b < SH::NPE; b++) Kup[i][b] = 0.0;
#pragma unroll
        for (int j = 0; j < SH::ND; j++) Kpu[j] = 0.0;
      }
    }
  }
};

// Constitutive result of one Gauss point (what Material::update and the degradation terms leave behind).
template <class SH, int FAM, bool KEEP>
struct IpMat {
  static constexpr bool PF = (FAM != FAM_LE), GEN = (FAM == FAM_PF_MIEHE);
  double d0, d1, d3;                 // tangent D in isotropic form
  double D[GEN ? SH::S : 1][GEN ? SH::S : 1];   // general symmetric tangent (Miehe)
  double sig[SH::S];                 // stress
  double sp[KEEP ? SH::S : 1];       // sigma+ = dPsi/deps (off-diagonal blocks only)
  double c1, cf;                     // N N^T coefficient of K4; N coefficient of f2
  double dg, dgl;                    // g'(phi), loading * g'(phi)
};

// Everything the row accumulation needs from one Gauss point.
template <class SH, int FAM, bool KEEP>
struct IpRec : IpMat<SH, FAM, KEEP> {
  double w;                          // w_ref * det J
  double g[SH::NPE][SH::RANK];       // dN_a/dx_i
  double gphi[(FAM != FAM_LE) ? SH::RANK : 1];    // grad phi
};

// gathered nodal data of one element
template <class SH, bool PF>
struct ElemData {
  double x[SH::NPE][SH::RANK], u[SH::NPE][SH::RANK];
  double phi[PF ? SH::NPE : 1], phi0[PF ? SH::NPE : 1], phiX[PF ? SH::NPE : 1];
};

template <class SH, bool PF, class SP = SpecRuntime>
__device__ __forceinline__ void gather_elem(const Params& P, const int32_t (&cn)[SH::NPE], ElemData<SH, PF>& E) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, NDOF = PF ? RANK + 1 : RANK;
#pragma unroll
  for (int a = 0; a < NPE; a++) {
    const int64_t n = cn[a];
    // XDofSpace::getDofIndices: node-major, type-minor unless the caller supplied its own dof map
    int64_t dofs[NDOF];
#pragma unroll
    for (int i = 0; i < NDOF; i++) dofs[i] = SP::custom_dofs(P) ? P.node_dofs[n * NDOF + i] : n * NDOF + i;
#pragma unroll
    for (int i = 0; i < RANK; i++) { E.x[a][i] = P.coords[n * RANK + i]; E.u[a][i] = P.state[dofs[i]]; }
    if constexpr (PF) {
      E.phi[a] = P.state[dofs[RANK]];
      if (SP::model(P) == 2 || SP::model(P) == 3) { E.phi0[a] = P.state0[dofs[RANK]]; E.phiX[a] = P.stateX[dofs[RANK]]; }
      else if ((SP::model(P) == 5 || SP::model(P) == 6)) { E.phi0[a] = 0.0; E.phiX[a] = P.stateX[dofs[RANK]]; }   // stateExt_ only
      else { E.phi0[a] = 0.0; E.phiX[a] = 0.0; }
    }
  }
}

// rational degradation function and its derivatives (PhaseFractureModel.cpp:660-676).
// ptype: 0 -> p = 2, 1 -> p = 2.5, 2 -> p = 4; pow(x,0) == 1 also at x == 0.
template <class SP = SpecRuntime>
__device__ __forceinline__ void degrade(const Params& P, double pf, double& g, double& dg, double& ddg) {
  const double x = 1.0 - pf;
  double gn, dgn, ddgn;
  if (SP::ptype(P) == 0)      { gn = x * x;               dgn = -2.0 * x;            ddgn = 2.0; }
  else if (SP::ptype(P) == 1) { const double sq = sqrt(x);
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
template <int RANK, int S, class SP = SpecRuntime>
__device__ __forceinline__ void iso_update(const Params& P, const double (&e)[S], double g,
                                           double& d0, double& d1, double& d3,
                                           double (&sig)[S], double (&sp)[S], double& Psi) {
  constexpr int NN = RANK;                                  // normal components carried
  const double tr = (RANK == 3) ? (e[0] + e[1] + e[2]) : (e[0] + e[1] + 0.0);
  if (SP::material(P) == 1) {
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

// One Jacobi rotation in the (p,q) plane of the symmetric 3x3 matrix a, eigenvector columns accumulated in V
// (same operation order as the checker's cyclic Jacobi so that both resolve near-degenerate strains alike).
template <int p, int q, int N>
__device__ __forceinline__ void jacobi_rot(double (&a)[3][3], double (&V)[3][3]) {
  if (a[p][q] == 0.0) return;
  const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
  const double t = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
  for (int k = 0; k < N; k++) { const double akp = a[k][p], akq = a[k][q]; a[k][p] = c * akp - s * akq; a[k][q] = s * akp + c * akq; }
#pragma unroll
  for (int k = 0; k < N; k++) { const double apk = a[p][k], aqk = a[q][k]; a[p][k] = c * apk - s * aqk; a[q][k] = s * apk + c * aqk; }
#pragma unroll
  for (int k = 0; k < N; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
}

// MiehePhaseMaterial::update + getposProj_ (MiehePhaseMaterial.cpp:200-258,342-406) in Voigt form.  With
// M_a = n_a n_a^T and S_ab = n_a n_b^T + n_b n_a^T the projector of :378-404 is
//   P+ = sum_a d_a M_a (x) M_a + sum_{b<a} theta_ab S_ab (x) S_ab         (G_ab + G_ba == S_ab (x) S_ab),
// so tensor2voigt(P+) = sum d_a m_a m_a^T + sum theta_ab s_ab s_ab^T with the Voigt vectors of M_a, S_ab.
// 2D: the strain tensor has no z entries; the z eigen-direction (eigenvalue 0, d = 0) only feeds zz / out-of-plane
// shear entries, which B never touches, so the in-plane 2x2 problem is solved.
// gfun(Psi, sigma+) supplies the degradation value once the g-independent part is known (a constant for the
// PhaseFracture models, the local Newton iteration of the micromorphic model).
template <int RANK, int S, class GF>
__device__ __forceinline__ void miehe_update_f(const Params& P, const double (&e)[S], GF&& gfun, double (&D)[S][S],
                                               double (&sig)[S], double (&sp)[S], double& Psi) {
  constexpr int N = RANK;
  constexpr int VI[6] = { 0, 1, 2, 0, 1, 2 }, VJ[6] = { 0, 1, 2, 1, 2, 0 };   // 3D Voigt: xx yy zz xy yz zx
  auto vi = [&](int I) { return (RANK == 2) ? (I == 2 ? 0 : I) : VI[I]; };
  auto vj = [&](int I) { return (RANK == 2) ? (I == 2 ? 1 : I) : VJ[I]; };
  double a[3][3], V[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) { a[i][j] = 0.0; V[i][j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
  for (int I = 0; I < S; I++) {                        // voigt2tensorStrain: shear halved (TensorUtils.cpp:467-492)
    const double v = (I < N) ? e[I] : 0.5 * e[I];
    a[vi(I)][vj(I)] = v; a[vj(I)][vi(I)] = v;
  }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off, dia;
    if constexpr (RANK == 2) { off = a[0][1] * a[0][1]; dia = a[0][0] * a[0][0] + a[1][1] * a[1][1]; }
    else { off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
           dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2]; }
    if (off <= 1e-300 || off <= 1e-34 * dia) break;
    jacobi_rot<0, 1, N>(a, V);
    if constexpr (RANK == 3) { jacobi_rot<0, 2, N>(a, V); jacobi_rot<1, 2, N>(a, V); }
  }
  double val[N], epos[N], dg[N];
#pragma unroll
  for (int r = 0; r < N; r++) { val[r] = a[r][r]; epos[r] = 0.5 * (val[r] + fabs(val[r])); dg[r] = (val[r] > 0.0) ? 1.0 : 0.0; }
  double Pv[S][S];
#pragma unroll
  for (int I = 0; I < S; I++)
#pragma unroll
    for (int J = 0; J < S; J++) Pv[I][J] = 0.0;
#pragma unroll
  for (int r = 0; r < N; r++) {
    double m[S];
#pragma unroll
    for (int I = 0; I < S; I++) m[I] = V[vi(I)][r] * V[vj(I)][r];
#pragma unroll
    for (int I = 0; I < S; I++)
#pragma unroll
      for (int J = I; J < S; J++) Pv[I][J] += (m[I] * m[J]) * dg[r];
  }
#pragma unroll
  for (int r = 0; r < N; r++)
#pragma unroll
    for (int b = 0; b < r; b++) {
      double th;
      if (fabs(val[r] - val[b]) <= 1e-13) th = 0.5 * (dg[r] + dg[b]) / 2.0;
      else                                th = 0.5 * (epos[r] - epos[b]) / (val[r] - val[b]);
      double sv[S];
#pragma unroll
      for (int I = 0; I < S; I++) sv[I] = V[vi(I)][r] * V[vj(I)][b] + V[vi(I)][b] * V[vj(I)][r];
#pragma unroll
      for (int I = 0; I < S; I++)
#pragma unroll
        for (int J = I; J < S; J++) Pv[I][J] += th * (sv[I] * sv[J]);
    }
#pragma unroll
  for (int I = 0; I < S; I++)
#pragma unroll
    for (int J = 0; J < I; J++) Pv[I][J] = Pv[J][I];
  // strainPos = P+ : T (engineering shear carries the factor 2 of the kl / lk pair)
  double spos[S];
#pragma unroll
  for (int I = 0; I < S; I++) {
    double s = 0.0;
#pragma unroll
    for (int J = 0; J < S; J++) s += Pv[I][J] * e[J];
    spos[I] = s;
  }
  double tr = 0.0;
#pragma unroll
  for (int i = 0; i < N; i++) tr += e[i];
  const double trP = 0.5 * (tr + fabs(tr)), trN = 0.5 * (tr - fabs(tr));
  double dd = 0.0;
#pragma unroll
  for (int I = 0; I < S; I++) dd += ((I < N) ? 1.0 : 2.0) * (spos[I] * spos[I]);
  Psi = 0.5 * P.lmda * (trP * trP) + P.G * dd;
#pragma unroll
  for (int I = 0; I < S; I++) sp[I] = ((I < N) ? P.lmda * trP : 0.0) + 2.0 * P.G * spos[I];
  const double g = gfun(Psi, sp);
#pragma unroll
  for (int I = 0; I < S; I++) {
    const double T = (I < N) ? e[I] : 0.5 * e[I];
    const double vN = ((I < N) ? P.lmda * trN : 0.0) + 2.0 * P.G * (T - spos[I]);
    sig[I] = g * sp[I] + vN;
  }
  const double H = (tr < 0.0) ? 0.0 : 1.0;            // heavisideP(0) == 1 (MathUtils.cpp:33-36)
#pragma unroll
  for (int I = 0; I < S; I++)
#pragma unroll
    for (int J = 0; J < S; J++) {
      const double I4 = (I == J) ? ((I < N) ? 1.0 : 0.5) : 0.0;
      const double II = (I < N && J < N) ? 1.0 : 0.0;
      D[I][J] = g * (2.0 * P.G * Pv[I][J] + P.lmda * H * II) + (2.0 * P.G * (I4 - Pv[I][J]) + P.lmda * (1.0 - H) * II);
    }
}

template <int RANK, int S>
__device__ __forceinline__ void miehe_update(const Params& P, const double (&e)[S], double g, double (&D)[S][S],
                                             double (&sig)[S], double (&sp)[S], double& Psi) {
  miehe_update_f<RANK, S>(P, e, [g](double, const double (&)[S]) { return g; }, D, sig, sp, Psi);
}

// Local damage of the micromorphic model (MicroPhaseFractureModel.cpp:668-796): Newton iteration on
// Res(d) = g'(d) Psi + gc/(cw l0) w'(d) + alpha (d - pf), |Res| <= 1e-4 or 100 iterations, then the loading /
// clipping branches.  Returns g(d) (no floor); out: d, g'(d) at the final d, dphidd and cdS with
// dphidStrain = cdS * sigma+.  In the `d > 1` branch the reference keeps a dphidStrain left over from an earlier
// Gauss point of its serial loop; there is no such thing on a GPU, cdS = 0 is used there (documented divergence).
struct MicroOut { double d, dgphi, dphidd, cdS; bool stale; };
__device__ __forceinline__ void micro_terms(const Params& P, double d, double& gn, double& gd, double& dg, double& ddg) {
  const double x = 1.0 - d;
  double dgn, ddgn;
  // The reference calls ::pow(1-d, p), ::pow(1-d, p-1), ::pow(1-d, p-2) (MicroPhaseFractureModel.cpp:708-715); glibc's pow
  // is correctly rounded in all but a vanishing share of cases.  p = 2: x*x, x, 1 are exactly that.  p = 2.5 and p = 4:
  // the powers are formed in double-double arithmetic (FMA residuals of x*x and of the IEEE square root) and rounded once,
  // which gives the correctly rounded value too -- with x*x*sqrt(x) the one-ulp error travelled through the quotient-rule
  // cancellations of g' and g'' into K at a few 1e-12 for the czmExponential law (round 2 asserted 1e-11 there).
  if (P.ptype == 0)      { gn = x * x;                  dgn = -2.0 * x;               ddgn = 2.0; }
  else {
    const double x2h = x * x, x2l = fma(x, x, -x2h);                       // x^2 = x2h + x2l exactly
    if (P.ptype == 1) {
      const double sq = sqrt(x), sql = (sq > 0.0) ? fma(-sq, sq, x) / (2.0 * sq) : 0.0;   // sqrt(x) = sq + sql
      const double p15 = x * sq, e15 = fma(x, sq, -p15) + x * sql;
      const double p25 = x2h * sq, e25 = fma(x2h, sq, -p25) + (x2h * sql + x2l * sq);
      gn = p25 + e25;                                  dgn = -2.5 * (p15 + e15);        ddgn = 2.5 * 1.5 * sq;
    } else {
      const double p3 = x2h * x, e3 = fma(x2h, x, -p3) + x2l * x;
      const double p4 = x2h * x2h, e4 = fma(x2h, x2h, -p4) + 2.0 * (x2h * x2l);
      gn = p4 + e4;                                    dgn = -4.0 * (p3 + e3);          ddgn = 4.0 * 3.0 * x2h;
    }
  }
  const double a12 = P.a1 * P.a2, a123 = a12 * P.a3;
  gd = gn + P.a1 * d + a12 * d * d + a123 * d * d * d;
  const double dgd  = dgn + P.a1 + 2.0 * a12 * d + 3.0 * a123 * d * d;
  const double ddgd = ddgn + 2.0 * a12 + 6.0 * a123 * d;
  const double q = dgn * gd - gn * dgd;
  dg  = q / (gd * gd);
  ddg = ((ddgn * gd - gn * ddgd) * gd - 2.0 * q * dgd) / (gd * gd * gd);
}
// Newton iteration proper: starts at d, returns the last iterate with g'(d), g''(d) at it.
__device__ __forceinline__ double micro_iterate(const Params& P, double Psi, double target, double d, double& dg,
                                                double& ddg) {
  const double ddw = 2.0 * (1.0 - P.eta);
  double gn, gd;
  micro_terms(P, d, gn, gd, dg, ddg);
  double Res = dg * Psi + P.gcl * (P.eta + 2.0 * (1.0 - P.eta) * d) + P.alpha * (d - target);
  double J = ddg * Psi + P.gcl * ddw + P.alpha;
  for (int iter = 0; fabs(Res) > 1.e-4 && iter < 100; iter++) {
    d -= Res / J;
    micro_terms(P, d, gn, gd, dg, ddg);
    Res = dg * Psi + P.gcl * (P.eta + 2.0 * (1.0 - P.eta) * d) + P.alpha * (d - target);
    J = ddg * Psi + P.gcl * ddw + P.alpha;
  }
  return d;
}
// dstart: where the iteration starts (d0 for MicroPhaseFracture; the first solve's damage for the Ext variants,
// MicroPhaseFractureExtModel.cpp:795-850); the branches always compare with the committed d0.
__device__ __forceinline__ double micro_newton(const Params& P, double Psi, double pf, double dstart, double d0,
                                               MicroOut& o) {
  const double ddw = 2.0 * (1.0 - P.eta);
  double gn, gd, dg, ddg;
  double d = micro_iterate(P, Psi, pf, dstart, dg, ddg);
  o.dphidd = 0.0; o.cdS = 0.0; o.stale = false;
  if (d > d0) {
    const double den = ddg * Psi + P.gcl * ddw + P.alpha;
    o.cdS = -dg / den; o.dphidd = P.alpha / den;
  } else if (d < 0.0) d = 0.0;
  else if (d > 1.0)   { d = 1.0; o.stale = true; }     // the reference leaves dphidStrain as an earlier point set it (:772-776)
  else                d = d0;
  micro_terms(P, d, gn, gd, dg, ddg);
  o.d = d; o.dgphi = dg;
  return gn / gd;
}
// First solve of the Ext variants (MicroPhaseFractureExtModel.cpp:728-790): damage for the extrapolated micromorphic
// variable, clipped to [0,1] and made irreversible; returns g(d) + 1e-10.
__device__ __forceinline__ double micro_first_solve(const Params& P, double Psi, double pfEx, double d0, double& dout) {
  double gn, gd, dg, ddg;
  double d = micro_iterate(P, Psi, pfEx, d0, dg, ddg);
  if (d < 0.0) d = 0.0; else if (d > 1.0) d = 1.0;
  d = (d > d0) ? d : d0;
  micro_terms(P, d, gn, gd, dg, ddg);
  dout = d;
  return gn / gd + 1.e-10;
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

// damageEvolution_ / getdOmegadKappa (DamageMaterial.cpp:712-855, laws :1044-1633): omega(kappa) and its derivative
__device__ __forceinline__ void damage_law(const Params& P, const double kappa, double& omega, double& domega, const int gsoft) {
  omega = 0.0; domega = 0.0;
  const double Ki = P.kI, Kc = P.kC, K = kappa;
  if (gsoft == 0) {
    if (K >= Ki) omega = 1.0 - Ki / K;
    domega = (K < Ki) ? 0.0 : Ki / (K * K);
  } else if (gsoft == 1) {
    if (K >= Ki) { omega = K > Kc ? 1.0 : Kc * (K - Ki) / (K * (Kc - Ki)); if (omega == 1.0) omega = 0.999; }
    domega = (K < Kc) ? (Ki * Kc) / ((Kc - Ki) * K * K) : 0.0;
  } else if (gsoft == 2) {
    if (K >= Ki) omega = K > Kc ? 0.999 : (K - Ki) / (Kc - Ki);
    domega = (K < Kc) ? 1.0 / (Kc - Ki) : 0.0;
  } else if (gsoft == 3) {
    const double ex = exp(-P.sB * (K - Ki));
    if (K >= Ki) omega = 1.0 - (Ki / K) * (1.0 - P.sA + P.sA * ex);
    const double a = Ki / (K * K), b = a * K, c = P.sA * ex;
    domega = a * (1.0 - P.sA + c) + P.sB * b * c;
  } else if (gsoft == 4) {
    const double ex = exp((Ki - K) / (Kc - Ki));
    if (K >= Ki) omega = 1.0 - (Ki / K) * ex;
    const double d = Ki / (K * K);
    domega = d * ex + d * K * ex / (Kc - Ki);
  } else if (gsoft == 5) {                       // power (:1389-1413, :1446-1472), Kcc = Kc
    const double a = Ki / K, b = 1.0 / (Kc - Ki), c = (Kc - K) * b;
    if (K >= Ki) {
      const double omegac = 1.0 - pow(Ki / Kc, P.sB) * pow((Kc - Kc) / (Kc - Ki), P.sA);
      omega = (K > Kc) ? omegac : 1.0 - pow(a, P.sB) * pow((Kc - K) / (Kc - Ki), P.sA);
    }
    domega = (K > Kc) ? 0.0 : pow(a, P.sB - 1.0) * pow(c, P.sA - 1.0) * ((a / K) * P.sB * c + a * P.sA * b);
  } else if (gsoft == 6) {                       // hyperbolic (:1350-1380), sA = b; isTiny as in the checker
    const double q = 1.0 + P.sA * (K - Ki);
    if (K >= Ki) { omega = 1.0 - 1.0 / q; if (fabs(1.0 - omega) < 2.2250738585072014e-308) omega = 0.999; }
    domega = P.sA / q / q;
  } else if (gsoft == 7) {                       // exponential2Reg: not in the overloads GradientDamage reaches
  } else if (gsoft == 8) {                       // expoEnergy (:1495-1523), sA = s
    if (K >= Ki) {
      const double t = (K - Ki) / Kc, tem1 = pow(t, P.sA), tem2 = pow(t, P.sA - 1.0);
      omega = 1.0 - exp(-tem1);
      domega = exp(-tem1) * tem2 * P.sA / Kc;
    }
  } else if (gsoft == 9) {                       // polynomial (:1530-1574); d from the un-normalised a, verbatim
    if (K >= Ki) {
      const double an = (K - Ki) / (Kc - Ki);
      omega = 1.0 - (Ki / K) * (1.0 + (an * an) * (2.0 * an - 3.0));
      const double a = Ki - K, b = Kc - Ki, c = a / b, d = 1.0 + (a * a) * (2.0 * a - 3.0);
      domega = -Ki / (K * K) * (2.0 * K * (c / b * (2.0 * c - 3.0) + (c * c) / b) - d);
    }
  } else {                                        // multilinear (:1581-1633), sA = kappaP
    const double Kp = P.sA;
    if (K >= Ki) {
      if (K > Kp) { omega = 1.0 - Ki / K; domega = Ki / (K * K); }
      else { omega = (K < Kp) ? 0.0 : (K > Kc ? 0.999 : (K - Kp) / (Kc - Kp)); domega = (K < Kc) ? 1.0 / (Kc - Kp) : 0.0; }
    }
  }
}

// von Mises equivalent strain (DamageMaterial.cpp:609-617, :667-684; VonMisesStrain.cpp:49-116; J2 and its gradient with
// respect to the tensor components, Invariants.cpp:111-118, :188-202).  3-D only.
template <int S>
__device__ __forceinline__ void vonmises_strain(const Params& P, const double (&eps)[S], double& eq, double (&deq)[S]) {
  if constexpr (S == 6) {
    const double v0 = eps[0], v1 = eps[1], v2 = eps[2], v3 = 0.5 * eps[3], v4 = 0.5 * eps[4], v5 = 0.5 * eps[5];
    const double I1 = v0 + v1 + v2;
    const double J2 = (1.0 / 3.0) * (v0 * v0 + v1 * v1 + v2 * v2 - v0 * v1 - v1 * v2 - v2 * v0) + v3 * v3 + v4 * v4 + v5 * v5;
    double d = sqrt(P.vmA * P.vmA * I1 * I1 + P.vmC * J2);
    eq = P.vmB * (P.vmA * I1 + d);
    if (d == 0.0) d = 1.0;
    const double e = P.vmB / d, dI1 = P.vmA * P.vmB + e * P.vmA * P.vmA * I1, dJ2 = e * P.vmC * 0.5;
    deq[0] = dI1 + dJ2 * ((1.0 / 3.0) * (2.0 * v0 - v1 - v2));
    deq[1] = dI1 + dJ2 * ((1.0 / 3.0) * (2.0 * v1 - v2 - v0));
    deq[2] = dI1 + dJ2 * ((1.0 / 3.0) * (2.0 * v2 - v0 - v1));
    deq[3] = dJ2 * (2.0 * v3); deq[4] = dJ2 * (2.0 * v4); deq[5] = dJ2 * (2.0 * v5);
  }
}

// Mazars equivalent strain and its derivative as DamageMaterial computes them (DamageMaterial.cpp:618-642, :886-1032):
// principal strains from the trigonometric solution of the characteristic cubic (Invariants.cpp:209-224,
// MathUtils.cpp:150-181) with the reference's PI = 3.14159265 (Constants.cpp:13).  eps: xx, yy, gamma_xy in 2-D (zz = 0).
template <int RANK, int S>
__device__ __forceinline__ void mazars_strain(const Params& P, const double (&eps)[S], double& eq, double (&deq)[S]) {
  double v[6];
  if constexpr (RANK == 2) { v[0] = eps[0]; v[1] = eps[1]; v[2] = 0.0; v[3] = 0.5 * eps[2]; v[4] = 0.0; v[5] = 0.0; }
  else { v[0] = eps[0]; v[1] = eps[1]; v[2] = eps[2]; v[3] = 0.5 * eps[3]; v[4] = 0.5 * eps[4]; v[5] = 0.5 * eps[5]; }
  const double I1 = v[0] + v[1] + v[2];
  const double I2 = v[0] * v[1] + v[1] * v[2] + v[2] * v[0] - (v[3] * v[3] + v[4] * v[4] + v[5] * v[5]);
  const double I3 = v[0] * v[1] * v[2] + 2.0 * v[3] * v[4] * v[5] - v[0] * v[4] * v[4] - v[1] * v[5] * v[5] - v[2] * v[3] * v[3];
  double pr[3];
  {
    const double RAD_120 = 0.6666666666666667 * 3.14159265;
    const double aa = -I1, bb = I2, cc = -I3, aa3 = aa / 3.0;
    const double q = (aa * aa - 3.0 * bb) / 9.0;
    const double rr = (2.0 * aa * aa * aa - 9.0 * aa * bb + 27.0 * cc) / 54.0;
    double help = rr / sqrt(q * q * q);
    if (fabs(help) > 1.0) help = (help < 0) ? -1.0 : 1.0;
    const double phi = acos(help), p = sqrt(q);
    pr[0] = -2.0 * p * cos(phi / 3.0) - aa3;
    pr[1] = -2.0 * p * cos(phi / 3.0 + RAD_120) - aa3;
    pr[2] = -2.0 * p * cos(phi / 3.0 - RAD_120) - aa3;
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++) if (pr[i] > 0) s += pr[i] * pr[i];
  eq = sqrt(s);
#pragma unroll
  for (int i = 0; i < S; i++) deq[i] = 0.0;
  if constexpr (RANK == 3) {
    if (eq < 1.e-16) return;
    const double dI2[6] = { v[1] + v[2], v[2] + v[0], v[0] + v[1], -2. * v[3], -2. * v[4], -2. * v[5] };
    const double dI3[6] = { v[1] * v[2] - v[4] * v[4], v[2] * v[0] - v[5] * v[5], v[0] * v[1] - v[3] * v[3],
                            2.0 * (v[4] * v[5] - v[2] * v[3]), 2.0 * (v[5] * v[3] - v[0] * v[4]), 2.0 * (v[3] * v[4] - v[1] * v[5]) };
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const double vi = pr[i];
      if (vi > 0) {
        const double temp = 1.0 / (3.0 * vi * vi - 2.0 * I1 * vi + I2);
#pragma unroll
        for (int k = 0; k < 6; k++) deq[k] += vi * (temp * (vi * vi * (k < 3 ? 1.0 : 0.0) - vi * dI2[k] + dI3[k]));
      }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) deq[k] *= (1.0 / eq);
  } else {
    const double exx = eps[0], eyy = eps[1], exy = 0.5 * eps[2];
    const double exxMeyy = exx - eyy, exxPeyy = exx + eyy;
    double d = sqrt(exxMeyy * exxMeyy + 4.0 * exy * exy);
    double p0 = 0.5 * (exxPeyy + d), p1 = exxPeyy - p0, p2 = P.pstress ? P.poiD * exxPeyy : 0.0;
    p0 = p0 < 0. ? 0. : p0; p1 = p1 < 0. ? 0. : p1; p2 = p2 < 0. ? 0. : p2;
    const double den = sqrt(p0 * p0 + p1 * p1 + p2 * p2);
    if (den < 1e-08) return;
    const double denInv = 1. / den;
    const double dedprin0 = p0 * denInv, dedprin1 = p1 * denInv, dedprin2 = p2 * denInv;
    if (d == 0.0) d = 1.0;
    const double fac = 0.5 / d;
    const double de1dexx = 0.5 + fac * exxMeyy, de1deyy = 1.0 - de1dexx, de1dexy = 2.0 * fac * exy;
    deq[0] = dedprin0 * de1dexx + dedprin1 * de1deyy + dedprin2 * P.poiD;
    deq[1] = dedprin0 * de1deyy + dedprin1 * de1dexx + dedprin2 * P.poiD;
    deq[2] = dedprin0 * de1dexy + dedprin1 * (-de1dexy);
  }
}

// Constitutive part of one Gauss point: degradation, material update, history, phase-field coefficients
// (PhaseFractureModel.cpp:656-703,714,728; Ext variants PhaseFractureExtModel.cpp:656-758).
// pf / pf0 / pfX: phase field, old phase field and state00 (Ext) or stateExt_ (ExtIt) interpolated to the point.
template <class SH, int FAM, bool KEEP, class SP = SpecRuntime>
__device__ __forceinline__ void ip_material(const Params& P, int64_t ipoint, const double (&eps)[SH::S], double pf,
                                            double pf0, double pfX, double H0, bool write, IpMat<SH, FAM, KEEP>& r,
                                            double& err_acc) {
  constexpr int RANK = SH::RANK, S = SH::S, STR = SH::STR;
  constexpr bool PF = (FAM != FAM_LE);
  const int srows = PF ? STR + 1 : STR;
  double sp[S], Psi = 0.0;
  r.c1 = 0.0; r.cf = 0.0; r.dg = 0.0; r.dgl = 0.0;
  if constexpr (!PF) {
    iso_update<RANK, S, SP>(P, eps, 1.0, r.d0, r.d1, r.d3, r.sig, sp, Psi);
  } else {
    if constexpr (FAM == FAM_PF_MIEHE) {
      if (SP::model(P) >= 4 && SP::model(P) <= 6) {     // MicroPhaseFractureModel.cpp:656-830 (general-tangent family only)
        MicroOut mo;
        double gphi = 1.0, dhist = 0.0;
        const bool ext = SP::model(P) != 4;               // MicroPhaseFractureExtModel.cpp:656-940, ...ExtItModel.cpp:712
        double pfEx = pfX;
        if (ext) {
          if (SP::model(P) == 6 && write) { const double dd = pf - pfEx; err_acc += dd * dd; }
          if (pfEx < 0.0) pfEx = 0.0; else if (pfEx > 1.0) pfEx = 1.0;
        }
        auto gfun = [&](double Psi_raw, const double (&)[S]) {
          const double PsiM = (Psi_raw > P.psi0) ? Psi_raw : P.psi0;
          if (ext) {
            gphi = micro_first_solve(P, PsiM, pfEx, H0, dhist);
            micro_newton(P, PsiM, pf, dhist, H0, mo);
          } else {
            gphi = micro_newton(P, PsiM, pf, H0, H0, mo);
            dhist = mo.d;
          }
          return gphi;
        };
        if (SP::material(P) == 3) {
          miehe_update_f<RANK, S>(P, eps, gfun, r.D, r.sig, sp, Psi);
        } else if constexpr (SP::ma) {
          // AmorPhase, split update (AmorPhaseMaterial.cpp:239-271), in instantiations of its own (SpecMA: inside the
          // runtime-model kernels it cost the Quad8 tile kernel its second workgroup per CU): stiffP = 2G PDev (+ K PVol in
          // tension); stiffN = K PVol is written at compression points only (:256-263) and the model's array is never reset
          // (MicroPhaseFractureModel.cpp:535), so a point in tension finds K PVol once the serial loop has passed a
          // compression point (first_comp_kernel) and the array's initial contents -- zero assumed, as in the checker -- before.
          const double tr = (RANK == 3) ? (eps[0] + eps[1] + eps[RANK - 1]) : (eps[0] + eps[1] + 0.0);
          // The one call-wide decision -- which point is the first in compression -- is the pre-pass's (first_comp_kernel), and
          // this kernel's own tr, whose last bits differ from path to path (unrolled sums, sym(F J^-1), ...), is overruled
          // where it could contradict it: every point before the first is in tension, the first one is in compression.
          // Behind it the order no longer matters (kn = K either way) and the point's own sign decides its own branch, as in
          // the plain AmorPhase kernels.
          const unsigned long long first = *P.first_comp;
          // FA_MICRO_AMOR_MASK=1 (P.comp_mask): the pre-pass classifies every point, so the kernel paths also agree with each
          // other on points whose tr is rounding noise (where the tangent jumps by O(K) and no two summation orders need
          // agree -- as in the plain AmorPhase kernels): a full sweep of the pre-pass and one more load per point, cfg3
          // mesh 5.2 -> 6.8 ms, hence opt-in.
          const bool comp = P.comp_mask ? (((P.comp_mask[ipoint / SH::NIP] >> (int)(ipoint % SH::NIP)) & 1u) != 0)
                          : (((unsigned long long)ipoint < first) ? false : (((unsigned long long)ipoint == first) ? true : (tr <= 0.0)));
          double dd = 2.0 * P.G * (2.0 / 3.0), od = 2.0 * P.G * (-1.0 / 3.0);
          const double sh = 2.0 * P.G * 0.5;
          if (!comp) { dd += P.Kb; od += P.Kb; }
          double psi2 = 0.0;
#pragma unroll
          for (int i = 0; i < S; i++) {
            double acc = 0.0;
            if (i < RANK) {
#pragma unroll
              for (int j = 0; j < RANK; j++) acc += ((i == j) ? dd : od) * eps[j];
            } else acc = sh * eps[i];
            sp[i] = acc; psi2 += acc * eps[i];
          }
          Psi = 0.5 * psi2;
          const double g = gfun(Psi, sp);
          const double kn = (comp || (unsigned long long)ipoint > first) ? P.Kb : 0.0;
          const double sn = kn * tr;
#pragma unroll
          for (int i = 0; i < S; i++) {
            r.sig[i] = g * sp[i] + ((i < RANK) ? sn : 0.0);
#pragma unroll
            for (int j = 0; j < S; j++)
              r.D[i][j] = (i < RANK && j < RANK) ? g * ((i == j) ? dd : od) + kn : ((i == j) ? g * sh : 0.0);
          }
        } else {                                        // BourdinPhase: stiffP = D, stiffN = 0 (BourdinPhaseMaterial.cpp:148-166)
          double psi2 = 0.0;
#pragma unroll
          for (int i = 0; i < S; i++) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < S; j++) {
              const double dij = (i < RANK && j < RANK) ? ((i == j) ? P.hd0 : P.hd1) : ((i == j) ? P.hd3 : 0.0);
              acc += dij * eps[j];
            }
            sp[i] = acc; psi2 += acc * eps[i];
          }
          Psi = 0.5 * psi2;
          const double g = gfun(Psi, sp);
#pragma unroll
          for (int i = 0; i < S; i++) {
            r.sig[i] = g * sp[i];
#pragma unroll
            for (int j = 0; j < S; j++)
              r.D[i][j] = g * ((i < RANK && j < RANK) ? ((i == j) ? P.hd0 : P.hd1) : ((i == j) ? P.hd3 : 0.0));
          }
        }
        r.d0 = r.d1 = r.d3 = 0.0;
        // `d > 1` branch: cdS = 0 here, a stale dphidStrain in the reference -- counted, so that the divergence is never
        // silent (fa_micro_stale_hits; reachable only with a committed d0 > 1)
        if (mo.stale && write && P.stale_hits != nullptr) atomicAdd(P.stale_hits, 1ull);
        if (!ext) {
          const double coef = mo.dgphi * mo.cdS;        // second elemMat1 term: dgphi * sigma+ (x) dphidStrain (:816)
#pragma unroll
          for (int i = 0; i < S; i++)
#pragma unroll
            for (int j = 0; j < S; j++) r.D[i][j] += coef * (sp[i] * sp[j]);
        }
        if (P.strain != nullptr && write) P.strain[ipoint * srows + STR] = gphi;
        if (write) P.hist_new[ipoint] = dhist;
        r.c1 = P.alpha * (1.0 - mo.dphidd);             // K4 (:817)
        r.cf = P.alpha * (pf - mo.d);                   // f2 (:830)
        r.dg = ext ? 0.0 : mo.dgphi * mo.dphidd;        // K2 (:822); the Ext variants have no elemMat2
        r.dgl = -P.alpha * mo.cdS;                      // K3 (:823)
        if constexpr (KEEP) {
#pragma unroll
          for (int c = 0; c < S; c++) r.sp[c] = sp[c];
        }
        return;
      }
    }
    if constexpr (FAM == FAM_PF_ISO && SP::gd) {
      {
        // GradientDamageModel.cpp:500-560 with DamageMaterial::update (DamageMaterial.cpp:356-440), eqvStrain = Energy:
        // pf is the nonlocal equivalent strain at the point (strain_(strCount, ipoint), :524), H0 the committed kappa
        // with the loading flag in its sign bit.
        double e0v[S];
        double psi2 = 0.0;                                // eps^t D0 eps; D0 eps = dEpsBardEps (:700-705) = stiff0 eps (:547)
#pragma unroll
        for (int i = 0; i < S; i++) {
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < S; j++) {
            const double dij = (i < RANK && j < RANK) ? ((i == j) ? P.hd0 : P.hd1) : ((i == j) ? P.hd3 : 0.0);
            acc += dij * eps[j];
          }
          e0v[i] = acc; psi2 += acc * eps[i];
        }
        double eq = 0.5 * psi2;                           // getEquiStrain, StrainEnergy (:645-650)
        double deq[S];                                    // getdEpsBardEps
#pragma unroll
        for (int i = 0; i < S; i++) deq[i] = e0v[i];
        if (SP::eqv(P) == 1) mazars_strain<RANK, S>(P, eps, eq, deq);
        else if (SP::eqv(P) == 2) vonmises_strain<S>(P, eps, eq, deq);
        const double kappa0 = fabs(H0);
        const int load0 = (H0 < 0.0) ? 1 : 0;
        const double fl = pf - kappa0;                    // loading function (:383)
        double kappa; int load;
        if (fl == 0.0)     { kappa = kappa0; load = load0; }
        else if (fl > 0.0) { kappa = pf; load = (kappa < P.kI) ? 0 : 1; }
        else               { kappa = kappa0; load = 0; }
        double omega, domega;
        damage_law(P, kappa, omega, domega, SP::soft(P));
        const double s1 = 1.0 - omega;
        r.d0 = s1 * P.hd0; r.d1 = s1 * P.hd1; r.d3 = s1 * P.hd3;      // stiff = (1 - omega) elasticMod_ (:421)
#pragma unroll
        for (int i = 0; i < S; i++) r.sig[i] = s1 * e0v[i];
        if (P.strain != nullptr && write) P.strain[ipoint * srows + STR] = pf;
        if (write) P.hist_new[ipoint] = load ? -kappa : kappa;
        r.c1 = 1.0;                                       // K4 = w (N N^t + c Be^t Be) (:563)
        r.cf = pf - eq;                                   // f2 = -w eq N + K4 eqv (:570-575)
        r.dg = load ? -domega : 0.0;                      // K2 = -w dOmega Bd^t stiff0 eps N^t when loading (:551-558)
        // Mazars: sp carries d eq / d eps for K3, so K2 takes stiff0 eps from the stress, sig = (1 - omega) stiff0 eps
        if (P.k2sig) r.dg = (load && s1 != 0.0) ? -domega / s1 : 0.0;
        r.dgl = -1.0;                                     // K3 = -w N dEpsBardEps^t Bd (:545-547)
        if constexpr (KEEP) {
#pragma unroll
          for (int c = 0; c < S; c++) r.sp[c] = deq[c];
        }
        return;
      }
    }
    double dgphi, ddgphi, loading = 1.0;
    double gmat;
    if (SP::model(P) == 1) {
      degrade<SP>(P, pf, gmat, dgphi, ddgphi);
    } else {
      // extrapolated phase field (PhaseFractureExtModel.cpp:656-691, PhaseFractureExtItModel.cpp:693)
      double pfEx;
      if (SP::model(P) == 2) {
        if (P.dt0 < 1.e-15) pfEx = pf0;
        else                pfEx = -pfX * P.dt / P.dt0 + pf0 * (P.dt + P.dt0) / P.dt0;
      } else {
        pfEx = pfX;
        if (write) { const double d = pf - pfEx; err_acc += d * d; }
      }
      if (pfEx < 0.0) pfEx = 0.0; else if (pfEx > 1.0) pfEx = 1.0;
      double t1, t2, t3;
      degrade<SP>(P, pfEx, gmat, t1, t2);
      degrade<SP>(P, pf, t3, dgphi, ddgphi);
    }
    if (P.strain != nullptr && write) P.strain[ipoint * srows + STR] = gmat;
    if constexpr (FAM == FAM_PF_MIEHE) { r.d0 = r.d1 = r.d3 = 0.0; miehe_update<RANK, S>(P, eps, gmat, r.D, r.sig, sp, Psi); }
    else iso_update<RANK, S, SP>(P, eps, gmat, r.d0, r.d1, r.d3, r.sig, sp, Psi);
    if (SP::model(P) == 1) {                                 // PhaseFractureModel.cpp:696-703
      if (Psi < H0) loading = 0.0;
      Psi = (Psi > H0) ? Psi : H0;
      if (write) P.hist_new[ipoint] = Psi;
    } else {
      Psi = (Psi > P.psi0) ? Psi : P.psi0;              // PhaseFractureExtModel.cpp:726
    }
    const double dw = P.eta + 2.0 * (1.0 - P.eta) * pf, ddw = 2.0 * (1.0 - P.eta);
    r.c1 = P.gcl * ddw + ddgphi * Psi;                  // K4 (:714)
    r.cf = P.gcl * dw + dgphi * Psi;                    // f2 (:728)
    if ((SP::model(P) == 2 || SP::model(P) == 3) && (pf0 - pf > 1.e-15)) {        // penalty, PhaseFractureExtModel.cpp:752-758
      r.c1 += P.gcl * P.beta;
      r.cf -= P.gcl * P.beta * (pf0 - pf);
    }
    r.dg = dgphi; r.dgl = loading * dgphi;
  }
  if constexpr (KEEP) {
#pragma unroll
    for (int c = 0; c < S; c++) r.sp[c] = sp[c];
  }
}

// One Gauss point of element e: IShape::getShapeGradients, strain = B u, degradation, material update,
// history (PhaseFractureModel.cpp:629-703; Ext variants PhaseFractureExtModel.cpp:656-758).
// `ip` may differ from lane to lane.  `write` = this lane is the one that stores the per-point outputs
// (history, strain cache, errExt_).
// ip_point_core takes the point's reference-element data (dN/dxi, N, weight) as values, so that a caller can pick
// them with selects from scalar loads; ip_point reads them from P.tab at index ip.
template <class SH, int FAM, bool WANT_K, bool KEEP, class SP = SpecRuntime>
__device__ __forceinline__ void ip_point_core(const Params& P, int64_t e, int ip, const double (&dn)[SH::NPE][SH::RANK],
                                              const double (&Nv)[SH::NPE], double wt,
                                              const ElemData<SH, FAM != FAM_LE>& E,
                                              double H0, bool write, IpRec<SH, FAM, KEEP>& r, double& err_acc) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, NIP = SH::NIP, S = SH::S, STR = SH::STR;
  constexpr bool PF = (FAM != FAM_LE);

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
      for (int k = 0; k < RANK; k++) J[i][k] += E.x[a][i] * dn[a][k];
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
  r.w = wt * det;
#pragma unroll
  for (int a = 0; a < NPE; a++)
#pragma unroll
    for (int i = 0; i < RANK; i++) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < RANK; k++) s += dn[a][k] * inv[k][i];
      r.g[a][i] = s;
    }
  // ---- strain = B u (ShapeUtils.cpp:73-145); 2D keeps xx, yy, gamma_xy ----
  double eps[S];
#pragma unroll
  for (int c = 0; c < S; c++) eps[c] = 0.0;
#pragma unroll
  for (int a = 0; a < NPE; a++) {
    if constexpr (RANK == 2) {
      eps[0] += r.g[a][0] * E.u[a][0];
      eps[1] += r.g[a][1] * E.u[a][1];
      eps[2] += r.g[a][1] * E.u[a][0]; eps[2] += r.g[a][0] * E.u[a][1];
    } else {
      eps[0] += r.g[a][0] * E.u[a][0];
      eps[1] += r.g[a][1] * E.u[a][1];
      eps[2] += r.g[a][2] * E.u[a][2];
      eps[3] += r.g[a][1] * E.u[a][0]; eps[3] += r.g[a][0] * E.u[a][1];
      eps[4] += r.g[a][2] * E.u[a][1]; eps[4] += r.g[a][1] * E.u[a][2];
      eps[5] += r.g[a][0] * E.u[a][2]; eps[5] += r.g[a][2] * E.u[a][0];
    }
  }
  const int64_t ipoint = e * NIP + ip;               // initializeIPMPMap_, PhaseFractureModel.cpp:1070-1093
  const int srows = PF ? STR + 1 : STR;
  if (P.strain != nullptr && write) {
    double* sc = P.strain + ipoint * srows;
    if constexpr (RANK == 2) { sc[0] = eps[0]; sc[1] = eps[1]; sc[2] = 0.0; sc[3] = eps[2]; }
    else {
#pragma unroll
      for (int c = 0; c < S; c++) sc[c] = eps[c];
    }
  }

  double pf = 0.0, pf0 = 0.0, pfX = 0.0;
  if constexpr (PF) {
#pragma unroll
    for (int i = 0; i < RANK; i++) r.gphi[i] = 0.0;
#pragma unroll
    for (int a = 0; a < NPE; a++) {
      pf += Nv[a] * E.phi[a];
#pragma unroll
      for (int i = 0; i < RANK; i++) r.gphi[i] += r.g[a][i] * E.phi[a];
    }
    if (SP::model(P) == 2 || SP::model(P) == 3) {
#pragma unroll
      for (int a = 0; a < NPE; a++) { pf0 += Nv[a] * E.phi0[a]; pfX += Nv[a] * E.phiX[a]; }
    }
    if ((SP::model(P) == 5 || SP::model(P) == 6)) {
#pragma unroll
      for (int a = 0; a < NPE; a++) pfX += Nv[a] * E.phiX[a];
    }
  }
  ip_material<SH, FAM, KEEP, SP>(P, ipoint, eps, pf, pf0, pfX, H0, write, r, err_acc);
}

template <class SH, int FAM, bool WANT_K, bool KEEP, class SP = SpecRuntime>
__device__ __forceinline__ void ip_point(const Params& P, int64_t e, int ip, const ElemData<SH, FAM != FAM_LE>& E,
                                         double H0, bool write, IpRec<SH, FAM, KEEP>& r, double& err_acc) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, NIP = SH::NIP;
  const const_dptr tabN = (const_dptr)P.tab;
  const const_dptr tabD = tabN + NIP * NPE;
  const const_dptr tabW = tabN + NIP * NPE * (1 + RANK);
  double dn[NPE][RANK], Nv[NPE];
#pragma unroll
  for (int a = 0; a < NPE; a++) {
    Nv[a] = tabN[ip * NPE + a];
#pragma unroll
    for (int k = 0; k < RANK; k++) dn[a][k] = tabD[(ip * NPE + a) * RANK + k];
  }
  ip_point_core<SH, FAM, WANT_K, KEEP, SP>(P, e, ip, dn, Nv, tabW[ip], E, H0, write, r, err_acc);
}

// Rows of local node la from one Gauss-point record.  Nip[b] = N_b(ip).
template <class SH, int FAM, bool WANT_K, bool KEEP>
__device__ __forceinline__ void accum_rows(const Params& P, const IpRec<SH, FAM, KEEP>& r, int la,
                                           const double (&Nip)[SH::NPE],
                                           RowBlock<SH, WANT_K, FAM != FAM_LE, KEEP>& R) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, S = SH::S;
  constexpr bool PF = (FAM != FAM_LE);
  double wga[RANK], Nla = 0.0;                        // w * grad N_la (select: no runtime register indexing)
#pragma unroll
  for (int i = 0; i < RANK; i++) wga[i] = 0.0;
#pragma unroll
  for (int a = 0; a < NPE; a++)
    if (a == la) {
#pragma unroll
      for (int i = 0; i < RANK; i++) wga[i] = r.w * r.g[a][i];
      Nla = Nip[a];
    }
  {
    double t[RANK];
    Bt_vec<RANK, S>(wga, r.sig, t);                   // f1 += w B^T sigma (:727)
#pragma unroll
    for (int i = 0; i < RANK; i++) R.fu[i] += t[i];
  }
  if constexpr (WANT_K && FAM == FAM_PF_MIEHE) {       // K1 += w B_a^T D B_b with a full symmetric D
    double A[RANK][S];                                // A = w B_a^T D
#pragma unroll
    for (int c = 0; c < S; c++) {
      double col[S], t[RANK];
#pragma unroll
      for (int k = 0; k < S; k++) col[k] = r.D[k][c];
      Bt_vec<RANK, S>(wga, col, t);
#pragma unroll
      for (int i = 0; i < RANK; i++) A[i][c] = t[i];
    }
#pragma unroll
    for (int b = 0; b < NPE; b++) {
      double gb[RANK];
#pragma unroll
      for (int i = 0; i < RANK; i++) gb[i] = r.g[b][i];
#pragma unroll
      for (int i = 0; i < RANK; i++) {
        double t[RANK];
        Bt_vec<RANK, S>(gb, A[i], t);
#pragma unroll
        for (int j = 0; j < RANK; j++) R.Kuu[i][b * RANK + j] += t[j];
      }
    }
  } else if constexpr (WANT_K) {
    if constexpr (RANK == 2) {                        // K1 += w B_a^T D B_b (:713)
      const double a00 = wga[0] * r.d0, a01 = wga[0] * r.d1, a03 = wga[0] * r.d3;
      const double a10 = wga[1] * r.d0, a11 = wga[1] * r.d1, a13 = wga[1] * r.d3;
#pragma unroll
      for (int b = 0; b < NPE; b++) {
        const double gbx = r.g[b][0], gby = r.g[b][1];
        R.Kuu[0][2 * b]     += a00 * gbx + a13 * gby;
        R.Kuu[0][2 * b + 1] += a01 * gby + a13 * gbx;
        R.Kuu[1][2 * b]     += a11 * gbx + a03 * gby;
        R.Kuu[1][2 * b + 1] += a10 * gby + a03 * gbx;
      }
    } else {
#pragma unroll
      for (int b = 0; b < NPE; b++) {
        const double dotp = wga[0] * r.g[b][0] + wga[1] * r.g[b][1] + wga[2] * r.g[b][2];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) {
            if (i == j) R.Kuu[i][3 * b + j] += (r.d0 - r.d3) * (wga[i] * r.g[b][i]) + r.d3 * dotp;
            else        R.Kuu[i][3 * b + j] += r.d1 * (wga[i] * r.g[b][j]) + r.d3 * (wga[j] * r.g[b][i]);
          }
      }
    }
  }
  if constexpr (PF) {
    const double wNa = r.w * Nla;
    double gg = 0.0;
#pragma unroll
    for (int i = 0; i < RANK; i++) gg += wga[i] * r.gphi[i];
    R.fp += wNa * r.cf + P.c2 * gg;                                   // f2 (:728-730)
    if constexpr (WANT_K) {
      const double wc1 = wNa * r.c1;
      double c2w[RANK];
#pragma unroll
      for (int i = 0; i < RANK; i++) c2w[i] = P.c2 * wga[i];
#pragma unroll
      for (int b = 0; b < NPE; b++) {                                 // K4 (:714-716): w c1 N N^T + w c2 grad N . grad N
        double t = R.Kpp[b] + wc1 * Nip[b];
#pragma unroll
        for (int i = 0; i < RANK; i++) t += c2w[i] * r.g[b][i];
        R.Kpp[b] = t;
      }
      if constexpr (KEEP) {
        if (P.keep2) {                                                // K2 += w g' B^T (sigma+ (x) N) (:720)
          double t[RANK];
          if (P.k2sig) Bt_vec<RANK, S>(wga, r.sig, t);                 // GradientDamage / Mazars: D0 eps travels as sig / (1 - omega)
          else Bt_vec<RANK, S>(wga, r.sp, t);
#pragma unroll
          for (int i = 0; i < RANK; i++)
#pragma unroll
            for (int b = 0; b < NPE; b++) R.Kup[i][b] += (r.dg * t[i]) * Nip[b];
        }
        if (P.keep3) {                                                // K3 += w loading g' N (x) (sigma+^T B) (:721)
          const double c = wNa * r.dgl;
#pragma unroll
          for (int b = 0; b < NPE; b++) {
            double t[RANK], gb[RANK];
#pragma unroll
            for (int i = 0; i < RANK; i++) gb[i] = r.g[b][i];
            Bt_vec<RANK, S>(gb, r.sp, t);
#pragma unroll
            for (int j = 0; j < RANK; j++) R.Kpu[b * RANK + j] += c * t[j];
          }
        }
      }
    }
  }
}

// Gauss-point loop of element e done by ONE lane, keeping the rows of local node la.
template <class SH, int FAM, bool WANT_K, bool KEEP, class SP = SpecRuntime>
__device__ __forceinline__ void compute_rowblock(const Params& P, int64_t e, int la, const int32_t (&cn)[SH::NPE],
                                                 bool owned, RowBlock<SH, WANT_K, FAM != FAM_LE, KEEP>& R,
                                                 double& err_acc) {
  constexpr int NPE = SH::NPE, NIP = SH::NIP;
  constexpr bool PF = (FAM != FAM_LE);
  double H0v[NIP];                                   // preHist_.H0_ of e: one contiguous read
#pragma unroll
  for (int ip = 0; ip < NIP; ip++) H0v[ip] = 0.0;
  if constexpr (PF) {
    if (SP::model(P) == 1 || SP::model(P) >= 4) {
#pragma unroll
      for (int ip = 0; ip < NIP; ip++) H0v[ip] = P.hist_pre[e * NIP + ip];
    }
  }
  ElemData<SH, PF> E;
  gather_elem<SH, PF, SP>(P, cn, E);
  R.zero();
#pragma unroll 1
  for (int ip = 0; ip < NIP; ip++) {
    IpRec<SH, FAM, KEEP> r;
    double H0 = 0.0;
#pragma unroll
    for (int k = 0; k < NIP; k++) if (k == ip) H0 = H0v[k];
    ip_point<SH, FAM, WANT_K, KEEP, SP>(P, e, ip, E, H0, owned && ((ip % NPE) == la), r, err_acc);
    double Nip[NPE];
#pragma unroll
    for (int b = 0; b < NPE; b++) Nip[b] = ((const_dptr)P.tab)[ip * NPE + b];
    accum_rows<SH, FAM, WANT_K, KEEP>(P, r, la, Nip, R);
  }
}

// ---- group broadcast: value of lane J of the NPE-lane group ----
template <int NPE, int J>
__device__ __forceinline__ double gbcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (NPE == 4) {                          // DPP quad_perm [J,J,J,J]
    lo = __builtin_amdgcn_update_dpp(0, lo, J * 0x55, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, J * 0x55, 0xF, 0xF, false);
  } else {
    const int src = (((int)threadIdx.x & 63) & ~(NPE - 1)) + J;
    lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
    hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
  }
  return __hiloint2double(hi, lo);
}

template <class SH, int FAM, bool KEEP, int J>
__device__ __forceinline__ void bcast_rec(const IpRec<SH, FAM, KEEP>& m, IpRec<SH, FAM, KEEP>& r) {
  constexpr int NPE = SH::NPE;
  constexpr bool PF = (FAM != FAM_LE);
  static_assert(FAM != FAM_PF_MIEHE, "patch kernel carries isotropic-form tangents only");
  r.w = gbcast<NPE, J>(m.w);
#pragma unroll
  for (int a = 0; a < NPE; a++)
#pragma unroll
    for (int i = 0; i < SH::RANK; i++) r.g[a][i] = gbcast<NPE, J>(m.g[a][i]);
  r.d0 = gbcast<NPE, J>(m.d0); r.d1 = gbcast<NPE, J>(m.d1); r.d3 = gbcast<NPE, J>(m.d3);
#pragma unroll
  for (int c = 0; c < SH::S; c++) r.sig[c] = gbcast<NPE, J>(m.sig[c]);
  if constexpr (KEEP) {
#pragma unroll
    for (int c = 0; c < SH::S; c++) r.sp[c] = gbcast<NPE, J>(m.sp[c]);
    r.dg = gbcast<NPE, J>(m.dg); r.dgl = gbcast<NPE, J>(m.dgl);
  }
  if constexpr (PF) {
    r.c1 = gbcast<NPE, J>(m.c1); r.cf = gbcast<NPE, J>(m.cf);
#pragma unroll
    for (int i = 0; i < SH::RANK; i++) r.gphi[i] = gbcast<NPE, J>(m.gphi[i]);
  }
}

// Gauss-point loop of element e shared by its NPE lanes (NIP == NPE): lane la evaluates point la (phase 1),
// then every lane accumulates the rows of its node from the NPE broadcast records (phase 2).
// Every lane of the group must be active (DPP / bpermute read the other lanes' registers).
template <class SH, int FAM, bool WANT_K, bool KEEP>
__device__ __forceinline__ void shared_phase2(const Params& P, int la, const IpRec<SH, FAM, KEEP>& mine,
                                              RowBlock<SH, WANT_K, FAM != FAM_LE, KEEP>& R) {
  constexpr int NPE = SH::NPE, NIP = SH::NIP;
  constexpr bool PF = (FAM != FAM_LE);
  static_assert(NIP == NPE, "shared evaluation needs one Gauss point per lane");
  R.zero();
  auto body = [&](auto jc) {
    constexpr int J = decltype(jc)::value;
    IpRec<SH, FAM, KEEP> r;
    bcast_rec<SH, FAM, KEEP, J>(mine, r);
    double Nip[NPE];
#pragma unroll
    for (int b = 0; b < NPE; b++) Nip[b] = P.tab[J * NPE + b];
    accum_rows<SH, FAM, WANT_K, KEEP>(P, r, la, Nip, R);
    __builtin_amdgcn_sched_barrier(0);               // keep the broadcast+accumulate rounds apart (registers)
  };
  SFor<0, NIP>::run(body);
}

// add one lane's row block into the LDS image of its node's CSR rows.  rs0 = LDS offset of the node's first
// row, nnb = number of neighbour nodes of the node (row i has nnb*cnt[i] entries), ps[b] = position of
// element node b in the node's neighbour list.
template <class SH, bool WANT_K, bool PF, bool KEEP, bool ATOMIC>
__device__ __forceinline__ void lds_add_rows(const Params& P, const RowBlock<SH, WANT_K, PF, KEEP>& R,
                                             double* sK, double* sF, int rs0, int nnb, int fo,
                                             const uint8_t (&ps)[SH::NPE]) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE;
  auto add = [](double& ref, double v) {
    if constexpr (ATOMIC) __hip_atomic_fetch_add(&ref, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else ref += v;
  };
#pragma unroll
  for (int i = 0; i < RANK; i++) add(sF[fo + i], R.fu[i]);
  if constexpr (PF) add(sF[fo + RANK], R.fp);
  if constexpr (WANT_K) {
    int rs = rs0;
#pragma unroll
    for (int i = 0; i < RANK; i++) {
#pragma unroll
      for (int b = 0; b < NPE; b++) {
        const int gs = rs + (int)ps[b] * P.cnt[i];
#pragma unroll
        for (int j = 0; j < RANK; j++) add(sK[gs + P.off[i][j]], R.Kuu[i][b * RANK + j]);
        if constexpr (KEEP) { if (P.keep2) add(sK[gs + P.off[i][RANK]], R.Kup[i][b]); }
      }
      rs += nnb * P.cnt[i];
    }
    if constexpr (PF) {
#pragma unroll
      for (int b = 0; b < NPE; b++) {
        const int gs = rs + (int)ps[b] * P.cnt[RANK];
        add(sK[gs + P.off[RANK][RANK]], R.Kpp[b]);
        if constexpr (KEEP) {
          if (P.keep3) {
#pragma unroll
            for (int j = 0; j < RANK; j++) add(sK[gs + P.off[RANK][j]], R.Kpu[b * RANK + j]);
          }
        }
      }
    }
  }
}

// Stream n doubles of an LDS image to HBM (coalesced, 4 independent LDS reads in flight per lane).
template <int BLK>
__device__ __forceinline__ void stream_out(double* __restrict__ g, const double* s, int n, bool overwrite) {
  int i = threadIdx.x;
  if (overwrite) {
    for (; i + 3 * BLK < n; i += 4 * BLK) {
      const double a0 = s[i], a1 = s[i + BLK], a2 = s[i + 2 * BLK], a3 = s[i + 3 * BLK];
      g[i] = a0; g[i + BLK] = a1; g[i + 2 * BLK] = a2; g[i + 3 * BLK] = a3;
    }
    for (; i < n; i += BLK) g[i] = s[i];
  } else {
    for (; i + 3 * BLK < n; i += 4 * BLK) {
      const double b0 = g[i], b1 = g[i + BLK], b2 = g[i + 2 * BLK], b3 = g[i + 3 * BLK];
      const double a0 = s[i], a1 = s[i + BLK], a2 = s[i + 2 * BLK], a3 = s[i + 3 * BLK];
      g[i] = b0 + a0; g[i + BLK] = b1 + a1; g[i + 2 * BLK] = b2 + a2; g[i + 3 * BLK] = b3 + a3;
    }
    for (; i < n; i += BLK) g[i] += s[i];
  }
}

// Same for the K image with 16-byte accesses: the image sits at s[0..n) with s - (base & 1) 16-byte aligned in LDS
// and g = K + base, so that element i of the image is 16-byte aligned in LDS exactly when it is in HBM
// (K itself 16-byte aligned, checked by the caller).  odd = base & 1.
#ifndef FA_NT_STORE
#define FA_NT_STORE 0     /* nontemporal stores of the K stream: no difference on cfg3 (4.503 against 4.515 ms) */
#endif
#ifndef FA_ACC_BATCH
#define FA_ACC_BATCH 10
#endif
typedef double d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store16(double2* q, const double2 v) {
#if FA_NT_STORE
  d2v t; t.x = v.x; t.y = v.y;
  __builtin_nontemporal_store(t, reinterpret_cast<d2v*>(q));      // K is written once and never read by this kernel
#else
  *q = v;
#endif
}
template <int BLK>
__device__ __forceinline__ void stream_out16(double* __restrict__ g, const double* s, int n, int odd, bool overwrite) {
  const int tid = threadIdx.x;
  if (odd && tid == 0 && n > 0) { if (overwrite) g[0] = s[0]; else g[0] += s[0]; }
  const int i0 = odd, npair = (n - i0) >> 1;
  double2* __restrict__ g2 = (double2*)(g + i0);
  const double2* s2 = (const double2*)(s + i0);
  int p = tid;
  if (overwrite) {
    for (; p + BLK < npair; p += 2 * BLK) { const double2 a0 = s2[p], a1 = s2[p + BLK]; store16(g2 + p, a0); store16(g2 + p + BLK, a1); }
    for (; p < npair; p += BLK) store16(g2 + p, s2[p]);
  } else {
    // accumulate: every read of K a lane makes for this tile is in flight before its first add -- FA_ACC_BATCH reads of
    // 16 bytes per lane cover a 40 KiB image with 256 lanes, one memory round trip per tile (one read at a time: 7.1 ms on
    // cfg3; four at a time, three round trips: 5.6 ms); the registers are free at this point of the kernel
    constexpr int NB = FA_ACC_BATCH;
    for (; p < npair; p += NB * BLK) {
      double2 b[NB];
#pragma unroll
      for (int k = 0; k < NB; k++) if (p + k * BLK < npair) b[k] = g2[p + k * BLK];
#pragma unroll
      for (int k = 0; k < NB; k++)
        if (p + k * BLK < npair) { const double2 a = s2[p + k * BLK]; b[k].x += a.x; b[k].y += a.y; g2[p + k * BLK] = b[k]; }
    }
  }
  const int last = i0 + 2 * npair;
  if (last < n && tid == BLK - 1) { if (overwrite) g[last] = s[last]; else g[last] += s[last]; }
}

// FA_ACCUMULATE as the lane-pair kernel does it: the LDS image STARTS as the caller's K (and f) instead of zero -- the
// reads are issued with the lane records, long before the first row block is added -- so the tile's contributions are
// added on top in LDS and the write-out is the plain store stream of the overwrite mode: no read-modify-write round
// trips at the end of the tile.  Same 16-byte alignment convention as stream_out16.
template <int BLK>
__device__ __forceinline__ void stream_in16(const double* __restrict__ g, double* s, int n, int odd) {
  const int tid = threadIdx.x;
  if (odd && tid == 0 && n > 0) s[0] = g[0];
  const int i0 = odd, npair = (n - i0) >> 1;
  const double2* __restrict__ g2 = (const double2*)(g + i0);
  double2* s2 = (double2*)(s + i0);
  int p = tid;
  for (; p + 3 * BLK < npair; p += 4 * BLK) {
    const double2 b0 = g2[p], b1 = g2[p + BLK], b2 = g2[p + 2 * BLK], b3 = g2[p + 3 * BLK];
    s2[p] = b0; s2[p + BLK] = b1; s2[p + 2 * BLK] = b2; s2[p + 3 * BLK] = b3;
  }
  for (; p < npair; p += BLK) s2[p] = g2[p];
  const int last = i0 + 2 * npair;
  if (last < n && tid == BLK - 1) s[last] = g[last];
}

template <int BLOCKSZ>
__device__ __forceinline__ void reduce_err(double err_acc, double* sErr, double* out) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) err_acc += __shfl_down(err_acc, o, 64);
  if ((threadIdx.x & 63) == 0) sErr[threadIdx.x >> 6] = err_acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int k = 0; k < (BLOCKSZ + 63) / 64; k++) s += sErr[k];
    *out = s;
  }
}

// ------------------------------------------------------------------------------------------------
// Patch kernel (default)
// ------------------------------------------------------------------------------------------------
// indices of one (patch element, lane): fetched one pass ahead of their use
template <int NPE>
struct LaneIdx {
  int64_t e; int own; int32_t cn[NPE]; uint8_t ps[NPE];
};

template <int NPE>
__device__ __forceinline__ void load_idx(const Params& P, int64_t recno, int la, bool valid, LaneIdx<NPE>& L) {
  constexpr int RS = prec_stride(NPE), OW = prec_own_words(NPE), PW = prec_pos_words(NPE);
  const int32_t* __restrict__ rec = P.prec + recno * RS;
  L.e = rec[0];
#pragma unroll
  for (int b = 0; b < NPE; b++) L.cn[b] = rec[1 + b];
  const uint32_t ow = (uint32_t)rec[1 + NPE + (la >> 1)];
  L.own = valid ? (int)((la & 1) ? (ow >> 16) : (ow & 0xFFFFu)) : 0xFFFF;
#pragma unroll
  for (int w = 0; w < PW; w++) {
    const uint32_t pw = (uint32_t)rec[1 + NPE + OW + la * PW + w];
#pragma unroll
    for (int q = 0; q < 4; q++) if (w * 4 + q < NPE) L.ps[w * 4 + q] = (uint8_t)(pw >> (8 * q));
  }
}

template <class SH, int FAM, bool WANT_K, bool KEEP>
__global__ __launch_bounds__(SH::PBLOCK) void assemble_patch(const Params P) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, BLK = SH::PBLOCK, GPB = BLK / NPE;   // element groups per pass
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? RANK + 1 : RANK;
  extern __shared__ double smem[];
  __shared__ double sErr[(BLK + 63) / 64];
  double* sK = smem;                                         // [plds]            LDS image of the CSR rows
  double* sF = smem + (WANT_K ? P.plds : 0);                 // [kMaxOwn * NDOF]  f_int of the owned nodes
  int64_t* sG = (int64_t*)(sF + kMaxOwn * NDOF);             // [kMaxOwn]         first CSR entry of each node
  int* sOff = (int*)(sG + kMaxOwn);                          // [kMaxOwn + 1]     LDS offset of each node's row block
  int* sNode = sOff + (kMaxOwn + 1);                         // [kMaxOwn]         global node id
  int* sNnb = sNode + kMaxOwn;                               // [kMaxOwn]         neighbour count
  int* sRun = sNnb + kMaxOwn;                                // [kMaxOwn]         runs of consecutive node ids

  const int tid = threadIdx.x;
  const int p = blockIdx.x;
  const int nb0 = P.pn_ptr[p], nown = P.pn_ptr[p + 1] - nb0;
  const int eb0 = P.pe_ptr[p], nelem = P.pe_ptr[p + 1] - eb0;
  const int la = tid % NPE, grp = tid / NPE;

  // software pipeline over the passes: indices two passes ahead, nodal gathers one pass ahead, so a wave
  // never waits a full memory latency between two Gauss-point blocks even at 1-2 waves per SIMD
  LaneIdx<NPE> cur, nxt;
  load_idx<NPE>(P, (int64_t)eb0 + min(grp, nelem - 1), la, grp < nelem, cur);
  nxt = cur;
  if (GPB < nelem) load_idx<NPE>(P, (int64_t)eb0 + min(GPB + grp, nelem - 1), la, GPB + grp < nelem, nxt);
  ElemData<SH, PF> E;
  double H0 = 0.0;
  gather_elem<SH, PF>(P, cur.cn, E);
  if constexpr (PF) { if (P.model == 1 || P.model >= 4) H0 = P.hist_pre[cur.e * SH::NIP + (SH::SHARE ? la : 0)]; }

  int cntsum = 0;
#pragma unroll
  for (int i = 0; i < NDOF; i++) cntsum += P.cnt[i];
  const int r0 = P.pr_ptr[p], nrun = P.pr_ptr[p + 1] - r0;
  for (int j = tid; j < nrun; j += BLK) sRun[j] = P.prun[r0 + j];
  for (int j = tid; j <= nown; j += BLK) {
    sOff[j] = P.poff[nb0 + p + j];                           // nown+1 offsets per patch
    if (j < nown) {
      const int node = P.pnode[nb0 + j];
      sNode[j] = node;
      if constexpr (WANT_K) sG[j] = P.row_ptr[(int64_t)node * NDOF];
    }
  }
  __syncthreads();
  const int ldslen = sOff[nown];
  for (int j = tid; j < nown; j += BLK) sNnb[j] = (sOff[j + 1] - sOff[j]) / cntsum;
  if constexpr (WANT_K)
    for (int i = tid; i < ldslen; i += BLK) sK[i] = 0.0;
  for (int i = tid; i < nown * NDOF; i += BLK) sF[i] = 0.0;
  __syncthreads();

  double err_acc = 0.0;
  for (int kb = 0; kb < nelem; kb += GPB) {
    const bool owned = cur.own != 0xFFFF;                    // idle groups redo the last element, write nothing
    const bool more = kb + GPB < nelem;
    RowBlock<SH, WANT_K, PF, KEEP> R;
    if constexpr (SH::SHARE) {
      IpRec<SH, FAM, KEEP> mine;
      ip_point<SH, FAM, WANT_K, KEEP>(P, cur.e, la, E, H0, owned, mine, err_acc);
      const LaneIdx<NPE> me = cur;
      if (more) {                                            // gathers of the next pass fly during phase 2
        cur = nxt;
        gather_elem<SH, PF>(P, cur.cn, E);
        if constexpr (PF) { if (P.model == 1 || P.model >= 4) H0 = P.hist_pre[cur.e * SH::NIP + la]; }
        const int k2 = kb + 2 * GPB + grp;
        if (kb + 2 * GPB < nelem) load_idx<NPE>(P, (int64_t)eb0 + min(k2, nelem - 1), la, k2 < nelem, nxt);
      }
      shared_phase2<SH, FAM, WANT_K, KEEP>(P, la, mine, R);
      if (me.own != 0xFFFF)
        lds_add_rows<SH, WANT_K, PF, KEEP, true>(P, R, sK, sF, sOff[me.own], sNnb[me.own], me.own * NDOF, me.ps);
    } else {
      if (owned) {
        compute_rowblock<SH, FAM, WANT_K, KEEP>(P, cur.e, la, cur.cn, true, R, err_acc);
        lds_add_rows<SH, WANT_K, PF, KEEP, true>(P, R, sK, sF, sOff[cur.own], sNnb[cur.own], cur.own * NDOF, cur.ps);
      }
      if (more) {
        cur = nxt;
        const int k2 = kb + 2 * GPB + grp;
        if (kb + 2 * GPB < nelem) load_idx<NPE>(P, (int64_t)eb0 + min(k2, nelem - 1), la, k2 < nelem, nxt);
      }
    }
  }
  __syncthreads();

  // ---- stream the finished rows out: runs of consecutive node ids are contiguous in the CSR value array ----
  for (int rr = 0; rr < nrun; rr++) {
    const int run = sRun[rr];
    const int j0 = run & 0xFFFF, cnt = run >> 16;
#if defined(FA_ABLATE) && (FA_ABLATE & 1)
    if (cnt < 0)
#endif
    if constexpr (WANT_K) {
      const int lo = sOff[j0], len = sOff[j0 + cnt] - lo;
      double* __restrict__ Kg = P.K + sG[j0];
      if (P.overwrite) for (int i = tid; i < len; i += BLK) Kg[i] = sK[lo + i];
      else             for (int i = tid; i < len; i += BLK) Kg[i] += sK[lo + i];
    }
    double* __restrict__ fg = P.f + (int64_t)sNode[j0] * NDOF;
    if (P.overwrite) for (int i = tid; i < cnt * NDOF; i += BLK) fg[i] = sF[j0 * NDOF + i];
    else             for (int i = tid; i < cnt * NDOF; i += BLK) fg[i] += sF[j0 * NDOF + i];
  }
  if (PF && P.err_part != nullptr) reduce_err<BLK>(err_acc, sErr, &P.err_part[blockIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// Row-gather kernel on contiguous node ranges, deterministic summation order (FA_DETERMINISTIC)
// ------------------------------------------------------------------------------------------------
template <class SH, int FAM, bool WANT_K, bool KEEP, bool ORDERED, class SP>
__global__ __launch_bounds__(SH::BLOCK) void assemble_rows(const Params P) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, kBlock = SH::BLOCK, kLdsDoubles = SH::LDS;
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? RANK + 1 : RANK;
  __shared__ __attribute__((aligned(16))) double sKbuf[WANT_K ? kLdsDoubles + 2 : 2];
  __shared__ double sF[kBlock * NDOF];
  __shared__ double sErr[kBlock / 64];

  const int tid = threadIdx.x;
  // blockIdx % 8 is the XCD: give every XCD one contiguous band of tiles, so that the tile rows that gather the
  // same nodes meet in one L2
  const int per = (int)gridDim.x >> 3;                 // the grid is a multiple of 8
  const int tile0 = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (tile0 >= P.ntile) return;
  const int tile = P.tile_list ? P.tile_list[tile0] : tile0;   // interface / interior part of a pass (fa_set_interface)
  const int32_t* __restrict__ wi = P.wginfo + (int64_t)tile * kWgInfo;
  const int n0 = wi[0], n1 = wi[1], t0 = wi[2], nlane = wi[3];
  const int64_t base = ((int64_t)(uint32_t)wi[4]) | ((int64_t)wi[5] << 32);
  const int odd = (int)(base & 1);                     // image shifted by one double when K + base is not 16-byte aligned
  double* sK = sKbuf + odd;
  const int len = WANT_K ? wi[6] : 0;
  const int nf = (n1 - n0) * NDOF;
  const bool active = tid < nlane;
  // lane record: element, its nodes, row offset / neighbour count of this lane's node, column positions
  constexpr int LS = lrec_stride(NPE);
  int32_t rec[LS];
  {
    const int4* __restrict__ rp4 = (const int4*)(P.lrec + ((int64_t)t0 + (active ? tid : 0)) * LS);
#pragma unroll
    for (int q = 0; q < LS / 4; q++) { const int4 v = rp4[q]; rec[4 * q] = v.x; rec[4 * q + 1] = v.y; rec[4 * q + 2] = v.z; rec[4 * q + 3] = v.w; }
  }
  if constexpr (WANT_K)
    for (int i = tid; i < len; i += kBlock) sK[i] = 0.0;
  for (int i = tid; i < nf; i += kBlock) sF[i] = 0.0;

  RowBlock<SH, WANT_K, PF, KEEP> R;
  int32_t cn[NPE];
  int la = 0, slot = -1, a = n0, rs0 = 0, nnb = 0;
  int64_t e = 0;
  uint8_t ps[NPE];
  double err_acc = 0.0;
  if (active) {
    e = rec[0] / NPE; la = rec[0] - (int)e * NPE;
#pragma unroll
    for (int b = 0; b < NPE; b++) cn[b] = rec[1 + b];
    const uint32_t w0 = (uint32_t)rec[1 + NPE];
    rs0 = (int)(w0 & 0xFFFFu); a = n0 + (int)((w0 >> 16) & 0xFFu); nnb = (int)(w0 >> 24);
#pragma unroll
    for (int b = 0; b < NPE; b++) ps[b] = (uint8_t)((uint32_t)rec[2 + NPE + (b >> 2)] >> (8 * (b & 3)));
    if constexpr (ORDERED) slot = t0 + tid - P.n2e_ptr[a];
    compute_rowblock<SH, FAM, WANT_K, KEEP, SP>(P, e, la, cn, true, R, err_acc);
  }
  __syncthreads();

  if constexpr (ORDERED) {
    // phased accumulation: two lanes of one phase never share a node => no atomics, and every CSR entry is
    // summed in ascending element order exactly like the reference's serial element loop
    for (int s = 0; s < P.max_valence; s++) {
      if (active && slot == s)
        lds_add_rows<SH, WANT_K, PF, KEEP, false>(P, R, sK, sF, rs0, nnb, (a - n0) * NDOF, ps);
      __syncthreads();
    }
  } else {
    // one pass of ds_add_f64: the (at most valence) addends of an entry arrive in any order
    if (active) lds_add_rows<SH, WANT_K, PF, KEEP, true>(P, R, sK, sF, rs0, nnb, (a - n0) * NDOF, ps);
    __syncthreads();
  }

  if constexpr (WANT_K) {
    if ((((uintptr_t)P.K) & 15) == 0) stream_out16<kBlock>(P.K + base, sK, len, odd, P.overwrite);
    else stream_out<kBlock>(P.K + base, sK, len, P.overwrite);
  }
  stream_out<kBlock>(P.f + (int64_t)n0 * NDOF, sF, nf, P.overwrite);
  if (PF && P.err_part != nullptr) reduce_err<kBlock>(err_acc, sErr, &P.err_part[tile]);
}

// ------------------------------------------------------------------------------------------------
// Row kernel with Gauss points shared by lane pairs (Quad4, Hex8: NIP even)
// ------------------------------------------------------------------------------------------------
// Same tiles and LDS row image as assemble_rows, but the lanes come in aligned pairs (2k, 2k+1) that work on the
// same element: lane "half" h evaluates the Gauss points [h*NIP/2, (h+1)*NIP/2), the two lanes swap every
// Gauss-point record through a DPP quad permute and each accumulates the rows of its own node from both.  An
// element with a single node in the tile still gets two lanes; the second one only helps (flag bit 0 clear).
// Per lane: NIP/2 Gauss points instead of NIP.  Record word LS-1: bit 0 owner, bit 1 half, bits 2.. = "this lane
// stores the per-point outputs (history, strain cache, errExt_) of its k-th point".
template <class T>
__device__ __forceinline__ T pair_swap(T v) {
  static_assert(sizeof(T) == 8, "doubles");
  int lo = __double2loint(v), hi = __double2hiint(v);
  // quad_perm [1,0,3,2]; bound_ctrl + no "old" operand: every lane of an active pair reads its active partner, so
  // the destination needs no initialisation
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <class SH, int FAM, bool KEEP>
__device__ __forceinline__ void swap_rec(const IpRec<SH, FAM, KEEP>& m, IpRec<SH, FAM, KEEP>& r) {
  constexpr bool PF = (FAM != FAM_LE), GEN = (FAM == FAM_PF_MIEHE);
  r.w = pair_swap(m.w);
#pragma unroll
  for (int a = 0; a < SH::NPE; a++)
#pragma unroll
    for (int i = 0; i < SH::RANK; i++) r.g[a][i] = pair_swap(m.g[a][i]);
  if constexpr (GEN) {
#pragma unroll
    for (int i = 0; i < SH::S; i++)
#pragma unroll
      for (int j = i; j < SH::S; j++) { r.D[i][j] = pair_swap(m.D[i][j]); r.D[j][i] = r.D[i][j]; }
    r.d0 = r.d1 = r.d3 = 0.0;
  } else {
    r.d0 = pair_swap(m.d0); r.d1 = pair_swap(m.d1); r.d3 = pair_swap(m.d3);
  }
#pragma unroll
  for (int c = 0; c < SH::S; c++) r.sig[c] = pair_swap(m.sig[c]);
  r.c1 = r.cf = r.dg = r.dgl = 0.0;
  if constexpr (KEEP) {
#pragma unroll
    for (int c = 0; c < SH::S; c++) r.sp[c] = pair_swap(m.sp[c]);
    r.dg = pair_swap(m.dg); r.dgl = pair_swap(m.dgl);
  }
  if constexpr (PF) {
    r.c1 = pair_swap(m.c1); r.cf = pair_swap(m.cf);
#pragma unroll
    for (int i = 0; i < SH::RANK; i++) r.gphi[i] = pair_swap(m.gphi[i]);
  }
}

// two waves per SIMD (256 registers): the variants with off-diagonal blocks would otherwise take 270+ registers and
// run at one wave per SIMD; a few spilled values cost less than the lost overlap
#ifndef FA_PAIR_ATTR
#define FA_PAIR_ATTR __attribute__((amdgpu_waves_per_eu(2)))
#endif
#ifndef FA_PAIR_ACC
#define FA_PAIR_ACC 0          /* 1: FA_ACCUMULATE passes of the lane-pair kernel initialise the LDS image from K instead of the
                                  read-modify-write stream-out.  Measured slower on cfg3 (6.14 against 5.6 ms): the reads sit in
                                  front of the gathers instead of behind the arithmetic, and a barrier waits for them. */
#endif
#ifndef FA_PAIR_EARLY_GATHER
#define FA_PAIR_EARLY_GATHER 1
#endif
#ifndef FA_PAIR_PERS_ACC
#define FA_PAIR_PERS_ACC 0   /* accumulate through the persistent kernel: 5.75 against 5.31 ms on cfg3 */
#endif
#ifndef FA_PAIR_PERS_GATHER
#define FA_PAIR_PERS_GATHER 0
#endif
#ifndef FA_PAIR_ACC_LDSTAB
#define FA_PAIR_ACC_LDSTAB 1   /* ... with the shape tables from the LDS copy, like the overwrite kernel */
#endif
// LDSTAB: the lane's shape-table row comes from an LDS copy (per-lane address, a broadcast read); otherwise both
// candidates come from scalar loads and are picked with selects.  Measured on cfg3 / cfg2 (MI355X, round 2): the LDS
// copy wins for the u-phi matrix pass in overwrite mode (4.55 against 4.89 ms), the selects win for the accumulate
// mode (7.06 / 7.40), with off-diagonal blocks (4.51 / 4.95) and for the elastic model (2.21 / 2.32).
// PERS (P.pair_pers workgroups, each walking tiles bid, bid + grid, ...): the lane records of the NEXT tile are requested
// before the image of the current one is streamed out and its gathers are issued right behind the stream-out, so the two
// memory round trips in front of a tile's arithmetic run under the write-out of the tile before it.  Needs padded records
// (the record address must not depend on the descriptor) and the early-gather order; the image itself is single-buffered
// (a barrier separates the last read of the stream-out from the zero fill).
template <class SH, int FAM, bool WANT_K, bool KEEP, class SP, bool LDSTAB, bool ACC = false, bool PERS = false>
__global__ __launch_bounds__(SH::BLOCK) FA_PAIR_ATTR void assemble_rows_pair(const Params P) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, NIP = SH::NIP, HALF = SH::NIP / 2, kBlock = SH::BLOCK, kLdsDoubles = SH::LDS;
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? RANK + 1 : RANK;
  static_assert(NIP % 2 == 0, "pairs split the Gauss points in halves");
  __shared__ __attribute__((aligned(16))) double sKbuf[WANT_K ? kLdsDoubles + 2 : 2];
  __shared__ double sF[kBlock * NDOF];
  __shared__ double sErr[kBlock / 64];
  // shape tables in LDS, one row per Gauss point: dN/dxi [NPE][RANK], N [NPE], w, padded to an even number of doubles.
  // A lane's point depends on its half, so table reads are per-lane: from LDS they cost a broadcast read instead of a
  // trip to the vector memory pipeline (or 2 selects per value from scalar loads: measured slower in this VALU-bound kernel)
  constexpr int TROW = (NPE * (RANK + 1) + 1 + 1) & ~1;
  __shared__ __attribute__((aligned(16))) double sTabL[LDSTAB ? NIP * TROW : 2];

  const int tid = threadIdx.x;
#ifdef FA_TRACE
  long long tr[8]; tr[0] = __builtin_readcyclecounter();
#endif
  if (LDSTAB && tid < NIP * TROW) {
    const int ipt = tid / TROW, q = tid - ipt * TROW;
    const const_dptr tN = (const_dptr)P.tab, tD = tN + NIP * NPE, tW = tN + NIP * NPE * (1 + RANK);
    sTabL[tid] = (q < NPE * RANK) ? tD[ipt * NPE * RANK + q] : ((q < NPE * (RANK + 1)) ? tN[ipt * NPE + (q - NPE * RANK)] : ((q == NPE * (RANK + 1)) ? tW[ipt] : 0.0));
  }
  // XCD bands as in assemble_rows: virtual block v -> tile (v & 7) * per + (v >> 3), v = blockIdx.x (+ k * gridDim.x: PERS)
  const int vtot = PERS ? ((P.ntile2 + 7) & ~7) : (int)gridDim.x;
  const int per = vtot >> 3;
  int vb = (int)blockIdx.x;
  const int tile0 = (vb & 7) * per + (vb >> 3);
  if (tile0 >= P.ntile2) return;
  int tile = P.tile_list ? P.tile_list[tile0] : tile0;   // interface / interior part of a pass (fa_set_interface)
  constexpr int LS = lrec_stride2(NPE);
  int32_t rec[LS];
  auto load_rec_padded = [&](int tl) {
    const int4* __restrict__ rp4 = (const int4*)(P.lrec2 + ((int64_t)tl * kBlock + tid) * LS);
#pragma unroll
    for (int q = 0; q < LS / 4; q++) { const int4 v = rp4[q]; rec[4 * q] = v.x; rec[4 * q + 1] = v.y; rec[4 * q + 2] = v.z; rec[4 * q + 3] = v.w; }
  };
  if (P.rec_padded) load_rec_padded(tile);             // one workgroup's worth of records per tile: no wait for the descriptor
  RowBlock<SH, WANT_K, PF, KEEP> R;
  int32_t cn[NPE];
  int la = 0, aloc = 0, rs0 = 0, nnb = 0, half = 0;
  int64_t e = 0;
  uint32_t fl = 0;
  bool owner = false;
  uint8_t ps[NPE];
  double err_acc = 0.0;
  double H0v[HALF];
  ElemData<SH, PF> E;
#pragma unroll
  for (int k = 0; k < HALF; k++) H0v[k] = 0.0;
  // the gathers only need the lane record: FA_PAIR_EARLY_GATHER issues them in front of the image initialisation and its
  // barrier, so that their round trip runs while the image is zeroed (which itself waits for the tile descriptor)
  auto decode_and_gather = [&] {
    e = rec[0] / NPE; la = rec[0] - (int)e * NPE;
#pragma unroll
    for (int b = 0; b < NPE; b++) cn[b] = rec[1 + b];
    const uint32_t w0 = (uint32_t)rec[1 + NPE];
    rs0 = (int)(w0 & 0xFFFFu); aloc = (int)((w0 >> 16) & 0xFFu); nnb = (int)(w0 >> 24);
#pragma unroll
    for (int b = 0; b < NPE; b++) ps[b] = (uint8_t)((uint32_t)rec[2 + NPE + (b >> 2)] >> (8 * (b & 3)));
    fl = (uint32_t)rec[LS - 1];
    owner = fl & 1u;
    half = (fl >> 1) & 1;
    if constexpr (PF) {
      if (SP::model(P) == 1 || SP::model(P) >= 4) {
#pragma unroll
        for (int k = 0; k < HALF; k++) H0v[k] = P.hist_pre[e * NIP + half * HALF + k];
      }
    }
    gather_elem<SH, PF, SP>(P, cn, E);
  };
  bool gathered = false;                               // PERS: the gathers of this tile were issued under the last stream-out
#pragma clang loop unroll(disable)
  for (;;) {
  const int32_t* __restrict__ wi = P.wginfo2 + (int64_t)tile * kWgInfo;
  const int n0 = wi[0], n1 = wi[1], nlane = wi[3];
  const int64_t t0 = ((int64_t)(uint32_t)wi[2]) | ((int64_t)wi[7] << 32);
  const int64_t base = ((int64_t)(uint32_t)wi[4]) | ((int64_t)wi[5] << 32);
  const int len = WANT_K ? wi[6] : 0;
  const int odd = (int)(base & 1);                     // image shifted by one double when K + base is not 16-byte aligned
  double* sK = sKbuf + odd;
  const int nf = (n1 - n0) * NDOF;
  const bool active = tid < nlane;                     // nlane is even: pairs are active together
  if (!P.rec_padded) {
    const int4* __restrict__ rp4 = (const int4*)(P.lrec2 + (t0 + (active ? tid : 0)) * LS);
#pragma unroll
    for (int q = 0; q < LS / 4; q++) { const int4 v = rp4[q]; rec[4 * q] = v.x; rec[4 * q + 1] = v.y; rec[4 * q + 2] = v.z; rec[4 * q + 3] = v.w; }
  }
  // (padded records: idle lanes hold a zero record, i.e. element 0 / node 0 -- a harmless gather --, so nobody waits for
  //  the descriptor's lane count either)
  // Measured on MI355X (round 3): cfg3 u-phi overwrite 4.52 -> 4.27 ms, MiehePhase 5.35 -> 5.21; with off-diagonal blocks
  // the longer live ranges spill (4.56 -> 5.74 ms) and the elastic kernel loses 2 %, so those keep the old order.
  const bool early = PERS || (FA_PAIR_EARLY_GATHER && !KEEP && PF && P.rec_padded);
  if (early && !gathered) decode_and_gather();
  if constexpr (ACC) {                                 // FA_ACCUMULATE: the image starts as the caller's K / f (stream_in16)
    if constexpr (WANT_K) {
      if ((((uintptr_t)P.K) & 15) == 0) stream_in16<kBlock>(P.K + base, sK, len, odd);
      else for (int i = tid; i < len; i += kBlock) sK[i] = P.K[base + i];
    }
    for (int i = tid; i < nf; i += kBlock) sF[i] = P.f[(int64_t)n0 * NDOF + i];
  } else {
    if constexpr (WANT_K)
      for (int i = tid; i < len; i += kBlock) sK[i] = 0.0;
    for (int i = tid; i < nf; i += kBlock) sF[i] = 0.0;
  }
  __syncthreads();                                     // image initialised: from here every wave adds as soon as it has computed

  if (active) {
    if (!early) decode_and_gather();
#ifdef FA_TRACE
    if (E.x[0][0] + E.u[3][1] + H0v[0] == 1.234e300) err_acc = 1.0;     // wait for the gathers
    tr[1] = __builtin_readcyclecounter();
#endif
    R.zero();
    // the matrix kernels of the isotropic phase-field family gain from both points in flight (more ILP and, as it
    // happens, fewer registers); the force-only, elastic and Miehe kernels are faster with the loop rolled
    constexpr int kUnroll = (WANT_K && FAM == FAM_PF_ISO) ? 2 : 1;
#pragma unroll kUnroll
    for (int k = 0; k < HALF; k++) {
      const int ip = half * HALF + k, ipo = (half ^ 1) * HALF + k;      // own point, partner's point
      IpRec<SH, FAM, KEEP> r, rp;
      double H0 = 0.0;
#pragma unroll
      for (int q = 0; q < HALF; q++) if (q == k) H0 = H0v[q];
      // reference-element data of the lane's own point (k or HALF + k) and of its partner's, from the LDS copy
      double dn[NPE][RANK], Nv[NPE], No[NPE], wt;
      if constexpr (LDSTAB) {
        const double* __restrict__ row = sTabL + ip * TROW;
        const double* __restrict__ rowo = sTabL + ipo * TROW;
#pragma unroll
        for (int a = 0; a < NPE; a++) {
#pragma unroll
          for (int d = 0; d < RANK; d++) dn[a][d] = row[a * RANK + d];
          Nv[a] = row[NPE * RANK + a]; No[a] = rowo[NPE * RANK + a];
        }
        wt = row[NPE * (RANK + 1)];
      } else {
        const const_dptr tN = (const_dptr)P.tab, tD = tN + NIP * NPE, tW = tN + NIP * NPE * (1 + RANK);
        const int i0 = k, i1 = HALF + k;
#pragma unroll
        for (int a = 0; a < NPE; a++) {
          const double n0 = tN[i0 * NPE + a], n1 = tN[i1 * NPE + a];
          Nv[a] = half ? n1 : n0; No[a] = half ? n0 : n1;
#pragma unroll
          for (int d = 0; d < RANK; d++) { const double d0 = tD[(i0 * NPE + a) * RANK + d], d1 = tD[(i1 * NPE + a) * RANK + d]; dn[a][d] = half ? d1 : d0; }
        }
        const double w0 = tW[i0], w1 = tW[i1];
        wt = half ? w1 : w0;
      }
      ip_point_core<SH, FAM, WANT_K, KEEP, SP>(P, e, ip, dn, Nv, wt, E, H0, (fl >> (2 + k)) & 1u, r, err_acc);
      accum_rows<SH, FAM, WANT_K, KEEP>(P, r, la, Nv, R);
      swap_rec<SH, FAM, KEEP>(r, rp);
      accum_rows<SH, FAM, WANT_K, KEEP>(P, rp, la, No, R);
    }
  }
#ifdef FA_TRACE
  tr[2] = __builtin_readcyclecounter();
  tr[3] = tr[2];
#endif
  if (owner) lds_add_rows<SH, WANT_K, PF, KEEP, true>(P, R, sK, sF, rs0, nnb, aloc * NDOF, ps);
#ifdef FA_TRACE
  tr[6] = __builtin_readcyclecounter();
#endif
  __syncthreads();
#ifdef FA_TRACE
  tr[7] = __builtin_readcyclecounter();
#endif
  // PERS: the next tile of this workgroup; its lane records are requested before the write-out of this one
  bool has_next = false;
  int tile_next = 0;
  if constexpr (PERS) {
    vb += (int)gridDim.x;
    if (vb < vtot) {
      const int tn0 = (vb & 7) * per + (vb >> 3);
      if (tn0 < P.ntile2) { has_next = true; tile_next = P.tile_list ? P.tile_list[tn0] : tn0; }
    }
    if (has_next) load_rec_padded(tile_next);          // rec[] of this tile is dead: everything was decoded out of it
  }
  if constexpr (WANT_K) {
    if ((((uintptr_t)P.K) & 15) == 0) stream_out16<kBlock>(P.K + base, sK, len, odd, ACC || P.overwrite);
    else stream_out<kBlock>(P.K + base, sK, len, ACC || P.overwrite);
  }
  stream_out<kBlock>(P.f + (int64_t)n0 * NDOF, sF, nf, ACC || P.overwrite);
#ifdef FA_TRACE
  tr[4] = __builtin_readcyclecounter();
  __builtin_amdgcn_s_waitcnt(0);
  tr[5] = __builtin_readcyclecounter();
  if ((tid & 63) == 0 && P.dbg != nullptr) {
    long long* out = (long long*)P.dbg + ((int64_t)tile * 4 + (tid >> 6)) * 8;
    out[0] = tr[0]; out[1] = tr[1]; out[2] = tr[2]; out[3] = tr[3]; out[4] = tr[6]; out[5] = tr[7]; out[6] = tr[4]; out[7] = tr[5];
  }
#endif
  if constexpr (PERS) {
    if (FA_PAIR_PERS_GATHER && has_next) { decode_and_gather(); gathered = true; }   // in flight while the stores drain and the image is zeroed
  }
  if (PF && P.err_part != nullptr) { reduce_err<kBlock>(err_acc, sErr, &P.err_part[tile]); err_acc = 0.0; }
  if (!PERS || !has_next) break;
  __syncthreads();                                     // every lane has read its share of the image: it may be zeroed
  tile = tile_next;
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

template <class SH, int FAM, bool WANT_K, bool KEEP, class SP>
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
    compute_rowblock<SH, FAM, WANT_K, KEEP, SP>(P, e, la, cn, true, R, err_acc);
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
      // fa_create has verified that the pattern holds every entry written here (create_impl, "the CSR pattern has no
      // entry"), so find_col always succeeds: nothing is ever dropped
      for (int b = 0; b < NPE; b++) {
        for (int i = 0; i < RANK; i++) {
          const int64_t row = dof(a, i);
          for (int j = 0; j < RANK; j++) atomicAdd(&P.K[find_col(P, row, dof(cn[b], j))], R.Kuu[i][b * RANK + j]);
          if constexpr (KEEP) {
            if (P.keep2) atomicAdd(&P.K[find_col(P, row, dof(cn[b], RANK))], R.Kup[i][b]);
          }
        }
        if constexpr (PF) {
          const int64_t row = dof(a, RANK);
          atomicAdd(&P.K[find_col(P, row, dof(cn[b], RANK))], R.Kpp[b]);
          if constexpr (KEEP) {
            if (P.keep3) {
              for (int j = 0; j < RANK; j++) atomicAdd(&P.K[find_col(P, row, dof(cn[b], j))], R.Kpu[b * RANK + j]);
            }
          }
        }
      }
    }
  }
  if (PF && P.err_part != nullptr) reduce_err<kBlock>(err_acc, sErr, &P.err_part[blockIdx.x]);
}

// ------------------------------------------------------------------------------------------------
template <class SH, int FAM, bool WANT_K, bool KEEP>
static int launch_one(const Params& P, int path, int grid, hipStream_t st) {
  if (path == PATH_PATCH) {
    if constexpr (FAM == FAM_PF_MIEHE) return -1000;
    else {
    auto kern = assemble_patch<SH, FAM, WANT_K, KEEP>;
    const size_t shmem = patch_shmem_bytes(WANT_K ? P.plds : 0, (FAM != FAM_LE) ? SH::RANK + 1 : SH::RANK);
    if (shmem > 65536) {                           // per device: set on every launch (microseconds)
      hipError_t rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (rc != hipSuccess) return (int)rc;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(SH::PBLOCK), shmem, st, P);
    }
  } else if (path == PATH_ROWS_PAIR) {
    if constexpr (SH::NIP % 2 != 0) return -1000;
    else {
      // persistent form (P.pair_pers workgroups), padded lane records: the overwrite GET_MATRIX0 pass of the isotropic
      // phase-field family only.  Measured (cfg3 mesh, round 3, one workgroup per tile -> persistent): AmorPhase 4.30 -> 3.93 ms,
      // WRITTEN pattern 3.70 -> 3.40, BourdinPhase 4.44 -> 4.18; and where it is NOT used: accumulate 5.31 -> 5.82, MiehePhase
      // 5.12 -> 5.43, elastic 2.15 -> 2.27, micromorphic 5.05-5.79 -> 5.77-6.13, GradientDamage 6.69 -> 6.69 (the loop costs
      // 15-45 registers -- uniform values kept live around it spill into vector registers -- which those kernels do not have)
      const bool pers = WANT_K && !KEEP && P.pair_pers > 0 && P.rec_padded && grid > P.pair_pers;
      const int pgrid = P.pair_pers;
      auto go = [&](auto spec) {
        using SP = decltype(spec);
        if constexpr (WANT_K && FAM == FAM_PF_ISO && !KEEP) {
          if constexpr (!SP::gd && SH::RANK == 2) {      // (the opt-in Hex8 lane pairs keep one workgroup per tile)
            if ((P.overwrite || FA_PAIR_PERS_ACC) && pers) {
              hipLaunchKernelGGL((assemble_rows_pair<SH, FAM, WANT_K, KEEP, SP, true, false, true>), dim3(pgrid), dim3(SH::BLOCK), 0, st, P);
              return;
            }
          }
          if (P.overwrite) { hipLaunchKernelGGL((assemble_rows_pair<SH, FAM, WANT_K, KEEP, SP, true>), dim3(grid), dim3(SH::BLOCK), 0, st, P); return; }
        }
        if constexpr (WANT_K) {                          // FA_ACCUMULATE: the image starts as K, the write-out is a store stream
          if (!P.overwrite && FA_PAIR_ACC) { hipLaunchKernelGGL((assemble_rows_pair<SH, FAM, WANT_K, KEEP, SP, FA_PAIR_ACC_LDSTAB && FAM == FAM_PF_ISO && !KEEP, true>), dim3(grid), dim3(SH::BLOCK), 0, st, P); return; }
        }
        hipLaunchKernelGGL((assemble_rows_pair<SH, FAM, WANT_K, KEEP, SP, false>), dim3(grid), dim3(SH::BLOCK), 0, st, P);
      };
      if constexpr (FAM == FAM_LE) go(SpecFixed<0, 0, 0>{});
      else if constexpr (FAM == FAM_PF_ISO) {
        if (P.model == 1 && P.ptype == 0 && P.material == 1) go(SpecFixed<1, 0, 1>{});
        else if (P.model == 7) gd_dispatch<SpecCanon>(P, go);
        else go(SpecCanon{});
      } else {
        if (P.model == 1 && P.ptype == 0) go(SpecFixed<1, 0, 3>{});
        else if (P.model == 4 && P.ptype == 0 && P.material == 3) go(SpecFixed<4, 0, 3>{});   // MicroPhaseFracture, AT1 / AT2, MiehePhase
        // (BourdinPhase / AmorPhase folded in the same way measured slower: 4.99 against 4.77 ms, 5.18 against 5.08 ms)
        else if (is_micro_amor(P)) go(SpecMA<SpecCanon>{});
        else go(SpecCanon{});
      }
    }
  } else if (path == PATH_ROWS || path == PATH_ROWS_ORDERED) {
    // the common configurations get a kernel with model / exponent / material folded in
    const bool ord = path == PATH_ROWS_ORDERED;
    auto go = [&](auto spec) {
      using SP = decltype(spec);
      if (ord) hipLaunchKernelGGL((assemble_rows<SH, FAM, WANT_K, KEEP, true, SP>), dim3(grid), dim3(SH::BLOCK), 0, st, P);
      else     hipLaunchKernelGGL((assemble_rows<SH, FAM, WANT_K, KEEP, false, SP>), dim3(grid), dim3(SH::BLOCK), 0, st, P);
    };
    if constexpr (FAM == FAM_LE) go(SpecFixed<0, 0, 0>{});
    else if constexpr (FAM == FAM_PF_ISO) {
      if (P.model == 1 && P.ptype == 0 && P.material == 1) go(SpecFixed<1, 0, 1>{});
      else if (P.model == 7) gd_dispatch<SpecCanon>(P, go);
      else go(SpecCanon{});
    } else {
      if (P.model == 1 && P.ptype == 0) go(SpecFixed<1, 0, 3>{});
      else if (is_micro_amor(P)) go(SpecMA<SpecCanon>{});
      else go(SpecCanon{});
    }
  } else {
    if constexpr (FAM == FAM_PF_ISO) {
      if (P.model == 7) { hipLaunchKernelGGL((assemble_generic<SH, FAM, WANT_K, KEEP, SpecGD<SpecRuntime>>), dim3(grid), dim3(SH::BLOCK), 0, st, P); return (int)hipGetLastError(); }
    }
    if constexpr (FAM == FAM_PF_MIEHE) {
      if (is_micro_amor(P)) { hipLaunchKernelGGL((assemble_generic<SH, FAM, WANT_K, KEEP, SpecMA<SpecRuntime>>), dim3(grid), dim3(SH::BLOCK), 0, st, P); return (int)hipGetLastError(); }
    }
    hipLaunchKernelGGL((assemble_generic<SH, FAM, WANT_K, KEEP, SpecRuntime>), dim3(grid), dim3(SH::BLOCK), 0, st, P);
  }
  return (int)hipGetLastError();
}

template <class SH>
static int launch_shape(const Params& P, int fam, bool want_K, bool keep, int path, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (fam == FAM_LE)
    return want_K ? launch_one<SH, FAM_LE, true, false>(P, path, grid, st)
                  : launch_one<SH, FAM_LE, false, false>(P, path, grid, st);
  if (fam == FAM_PF_ISO) {
    if (!want_K) return launch_one<SH, FAM_PF_ISO, false, false>(P, path, grid, st);
    return keep ? launch_one<SH, FAM_PF_ISO, true, true>(P, path, grid, st)
                : launch_one<SH, FAM_PF_ISO, true, false>(P, path, grid, st);
  }
  if (fam == FAM_PF_MIEHE) {
    if (!want_K) return launch_one<SH, FAM_PF_MIEHE, false, false>(P, path, grid, st);
    return keep ? launch_one<SH, FAM_PF_MIEHE, true, true>(P, path, grid, st)
                : launch_one<SH, FAM_PF_MIEHE, true, false>(P, path, grid, st);
  }
  return -1000;   // family not built for this shape
}

}  // namespace fa
