// fa_emat_impl.h -- element-matrix tile kernel for the shapes whose element matrices fit a lane's registers
// (Quad4, Triangle3): BASELINE configs 2 and 3, GET_MATRIX0 and GET_INT_VECTOR.
//
// Same tiles as the tile kernel (fa_tile_impl.h: a compact set of owned nodes + every element that touches one of
// them; one workgroup per tile), different split of the work:
//   phase 0  stage    nodal coordinates / state of the tile's nodes and the local connectivity go to LDS
//   phase 1  elements LPE lanes per element (1 or 2): each evaluates its share of the Gauss points ONCE
//                     (PhaseFractureModel.cpp:629-703) and sums w B^T D B, the phi-phi block, the optional
//                     off-diagonal blocks and f1 / f2 (:713-730) over them in registers.  K1 and K4 are symmetric, so
//                     only the node-pair blocks a <= b are formed (36 + 10 instead of 64 + 16 values for Quad4).  The
//                     finished element matrix goes to the element's LDS slot (the second lane of a pair adds to it).
//   phase 2  pairs    one lane per (owned node a, neighbour b): the (rank+1)^2 block of the global matrix is the sum of
//                     the slots' blocks over the elements that hold both nodes (<= 4 reads of 5 doubles for Quad4),
//                     written to its CSR entries through a wave-private staging row, 64 consecutive doubles per store.
//                     The diagonal lane (a, a) also sums the node's f_int rows.
// No atomics on K or f_int, in LDS or in HBM: every entry has one writer and a fixed summation order (elements
// ascending, like the reference's serial loop), so the result is bitwise reproducible from run to run.
// GET_INT_VECTOR passes keep 12-double slots (f only) and tiles of up to 256 elements.
//
// Against the lane-pair row kernel (fa_kernel_impl.h) every Gauss point is evaluated 1.2 times per pass instead of
// twice, the node-pair blocks are formed once per element instead of once per (element, node), and nothing is swapped
// between lanes.
#pragma once
#include "fa_kernel_impl.h"

namespace fa {

template <class SH, bool PF, bool WANT_K, bool KEEP>
struct EmatLayout {                                     // offsets (doubles) inside one element slot
  static constexpr int R = SH::RANK, NPE = SH::NPE, NDOF = PF ? R + 1 : R;
  static constexpr int NB = NPE * (NPE + 1) / 2;       // node-pair blocks a <= b
  static constexpr int BW = R * R + (PF ? 1 : 0);      // K1 block + K4 entry
  static constexpr int O_F = WANT_K ? NB * BW : 0;     // f1 / f2: [a][NDOF]
  static constexpr int O_UP = O_F + NPE * NDOF;        // K2: [a][b][R]
  static constexpr int O_PU = O_UP + ((WANT_K && KEEP) ? NPE * NPE * R : 0);   // K3: [a][b][R]
  static constexpr int W = O_PU + ((WANT_K && KEEP) ? NPE * NPE * R : 0);
  static constexpr int SLOTW = W | 1;                  // odd stride: slots spread over the LDS banks
  static_assert(SLOTW == emat_slot_doubles(R, NPE, PF, WANT_K, KEEP), "host and device disagree on the slot width");
  static __device__ __forceinline__ constexpr int blk(int a, int b) { return a * NPE - a * (a - 1) / 2 + (b - a); }   // a <= b
};

// Sums of one element over its Gauss points: the upper node-pair blocks of K1 / K4, full K2 / K3, f1 / f2.
template <class SH, int FAM, bool WANT_K, bool KEEP>
struct EmatAcc {
  static constexpr int R = SH::RANK, NPE = SH::NPE, NB = NPE * (NPE + 1) / 2;
  static constexpr bool PF = (FAM != FAM_LE), K2 = WANT_K && KEEP;
  double fu[NPE][R], fp[PF ? NPE : 1];
  double Kuu[WANT_K ? NB : 1][R][R], Kpp[(WANT_K && PF) ? NB : 1];
  double Kup[K2 ? NPE : 1][K2 ? NPE : 1][R], Kpu[K2 ? NPE : 1][K2 ? NPE : 1][R];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < NPE; a++) {
#pragma unroll
      for (int i = 0; i < R; i++) fu[a][i] = 0.0;
      if constexpr (PF) fp[a] = 0.0;
    }
    if constexpr (WANT_K) {
#pragma unroll
      for (int b = 0; b < NB; b++) {
#pragma unroll
        for (int i = 0; i < R; i++)
#pragma unroll
          for (int j = 0; j < R; j++) Kuu[b][i][j] = 0.0;
        if constexpr (PF) Kpp[b] = 0.0;
      }
      if constexpr (K2) {
#pragma unroll
        for (int a = 0; a < NPE; a++)
#pragma unroll
          for (int b = 0; b < NPE; b++)
#pragma unroll
            for (int i = 0; i < R; i++) { Kup[a][b][i] = 0.0; Kpu[a][b][i] = 0.0; }
      }
    }
  }
};

// One Gauss-point record into the element sums (PhaseFractureModel.cpp:713-730).  Nv[b] = N_b(ip).
template <class SH, int FAM, bool WANT_K, bool KEEP>
__device__ __forceinline__ void accum_emat(const Params& P, const IpRec<SH, FAM, KEEP>& r, const double (&Nv)[SH::NPE],
                                           EmatAcc<SH, FAM, WANT_K, KEEP>& A) {
  constexpr int R = SH::RANK, NPE = SH::NPE, S = SH::S;
  constexpr bool PF = (FAM != FAM_LE);
  using L = EmatLayout<SH, PF, WANT_K, KEEP>;
  double wg[NPE][R];
#pragma unroll
  for (int a = 0; a < NPE; a++)
#pragma unroll
    for (int i = 0; i < R; i++) wg[a][i] = r.w * r.g[a][i];
#pragma unroll
  for (int a = 0; a < NPE; a++) {
    double t[R];
    Bt_vec<R, S>(wg[a], r.sig, t);                     // f1 += w B^T sigma (:727)
#pragma unroll
    for (int i = 0; i < R; i++) A.fu[a][i] += t[i];
    if constexpr (PF) {
      double gg = 0.0;
#pragma unroll
      for (int i = 0; i < R; i++) gg += wg[a][i] * r.gphi[i];
      A.fp[a] += (r.w * Nv[a]) * r.cf + P.c2 * gg;      // f2 (:728-730)
    }
  }
  if constexpr (WANT_K) {
#pragma unroll
    for (int a = 0; a < NPE; a++) {
      if constexpr (FAM == FAM_PF_MIEHE) {             // K1 += w B_a^T D B_b with a full symmetric D
        double Aw[R][S];
#pragma unroll
        for (int c = 0; c < S; c++) {
          double col[S], t[R];
#pragma unroll
          for (int k = 0; k < S; k++) col[k] = r.D[k][c];
          Bt_vec<R, S>(wg[a], col, t);
#pragma unroll
          for (int i = 0; i < R; i++) Aw[i][c] = t[i];
        }
#pragma unroll
        for (int b = a; b < NPE; b++) {
          const int bi = L::blk(a, b);
#pragma unroll
          for (int i = 0; i < R; i++) {
            double t[R];
            Bt_vec<R, S>(r.g[b], Aw[i], t);
#pragma unroll
            for (int j = 0; j < R; j++) if (b != a || j >= i) A.Kuu[bi][i][j] += t[j];
          }
        }
      } else if constexpr (R == 2) {                   // K1 += w B_a^T D B_b (:713), isotropic-form D
        const double a00 = wg[a][0] * r.d0, a01 = wg[a][0] * r.d1, a03 = wg[a][0] * r.d3;
        const double a10 = wg[a][1] * r.d0, a11 = wg[a][1] * r.d1, a13 = wg[a][1] * r.d3;
#pragma unroll
        for (int b = a; b < NPE; b++) {
          const int bi = L::blk(a, b);
          const double gbx = r.g[b][0], gby = r.g[b][1];
          A.Kuu[bi][0][0] += a00 * gbx + a13 * gby;
          A.Kuu[bi][0][1] += a01 * gby + a13 * gbx;
          if (b != a) A.Kuu[bi][1][0] += a11 * gbx + a03 * gby;
          A.Kuu[bi][1][1] += a10 * gby + a03 * gbx;
        }
      } else {
#pragma unroll
        for (int b = a; b < NPE; b++) {
          const int bi = L::blk(a, b);
          const double dotp = wg[a][0] * r.g[b][0] + wg[a][1] * r.g[b][1] + wg[a][2] * r.g[b][2];
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
              if (b == a && j < i) continue;
              if (i == j) A.Kuu[bi][i][j] += (r.d0 - r.d3) * (wg[a][i] * r.g[b][i]) + r.d3 * dotp;
              else        A.Kuu[bi][i][j] += r.d1 * (wg[a][i] * r.g[b][j]) + r.d3 * (wg[a][j] * r.g[b][i]);
            }
        }
      }
      if constexpr (PF) {
        const double wNa = r.w * Nv[a];
        const double wc1 = wNa * r.c1;
        double c2w[R];
#pragma unroll
        for (int i = 0; i < R; i++) c2w[i] = P.c2 * wg[a][i];
#pragma unroll
        for (int b = a; b < NPE; b++) {                // K4 (:714-716): w c1 N N^T + w c2 grad N . grad N
          double t = A.Kpp[L::blk(a, b)] + wc1 * Nv[b];
#pragma unroll
          for (int i = 0; i < R; i++) t += c2w[i] * r.g[b][i];
          A.Kpp[L::blk(a, b)] = t;
        }
        if constexpr (KEEP) {
          if (P.keep2) {                                // K2 += w g' B^T (sigma+ (x) N) (:720)
            double t[R];
            Bt_vec<R, S>(wg[a], r.sp, t);
#pragma unroll
            for (int b = 0; b < NPE; b++)
#pragma unroll
              for (int i = 0; i < R; i++) A.Kup[a][b][i] += (r.dg * t[i]) * Nv[b];
          }
          if (P.keep3) {                                // K3 += w loading g' N (x) (sigma+^T B) (:721)
            const double c = wNa * r.dgl;
#pragma unroll
            for (int b = 0; b < NPE; b++) {
              double t[R];
              Bt_vec<R, S>(r.g[b], r.sp, t);
#pragma unroll
              for (int j = 0; j < R; j++) A.Kpu[a][b][j] += c * t[j];
            }
          }
        }
      }
    }
  }
}

// The element sums into the element's LDS slot: plain stores (ADD = false) or ds_add_f64 (the second lane of a pair).
template <class SH, int FAM, bool WANT_K, bool KEEP, bool ADD>
__device__ __forceinline__ void emat_to_slot(const EmatAcc<SH, FAM, WANT_K, KEEP>& A, double* sl) {
  constexpr int R = SH::RANK, NPE = SH::NPE;
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? R + 1 : R;
  using L = EmatLayout<SH, PF, WANT_K, KEEP>;
  auto put = [](double* q, double v) {
    if constexpr (ADD) __hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else *q = v;
  };
  if constexpr (WANT_K) {
#pragma unroll
    for (int a = 0; a < NPE; a++)
#pragma unroll
      for (int b = a; b < NPE; b++) {
        const int bi = L::blk(a, b);
#pragma unroll
        for (int i = 0; i < R; i++)
#pragma unroll
          for (int j = 0; j < R; j++)                  // diagonal blocks: the lower half mirrors the upper one
            put(sl + bi * L::BW + i * R + j, (b == a && j < i) ? A.Kuu[bi][j][i] : A.Kuu[bi][i][j]);
        if constexpr (PF) put(sl + bi * L::BW + R * R, A.Kpp[bi]);
      }
    if constexpr (KEEP) {
#pragma unroll
      for (int a = 0; a < NPE; a++)
#pragma unroll
        for (int b = 0; b < NPE; b++)
#pragma unroll
          for (int i = 0; i < R; i++) {
            put(sl + L::O_UP + (a * NPE + b) * R + i, A.Kup[a][b][i]);
            put(sl + L::O_PU + (a * NPE + b) * R + i, A.Kpu[a][b][i]);
          }
    }
  }
#pragma unroll
  for (int a = 0; a < NPE; a++) {
#pragma unroll
    for (int i = 0; i < R; i++) put(sl + L::O_F + a * NDOF + i, A.fu[a][i]);
    if constexpr (PF) put(sl + L::O_F + a * NDOF + R, A.fp[a]);
  }
}

#ifndef FA_FVEC_UNROLL
#define FA_FVEC_UNROLL 1
#endif
#ifndef FA_EMAT_ATTR
#define FA_EMAT_ATTR __attribute__((amdgpu_waves_per_eu(2)))
#endif

template <class SH, int FAM, bool WANT_K, bool KEEP, class SP, int BLK, int LPE>
__global__ __launch_bounds__(BLK) FA_EMAT_ATTR void assemble_emat(const Params P) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, NIP = SH::NIP;
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? RANK + 1 : RANK;
  using L = EmatLayout<SH, PF, WANT_K, KEEP>;
  constexpr int SLOTW = L::SLOTW, NODW = tile_node_doubles(RANK, PF), NK = NIP / LPE;
  static_assert(NIP % LPE == 0 && BLK % LPE == 0 && (LPE == 1 || LPE == 2), "lanes of an element split its Gauss points evenly");
  extern __shared__ double smem[];
  __shared__ double sErr[BLK / 64];

  const int tid = threadIdx.x;
  const int per = (int)gridDim.x >> 3;                 // XCD bands: neighbouring tiles share halo nodes in one L2
  const int tile0 = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (tile0 >= P.ntile3) return;
  const int tile = P.tile_list ? P.tile_list[tile0] : tile0;   // interface / interior part of a pass (fa_set_interface)
  const int32_t* __restrict__ ti = P.tinfo + (int64_t)tile * kTileInfo;
  const int node_off = ti[0], n_all = ti[1], n_own = ti[2], elem_off = ti[3], n_el = ti[4];
  const int pair_off = ti[5], n_pairs = ti[6], ent_off = ti[7];

  double* sSlot = smem;
  double* sNod = sSlot + n_el * SLOTW;
  double* sExt = sNod + n_all * NODW;                 // phi of state0 / state00 (stateExt_), Ext variants only
  const bool ext_state = PF && (SP::model(P) == 2 || SP::model(P) == 3 || (SP::model(P) == 5 || SP::model(P) == 6));
  long long* sRow = reinterpret_cast<long long*>(sExt + (ext_state ? 2 * n_all : 0));   // first CSR entry of every owned node
  int* sDeg = reinterpret_cast<int*>(sRow + n_own);                                     // neighbours of every owned node
  uint16_t* sConn = reinterpret_cast<uint16_t*>(sDeg + ((n_own + 1) & ~1));

  // ---- phase 0: stage ----
  for (int i = tid; i < n_all; i += BLK) {
    const int64_t n = P.tnodes[node_off + i];
    double* nd = sNod + i * NODW;
#pragma unroll
    for (int d = 0; d < RANK; d++) { nd[d] = P.coords[n * RANK + d]; nd[RANK + d] = P.state[n * NDOF + d]; }
    if constexpr (PF) {
      const int64_t dp = n * NDOF + RANK;
      nd[2 * RANK] = P.state[dp];
      if (SP::model(P) == 2 || SP::model(P) == 3) { sExt[2 * i] = P.state0[dp]; sExt[2 * i + 1] = P.stateX[dp]; }
      else if ((SP::model(P) == 5 || SP::model(P) == 6)) { sExt[2 * i] = 0.0; sExt[2 * i + 1] = P.stateX[dp]; }
    }
  }
  if constexpr (WANT_K) {
    for (int i = tid; i < n_own; i += BLK) {
      const int64_t r0 = (int64_t)P.tnodes[node_off + i] * NDOF;
      const long long b0 = P.row_ptr[r0], b1 = P.row_ptr[r0 + 1];
      sRow[i] = b0; sDeg[i] = (P.cnt[0] > 0) ? (int)((b1 - b0) / P.cnt[0]) : 0;
    }
  }
  for (int i = tid; i < n_el * NPE; i += BLK) sConn[i] = P.tconn[(int64_t)elem_off * NPE + i];
  __syncthreads();

  // ---- phase 1: element matrices ----
  double err_acc = 0.0;
  // constant address space: the tables are never written, so the loads stay scalar (s_load) even after the kernel's
  // own global stores -- as plain global pointers they turn into vector loads inside the Gauss-point loop
  const const_dptr tabN = (const_dptr)P.tab;
  const const_dptr tabD = tabN + NIP * NPE;
  const const_dptr tabW = tabN + NIP * NPE * (1 + RANK);
  for (int t0 = 0; t0 < n_el * LPE; t0 += BLK) {         // uniform trip count (the lanes of a pair stay together)
    const int t = t0 + tid;
    const bool act = t < n_el * LPE;
    const int s = act ? t / LPE : 0, h = t - (t / LPE) * LPE;
    const uint32_t ew = (uint32_t)P.telems[elem_off + s];
    const int64_t e = (int64_t)(ew & 0x7FFFFFFFu);
    const bool write = act && (ew >> 31) != 0;         // this tile stores the element's history / strain / errExt_
    const uint16_t* lc = sConn + s * NPE;
    ElemData<SH, PF> E;
#pragma unroll
    for (int a = 0; a < NPE; a++) {
      const int li = (int)lc[a];
      const double* nd = sNod + li * NODW;
#pragma unroll
      for (int i = 0; i < RANK; i++) { E.x[a][i] = nd[i]; E.u[a][i] = nd[RANK + i]; }
      if constexpr (PF) {
        E.phi[a] = nd[2 * RANK];
        if (ext_state) { E.phi0[a] = sExt[2 * li]; E.phiX[a] = sExt[2 * li + 1]; }
        else { E.phi0[a] = 0.0; E.phiX[a] = 0.0; }
      }
    }
    double H0v[NK];
#pragma unroll
    for (int k = 0; k < NK; k++) H0v[k] = 0.0;
    if constexpr (PF) {
      if (SP::model(P) == 1 || SP::model(P) >= 4) {
#pragma unroll
        for (int k = 0; k < NK; k++) H0v[k] = P.hist_pre[e * NIP + k * LPE + h];
      }
    }
    EmatAcc<SH, FAM, WANT_K, KEEP> A;
    A.zero();
#pragma unroll
    for (int k = 0; k < NK; k++) {
      // lane h of the element takes the points k * LPE + h; the reference-element data of both candidates come from
      // scalar loads and are picked with a select (no per-lane table reads)
      double dn[NPE][RANK], Nv[NPE], wt;
      {
        const int i0 = k * LPE;
#pragma unroll
        for (int a = 0; a < NPE; a++) {
          Nv[a] = tabN[i0 * NPE + a];
#pragma unroll
          for (int d = 0; d < RANK; d++) dn[a][d] = tabD[(i0 * NPE + a) * RANK + d];
        }
        wt = tabW[i0];
        if constexpr (LPE == 2) {
          const int i1 = i0 + 1;
#pragma unroll
          for (int a = 0; a < NPE; a++) {
            const double n1 = tabN[i1 * NPE + a];
            Nv[a] = h ? n1 : Nv[a];
#pragma unroll
            for (int d = 0; d < RANK; d++) { const double d1 = tabD[(i1 * NPE + a) * RANK + d]; dn[a][d] = h ? d1 : dn[a][d]; }
          }
          const double w1 = tabW[i1];
          wt = h ? w1 : wt;
        }
      }
      IpRec<SH, FAM, KEEP> r;
      ip_point_core<SH, FAM, WANT_K, KEEP, SP>(P, e, k * LPE + h, dn, Nv, wt, E, H0v[k], write, r, err_acc);
      accum_emat<SH, FAM, WANT_K, KEEP>(P, r, Nv, A);
      __builtin_amdgcn_sched_barrier(0);               // one point at a time: interleaved points cost > 256 registers
    }
    double* sl = sSlot + s * SLOTW;
    if constexpr (LPE == 1) {
      if (act) emat_to_slot<SH, FAM, WANT_K, KEEP, false>(A, sl);
    } else {
      // lane 0 of the pair stores, lane 1 adds: both sit in one wave, whose LDS operations complete in order
      if (act && h == 0) emat_to_slot<SH, FAM, WANT_K, KEEP, false>(A, sl);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (act && h == 1) emat_to_slot<SH, FAM, WANT_K, KEEP, true>(A, sl);
    }
  }
  __syncthreads();

  // ---- phase 2: node-pair blocks of the global matrix, f_int rows ----
  // wave-private staging rows for the coalesced write-out (64 chunks of a row + 64 row bases per wave)
  double* stg = reinterpret_cast<double*>(sConn + ((n_el * NPE + 3) & ~3)) + (tid >> 6) * (64 * NDOF + 64);
  long long* stg_base = reinterpret_cast<long long*>(stg + 64 * NDOF);
  const int lane = tid & 63;
  const uint4* __restrict__ prec = reinterpret_cast<const uint4*>(P.tpairs) + pair_off;
  for (int p0 = 0; p0 < n_pairs; p0 += BLK) {          // uniform trip count: every lane of a wave takes part in the write-out
    const int p = p0 + tid;
    const bool active = p < n_pairs;
    const uint4 pw = active ? prec[p] : make_uint4(0u, 0u, 0u, 0u);
    const uint32_t w0 = pw.x;
    const int ia = (int)(w0 & 0xFFFFu), kpos = (int)((w0 >> 16) & 0xFFu), cnt = (int)(w0 >> 24);
    const uint32_t* __restrict__ ents = P.tents + (int64_t)ent_off + pw.y;
    const bool diag = ((pw.z >> 30) & 1u) != 0;
    double Kuu[RANK][RANK], Kpp = 0.0, Kup[RANK], Kpu[RANK], fs[NDOF];
#pragma unroll
    for (int i = 0; i < RANK; i++) {
      Kup[i] = 0.0; Kpu[i] = 0.0;
#pragma unroll
      for (int j = 0; j < RANK; j++) Kuu[i][j] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < NDOF; i++) fs[i] = 0.0;
    uint32_t ent_next = pw.w;                            // the first entry travels in the pair record
#pragma unroll 1
    for (int t = 0; t < cnt; t++) {
      const uint32_t ent = ent_next;
      if (t + 1 < cnt) ent_next = ents[t + 1];
      const int slot = (int)(ent & 0xFFFFu), la = (int)((ent >> 16) & 0xFFu), lb = (int)(ent >> 24);
      const double* sl = sSlot + slot * SLOTW;
      if constexpr (WANT_K) {
        const bool swapped = la > lb;
        const int lo = swapped ? lb : la, hi = swapped ? la : lb;
        const double* bk = sl + (lo * NPE - ((lo * (lo - 1)) >> 1) + (hi - lo)) * L::BW;
        double v[RANK][RANK];
#pragma unroll
        for (int i = 0; i < RANK; i++)
#pragma unroll
          for (int j = 0; j < RANK; j++) v[i][j] = bk[i * RANK + j];
#pragma unroll
        for (int i = 0; i < RANK; i++)
#pragma unroll
          for (int j = 0; j < RANK; j++) Kuu[i][j] += swapped ? v[j][i] : v[i][j];   // block (b, a) = block (a, b)^T
        if constexpr (PF) Kpp += bk[RANK * RANK];
        if constexpr (KEEP) {
#pragma unroll
          for (int i = 0; i < RANK; i++) {
            Kup[i] += sl[L::O_UP + (la * NPE + lb) * RANK + i];
            Kpu[i] += sl[L::O_PU + (la * NPE + lb) * RANK + i];
          }
        }
      }
      if (diag) {
#pragma unroll
        for (int i = 0; i < NDOF; i++) fs[i] += sl[L::O_F + la * NDOF + i];
      }
    }
    if (active && diag) {                               // f_int rows of owned node ia (:748-749)
      const int64_t q = (int64_t)P.tnodes[node_off + ia] * NDOF;
#pragma unroll
      for (int i = 0; i < NDOF; i++) P.f[q + i] = P.overwrite ? fs[i] : P.f[q + i] + fs[i];
    }
    if constexpr (WANT_K) {
      double blk[NDOF][NDOF];
#pragma unroll
      for (int i = 0; i < RANK; i++) {
#pragma unroll
        for (int j = 0; j < RANK; j++) blk[i][j] = Kuu[i][j];
        if constexpr (PF) {
          blk[i][RANK] = (KEEP && P.keep2) ? Kup[i] : 0.0;
          blk[RANK][i] = (KEEP && P.keep3) ? Kpu[i] : 0.0;
        }
      }
      if constexpr (PF) blk[RANK][RANK] = Kpp;
      // Lanes are in (node, neighbour) order: the chunks of one CSR row held by consecutive lanes are consecutive in
      // memory.  Row by row they pass through LDS so that a store instruction writes 64 consecutive doubles
      // (LDS operations of one wave complete in order; the wave barriers only pin the compiler's schedule).
      long long rb = 0;
      int deg = 0;
      if (active) { rb = sRow[ia]; deg = sDeg[ia]; }
#pragma unroll
      for (int i = 0; i < NDOF; i++) {
        const int ci = P.cnt[i];
        if (ci == 0) continue;
        if (active) {
#pragma unroll
          for (int j = 0; j < NDOF; j++) { const int o = P.off[i][j]; if (o >= 0) stg[lane * ci + o] = blk[i][j]; }
          stg_base[lane] = rb + (long long)kpos * ci;
        } else {
          stg_base[lane] = -1;
        }
        rb += (long long)deg * ci;                      // first entry of the node's next row
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int d = lane; d < 64 * ci; d += 64) {
          const int owner = (ci == 3) ? d / 3 : ((ci == 4) ? (d >> 2) : ((ci == 2) ? (d >> 1) : ((ci == 1) ? d : d / ci)));
          const long long b = stg_base[owner];
          if (b >= 0) {
            const int64_t q = b + (d - owner * ci);
            P.K[q] = P.overwrite ? stg[d] : P.K[q] + stg[d];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (PF && P.err_part != nullptr) reduce_err<BLK>(err_acc, sErr, &P.err_part[tile]);
}

// ------------------------------------------------------------------------------------------------
// GET_INT_VECTOR passes: f_int only
// ------------------------------------------------------------------------------------------------
// One lane per tile element evaluates its Gauss points once and keeps f1 / f2 (PhaseFractureModel.cpp:727-730) of its
// NPE nodes in registers.  The rows of an owned node go to that node's LDS buffer, at the rank the element has among
// the node's incident elements (a byte per (element, node) from the tiling); one lane per (owned node, dof) then sums
// the node's buffer in that order -- the element order of the reference's serial loop -- and writes f_int.  No
// atomics, no metadata read after the staging phase, tiles of a few hundred elements (LDS: 5-7 doubles per node and
// valence x dofs doubles per owned node).
#ifndef FA_FVEC_ATTR
#define FA_FVEC_ATTR
#endif
// Instantiations a few registers above an occupancy step (Quad4 with a general tangent: 173, Triangle3 GradientDamage: 171;
// the LDS of a tile leaves room for three waves per SIMD, 168 registers) are asked for that step: FA_FVEC_CAP.
#ifndef FA_FVEC_CAP_GD
#define FA_FVEC_CAP_GD 1
#endif
#ifndef FA_FVEC_CAP
#define FA_FVEC_CAP 1
#endif
template <class SH, int FAM, class SP>
constexpr int fvec_min_waves() {
  if (FA_FVEC_CAP && SH::RANK == 3) return 2;
  if (FA_FVEC_CAP_GD && SH::RANK == 2 && SH::NPE == 4 && SP::gd_fixed) return 3;   // exponential1 + Mazars: 191 VGPRs
  return (FA_FVEC_CAP && SH::RANK == 2 && ((FAM == FAM_PF_MIEHE && SH::NPE == 4) || (SP::gd && SH::NPE == 3))) ? 3 : 1;
}
template <class SH, int FAM, class SP, int BLK>
__global__ __launch_bounds__(BLK, (fvec_min_waves<SH, FAM, SP>())) FA_FVEC_ATTR void assemble_fvec(const Params P) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, NIP = SH::NIP;
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? RANK + 1 : RANK;
  constexpr int NODW = tile_node_doubles(RANK, PF);
  static_assert(NPE <= 8, "one byte of the rank words per element node");
  constexpr int EW = (NPE + 3) / 4;                    // rank words per element
  extern __shared__ double smem[];
  __shared__ double sErr[BLK / 64];

  const int tid = threadIdx.x;
  const int per = (int)gridDim.x >> 3;                 // XCD bands
  const int tile0 = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (tile0 >= P.ntile3) return;
  const int tile = P.tile_list ? P.tile_list[tile0] : tile0;   // interface / interior part of a pass (fa_set_interface)
  const int32_t* __restrict__ ti = P.tinfo + (int64_t)tile * kTileInfo;
  // Lists padded to a fixed stride per tile (P.tn_stride, P.te_stride > 0): the lane's first tile node and element are
  // addressed from the tile index alone, so their ids -- and then the nodal data and the element's history -- are requested
  // before the descriptor has arrived: two memory round trips in front of the barrier instead of three.  (Idle slots
  // hold node 0 / element 0: harmless reads.)
  const bool pre = P.tn_stride > 0;
  int64_t n_pre = 0;
  uint32_t ew_pre = 0, es_pre[(NPE + 3) / 4], nv_pre = 0;
  double pre_x[RANK], pre_u[NDOF], pre_H[NIP];
  if (pre) {
    const int64_t pn = (int64_t)tile * P.tn_stride + (tid < P.tn_stride ? tid : 0);
    const int64_t pe = (int64_t)tile * P.te_stride + (tid < P.te_stride ? tid : 0);
    n_pre = P.tnodes[pn]; nv_pre = P.nval[pn];
    ew_pre = (uint32_t)P.telems[pe];
#pragma unroll
    for (int q = 0; q < (NPE + 3) / 4; q++) es_pre[q] = P.eslot[pe * ((NPE + 3) / 4) + q];
#pragma unroll
    for (int d = 0; d < RANK; d++) pre_x[d] = P.coords[n_pre * RANK + d];
#pragma unroll
    for (int d = 0; d < NDOF; d++) pre_u[d] = P.state[n_pre * NDOF + d];
    if (PF && (SP::model(P) == 1 || SP::model(P) >= 4)) {
      const int64_t e = (int64_t)(ew_pre & 0x7FFFFFFFu);
#pragma unroll
      for (int k = 0; k < NIP; k++) pre_H[k] = P.hist_pre[e * NIP + k];
    }
  }
  const int node_off = ti[0], n_all = ti[1], n_own = ti[2], elem_off = ti[3], n_el = ti[4];
  const int V = P.fvec_val;                            // buffer rows per owned node (largest valence of the mesh)

  double* sNod = smem;
  double* sExt = sNod + n_all * NODW;
  const bool ext_state = PF && (SP::model(P) == 2 || SP::model(P) == 3 || (SP::model(P) == 5 || SP::model(P) == 6));
  // [n_own][V][NDOF], the rows of a node padded to an odd number of doubles: consecutive nodes then start on 16
  // different bank pairs (12 doubles per node = 4 starting banks = 4-way conflicts on every write of phase 1)
  // P.fvec_atomic: one f row per owned node, added to with LDS atomics, instead of one row per (node, incident element)
  // summed in order afterwards -- a quarter of the buffer, so twice the workgroups fit a CU; not bitwise repeatable
  // (FA_DETERMINISTIC keeps the ordered form)
  const bool atom = P.fvec_atomic != 0;
  const int SB = atom ? NDOF : ((V * NDOF) | 1);
  double* sBuf = sExt + (ext_state ? 2 * n_all : 0);
  int* sNode = reinterpret_cast<int*>(sBuf + n_own * SB);
  uint16_t* sConn = reinterpret_cast<uint16_t*>(sNode + n_own);
  const int CS = P.conn_stride;                        // tconn entries per element (NPE, or 4 when padded)
  uint8_t* sVal = reinterpret_cast<uint8_t*>(sConn + n_el * CS);

  // ---- phase 0: stage (the first pass's element words and history values are requested before the barrier) ----
  uint32_t ew0 = 0, es0[EW];
#pragma unroll
  for (int q = 0; q < EW; q++) es0[q] = 0;
  if (tid < n_el) {
    if (pre) {
      ew0 = ew_pre;
#pragma unroll
      for (int q = 0; q < EW; q++) es0[q] = es_pre[q];
    } else {
      ew0 = (uint32_t)P.telems[elem_off + tid];
#pragma unroll
      for (int q = 0; q < EW; q++) es0[q] = P.eslot[(int64_t)(elem_off + tid) * EW + q];
    }
  }
  auto stage_ext = [&](int i, int64_t n) {
    if constexpr (PF) {
      const int64_t dp = n * NDOF + RANK;
      if (SP::model(P) == 2 || SP::model(P) == 3) { sExt[2 * i] = P.state0[dp]; sExt[2 * i + 1] = P.stateX[dp]; }
      else if ((SP::model(P) == 5 || SP::model(P) == 6)) { sExt[2 * i] = 0.0; sExt[2 * i + 1] = P.stateX[dp]; }
    }
  };
  if (pre && tid < n_all) {
    if (tid < n_own) { sNode[tid] = (int)n_pre; sVal[tid] = (uint8_t)nv_pre; }
    double* nd = sNod + tid * NODW;
#pragma unroll
    for (int d = 0; d < RANK; d++) { nd[d] = pre_x[d]; nd[RANK + d] = pre_u[d]; }
    if constexpr (PF) nd[2 * RANK] = pre_u[RANK];
    stage_ext(tid, n_pre);
  }
  for (int i = pre ? tid + BLK : tid; i < n_all; i += BLK) {
    const int64_t n = P.tnodes[node_off + i];
    if (i < n_own) { sNode[i] = (int)n; sVal[i] = (uint8_t)P.nval[node_off + i]; }
    double* nd = sNod + i * NODW;
#pragma unroll
    for (int d = 0; d < RANK; d++) { nd[d] = P.coords[n * RANK + d]; nd[RANK + d] = P.state[n * NDOF + d]; }
    if constexpr (PF) nd[2 * RANK] = P.state[n * NDOF + RANK];
    stage_ext(i, n);
  }
  for (int i = tid; i < n_el * CS; i += BLK) sConn[i] = P.tconn[(int64_t)elem_off * CS + i];
  if (atom) for (int i = tid; i < n_own * NDOF; i += BLK) sBuf[i] = 0.0;
  const bool hist_model = PF && (SP::model(P) == 1 || SP::model(P) >= 4);
  double H0v[NIP];
#pragma unroll
  for (int k = 0; k < NIP; k++) H0v[k] = 0.0;
  if (hist_model && tid < n_el) {
    if (pre) {
#pragma unroll
      for (int k = 0; k < NIP; k++) H0v[k] = pre_H[k];
    } else {
      const int64_t e = (int64_t)(ew0 & 0x7FFFFFFFu);
#pragma unroll
      for (int k = 0; k < NIP; k++) H0v[k] = P.hist_pre[e * NIP + k];
    }
  }
  __syncthreads();

  // ---- phase 1: element force vectors ----
  double err_acc = 0.0;
  // constant address space: the tables are never written, so the loads stay scalar (s_load) even after the kernel's
  // own global stores -- as plain global pointers they turn into vector loads inside the Gauss-point loop
  const const_dptr tabN = (const_dptr)P.tab;
  const const_dptr tabD = tabN + NIP * NPE;
  const const_dptr tabW = tabN + NIP * NPE * (1 + RANK);
  for (int t = tid; t < n_el; t += BLK) {
    uint32_t ew = ew0, es[EW];
#pragma unroll
    for (int q = 0; q < EW; q++) es[q] = es0[q];
    if (t != tid) {                                    // later passes (tiles of more than BLK elements)
      ew = (uint32_t)P.telems[elem_off + t];
#pragma unroll
      for (int q = 0; q < EW; q++) es[q] = P.eslot[(int64_t)(elem_off + t) * EW + q];
      if (hist_model) {
#pragma unroll
        for (int k = 0; k < NIP; k++) H0v[k] = P.hist_pre[(int64_t)(ew & 0x7FFFFFFFu) * NIP + k];
      }
    }
    const int64_t e = (int64_t)(ew & 0x7FFFFFFFu);
    const bool write = (ew >> 31) != 0;                // this tile stores the element's history / strain / errExt_
    const uint16_t* lc = sConn + t * CS;
    int li[NPE];
#pragma unroll
    for (int a = 0; a < NPE; a++) li[a] = (int)lc[a];
    EmatAcc<SH, FAM, false, false> A;
    A.zero();
    if constexpr (RANK == 3) {
      // 3-D shapes (Hex8): no gathered element data and no nodal gradients in registers.  One pass over the nodes gives
      // J = sum x_a (x) dN_a, F = sum u_a (x) dN_a and sum phi_a dN_a in reference space; strain = sym(F J^-1),
      // grad phi = J^-T (.), and the rows are f1_a = dN_a . T with T = w J^-1 sigma, f2_a = w N_a cf + dN_a . G with
      // G = c2 w J^-1 grad phi -- the same expressions as the Gauss-point lanes of the tile kernel (fa_tile_impl.h).
      constexpr int S = SH::S, STR = SH::STR;
#pragma unroll 1
      for (int ip = 0; ip < NIP; ip++) {
        asm volatile("" ::: "memory");
        double J[RANK][RANK], Fm[RANK][RANK], gref[RANK], pf = 0.0, pf0 = 0.0, pfX = 0.0;
#pragma unroll
        for (int i = 0; i < RANK; i++) {
          gref[i] = 0.0;
#pragma unroll
          for (int k = 0; k < RANK; k++) { J[i][k] = 0.0; Fm[i][k] = 0.0; }
        }
#pragma unroll
        for (int a = 0; a < NPE; a++) {
          const double* nd = sNod + li[a] * NODW;
#pragma unroll
          for (int k = 0; k < RANK; k++) {
            const double dn = tabD[(ip * NPE + a) * RANK + k];
#pragma unroll
            for (int i = 0; i < RANK; i++) { J[i][k] += nd[i] * dn; Fm[i][k] += nd[RANK + i] * dn; }
            if constexpr (PF) gref[k] += nd[2 * RANK] * dn;
          }
          if constexpr (PF) {
            const double Na = tabN[ip * NPE + a];
            pf += Na * nd[2 * RANK];
            if (SP::model(P) == 2 || SP::model(P) == 3) { pf0 += Na * sExt[2 * li[a]]; pfX += Na * sExt[2 * li[a] + 1]; }
            else if (SP::model(P) == 5 || SP::model(P) == 6) pfX += Na * sExt[2 * li[a] + 1];
          }
        }
        double inv[RANK][RANK];
        const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
        const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
        const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
        const double det = J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02;
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
        const double w = tabW[ip] * det;
        double Hm[RANK][RANK], eps[S], gphi[RANK];
#pragma unroll
        for (int i = 0; i < RANK; i++)
#pragma unroll
          for (int j = 0; j < RANK; j++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < RANK; k++) acc += Fm[i][k] * inv[k][j];
            Hm[i][j] = acc;
          }
        eps[0] = Hm[0][0]; eps[1] = Hm[1][1]; eps[2] = Hm[2][2];           // ShapeUtils.cpp:114-145: xx yy zz xy yz zx
        eps[3] = Hm[0][1] + Hm[1][0]; eps[4] = Hm[1][2] + Hm[2][1]; eps[5] = Hm[2][0] + Hm[0][2];
#pragma unroll
        for (int i = 0; i < RANK; i++) {
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < RANK; k++) acc += gref[k] * inv[k][i];
          gphi[i] = acc;
        }
        const int64_t ipoint = e * NIP + ip;
        if (P.strain != nullptr && write) {
          double* sc = P.strain + ipoint * (PF ? STR + 1 : STR);
#pragma unroll
          for (int c = 0; c < S; c++) sc[c] = eps[c];
        }
        double H0 = 0.0;
#pragma unroll
        for (int k = 0; k < NIP; k++) if (k == ip) H0 = H0v[k];
        IpMat<SH, FAM, false> m;
        ip_material<SH, FAM, false, SP>(P, ipoint, eps, pf, pf0, pfX, H0, write, m, err_acc);
        double sm[RANK][RANK], T[RANK][RANK], G[RANK];
        sm[0][0] = m.sig[0]; sm[1][1] = m.sig[1]; sm[2][2] = m.sig[2];
        sm[0][1] = sm[1][0] = m.sig[3]; sm[1][2] = sm[2][1] = m.sig[4]; sm[0][2] = sm[2][0] = m.sig[5];
#pragma unroll
        for (int k = 0; k < RANK; k++) {
#pragma unroll
          for (int i = 0; i < RANK; i++) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < RANK; j++) acc += inv[k][j] * sm[j][i];
            T[k][i] = w * acc;
          }
          if constexpr (PF) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < RANK; j++) acc += inv[k][j] * gphi[j];
            G[k] = (P.c2 * w) * acc;
          }
        }
        const double wcf = PF ? w * m.cf : 0.0;
#pragma unroll
        for (int a = 0; a < NPE; a++) {
          double dn[RANK];
#pragma unroll
          for (int k = 0; k < RANK; k++) dn[k] = tabD[(ip * NPE + a) * RANK + k];
#pragma unroll
          for (int i = 0; i < RANK; i++) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < RANK; k++) acc += dn[k] * T[k][i];
            A.fu[a][i] += acc;
          }
          if constexpr (PF) {
            double acc = wcf * tabN[ip * NPE + a];
#pragma unroll
            for (int k = 0; k < RANK; k++) acc += dn[k] * G[k];
            A.fp[a] += acc;
          }
        }
      }
    } else {
    // elastic passes: the coordinates stay in registers across the points (cfg2 mesh at 16 M elements 0.836 -> 0.735 ms; the
    // u-phi kernel gains nothing, 1.207 -> 1.191, and the general-tangent one spills at its three-wave cap, 2.34 -> 2.65)
    constexpr bool KEEP_X = (FAM == FAM_LE);
    double X[KEEP_X ? NPE : 1][RANK];
    if constexpr (KEEP_X) {
#pragma unroll
      for (int a = 0; a < NPE; a++)
#pragma unroll
        for (int i = 0; i < RANK; i++) X[a][i] = sNod[li[a] * NODW + i];
    }
#pragma unroll 1
    for (int ip = 0; ip < NIP; ip++) {
      // the nodal values are re-read from LDS for every point: 40 registers less than keeping them across the loop
      // (the asm statement keeps the compiler from hoisting the reads)
      asm volatile("" ::: "memory");
      ElemData<SH, PF> E;
#pragma unroll
      for (int a = 0; a < NPE; a++) {
        const double* nd = sNod + li[a] * NODW;
#pragma unroll
        for (int i = 0; i < RANK; i++) { E.x[a][i] = KEEP_X ? X[KEEP_X ? a : 0][i] : nd[i]; E.u[a][i] = nd[RANK + i]; }
        if constexpr (PF) {
          E.phi[a] = nd[2 * RANK];
          if (ext_state) { E.phi0[a] = sExt[2 * li[a]]; E.phiX[a] = sExt[2 * li[a] + 1]; }
          else { E.phi0[a] = 0.0; E.phiX[a] = 0.0; }
        }
      }
      double dn[NPE][RANK], Nv[NPE];
#pragma unroll
      for (int a = 0; a < NPE; a++) {
        Nv[a] = tabN[ip * NPE + a];
#pragma unroll
        for (int d = 0; d < RANK; d++) dn[a][d] = tabD[(ip * NPE + a) * RANK + d];
      }
      double H0 = 0.0;
#pragma unroll
      for (int k = 0; k < NIP; k++) if (k == ip) H0 = H0v[k];
      IpRec<SH, FAM, false> r;
      ip_point_core<SH, FAM, false, false, SP>(P, e, ip, dn, Nv, tabW[ip], E, H0, write, r, err_acc);
      accum_emat<SH, FAM, false, false>(P, r, Nv, A);
    }
    }
#pragma unroll
    for (int a = 0; a < NPE; a++) {
      if (li[a] < n_own) {
        if (atom) {
          double* dst = sBuf + li[a] * NDOF;
#pragma unroll
          for (int i = 0; i < RANK; i++) atomicAdd(&dst[i], A.fu[a][i]);
          if constexpr (PF) atomicAdd(&dst[RANK], A.fp[a]);
        } else {
          double* dst = sBuf + li[a] * SB + (int)((es[a >> 2] >> (8 * (a & 3))) & 0xFFu) * NDOF;
#pragma unroll
          for (int i = 0; i < RANK; i++) dst[i] = A.fu[a][i];
          if constexpr (PF) dst[RANK] = A.fp[a];
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 2: f_int rows of the owned nodes (:748-749), summed in element order ----
  for (int i = tid; i < n_own * NDOF; i += BLK) {
    const int ia = i / NDOF, c = i - ia * NDOF;
    const int val = atom ? 1 : (int)sVal[ia];
    const double* src = sBuf + ia * SB + c;
    double s = 0.0;
    for (int k = 0; k < val; k++) s += src[k * NDOF];
    const int64_t q = (int64_t)sNode[ia] * NDOF + c;
    if (P.f_atomic) atomicAdd(&P.f[q], s);             // element tiles (fa_tiles.h): a node's row is summed over several tiles
    else P.f[q] = P.overwrite ? s : P.f[q] + s;
  }
  if (PF && P.err_part != nullptr) reduce_err<BLK>(err_acc, sErr, &P.err_part[tile]);
}

// ---- LDS-DMA staging shared by the persistent kernels ----
// global_load_lds: every lane gives its own source address, the destination is wave-uniform base + lane x size.
// Issued as inline asm: the compiler then keeps no book on it and never drains it early (with the builtin it waits
// vmcnt(0) in front of LDS reads it cannot tell apart from the DMA's destination, i.e. at the start of the very compute
// phase the transfer is meant to hide behind).  The kernel waits for its DMA itself: `s_waitcnt vmcnt(0)` in front of
// the barrier that precedes the first read of a staging buffer (dma_wait_all).  The compiler's own vmcnt waits for
// ordinary loads stay correct -- the memory counter retires in order, extra entries only make a wait longer.
// M0 (the destination base) is written and restored inside the statement.
typedef __attribute__((address_space(3))) char* lds_cptr;
template <int SIZE>
__device__ __forceinline__ void glds(const void* src, void* lds_wave_base) {
  // LDS byte address, the same in every lane of the wave: into a scalar register by hand (an "s" constraint alone does
  // not always legalise a value the compiler believes to be per-lane)
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(lds_cptr)lds_wave_base);
  unsigned keep;
  if constexpr (SIZE == 16)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  else if constexpr (SIZE == 12)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  else {
    static_assert(SIZE == 4, "LDS-DMA widths used here: 4, 12, 16 bytes");
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  }
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// byte offsets of the arrays of one staging buffer (MAXN nodes, MAXE elements, both multiples of 64)
template <class SH, bool PF>
struct StageLayout {
  static constexpr int RANK = SH::RANK, NIP = SH::NIP;
  static_assert(RANK == 2 && (NIP == 4 || NIP == 1) && SH::NPE <= 4, "staging pieces are sized for Quad4 / Triangle3");
  static constexpr int HP = (NIP == 4) ? 16 : 4;         // bytes of one history piece (two pieces per element)
  int oXL, oSA, oSB, oNV, oNID, oH0, oH1, oC0, oC1, oEW, oES, bytes;
  __host__ __device__ StageLayout(int MAXN, int MAXE, bool hist) {
    // a 12-byte DMA piece lands with a 16-byte lane stride (measured: scripts/calib/glds_layout.hip), so the two halves
    // of a 24-byte state record take 16 bytes per node each
    oXL = 0; oSA = oXL + MAXN * 16; oSB = oSA + MAXN * 16; oNV = oSB + (PF ? MAXN * 16 : 0);
    oNID = oNV + MAXN * 4; oH0 = oNID + MAXN * 4; oH1 = oH0 + (hist ? MAXE * HP : 0); oC0 = oH1 + (hist ? MAXE * HP : 0);
    oC1 = oC0 + MAXE * 4; oEW = oC1 + MAXE * 4; oES = oEW + MAXE * 4; bytes = oES + MAXE * 4;
  }
};

// Persistent form of assemble_fvec: a workgroup walks a sequence of tiles.  While it computes tile i the nodal values,
// history values and index words of tile i+1 travel HBM -> LDS by LDS-DMA (no registers), and the node / element ids of
// tile i+2 (which the DMA addresses of the next round need) by ordinary loads: the dependent chain tile descriptor ->
// ids -> gathers never stalls a compute phase.  PhaseFracture / MicroPhaseFracture / LinearElasticity (the Ext
// variants read two more state vectors and keep the plain kernel).
template <class SH, int FAM, class SP, int BLK>
__global__ __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(3))) void assemble_fvec_p(const Params P) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, NIP = SH::NIP;
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? RANK + 1 : RANK;
  extern __shared__ double smem[];
  char* lds = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, wv64 = (tid >> 6) * 64;
  const int MAXN = P.st_maxn, MAXE = P.st_maxe, V = P.fvec_val;
  const bool hist_model = PF && (SP::model(P) == 1 || SP::model(P) >= 4);
  const StageLayout<SH, PF> L(MAXN, MAXE, hist_model);
  const int SB = (V * NDOF) | 1;                       // see assemble_fvec
  double* sBuf = reinterpret_cast<double*>(lds + 2 * L.bytes);     // [n_own][V][NDOF] + pad

  // tiles of this workgroup: XCD x = blockIdx % 8 owns one contiguous band of tiles
  const int nt = P.ntile3, perT = (nt + 7) >> 3, stride = (int)gridDim.x >> 3;
  const int band0 = ((int)blockIdx.x & 7) * perT, band1 = min(band0 + perT, nt);
  int t_cur = band0 + ((int)blockIdx.x >> 3);
  if (t_cur >= band1) return;

  struct Ids { int n0, n1; uint32_t ew; int node_off, n_all, n_own, elem_off, n_el; };
  auto fetch_ids = [&](int tile) -> Ids {
    Ids d{};
    if (tile >= band1) return d;
    const const_iptr ti = (const_iptr)P.tinfo + (int64_t)tile * kTileInfo;      // scalar loads: no vmcnt wait behind the DMA
    d.node_off = ti[0]; d.n_all = ti[1]; d.n_own = ti[2]; d.elem_off = ti[3]; d.n_el = ti[4];
    if (tid < d.n_all) d.n0 = P.tnodes[d.node_off + tid];
    if (tid + BLK < d.n_all) d.n1 = P.tnodes[d.node_off + tid + BLK];
    if (tid < d.n_el) d.ew = (uint32_t)P.telems[d.elem_off + tid];
    return d;
  };
  auto issue = [&](const Ids& d, int buf) {
    char* B = lds + buf * L.bytes;
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const int i = tid + pass * BLK, wb = pass * BLK + wv64;
      const int64_t n = pass ? d.n1 : d.n0;
      if (i < d.n_all) {
        glds<16>(P.coords + n * RANK, B + L.oXL + wb * 16);
        if constexpr (PF) {
          glds<12>(P.state + n * NDOF, B + L.oSA + wb * 16);
          glds<12>(reinterpret_cast<const char*>(P.state + n * NDOF) + 12, B + L.oSB + wb * 16);
        } else {
          glds<16>(P.state + n * NDOF, B + L.oSA + wb * 16);
        }
        glds<4>(P.nval + d.node_off + i, B + L.oNV + wb * 4);
        reinterpret_cast<int*>(B + L.oNID)[i] = (int)n;
      }
    }
    if (tid < d.n_el) {
      const int64_t e = (int64_t)(d.ew & 0x7FFFFFFFu);
      if (hist_model) {
        const char* hp = reinterpret_cast<const char*>(P.hist_pre + e * NIP);
        constexpr int HP = StageLayout<SH, PF>::HP;
        glds<HP>(hp, B + L.oH0 + wv64 * HP);
        glds<HP>(hp + HP, B + L.oH1 + wv64 * HP);
      }
      const char* cp = reinterpret_cast<const char*>(P.tconn + ((int64_t)d.elem_off + tid) * 4);   // padded rows: 8 bytes
      glds<4>(cp, B + L.oC0 + wv64 * 4);
      glds<4>(cp + 4, B + L.oC1 + wv64 * 4);
      glds<4>(P.eslot + d.elem_off + tid, B + L.oES + wv64 * 4);
      reinterpret_cast<uint32_t*>(B + L.oEW)[tid] = d.ew;
    }
  };

  // constant address space: the tables are never written, so the loads stay scalar (s_load) even after the kernel's
  // own global stores -- as plain global pointers they turn into vector loads inside the Gauss-point loop
  const const_dptr tabN = (const_dptr)P.tab;
  const const_dptr tabD = tabN + NIP * NPE;
  const const_dptr tabW = tabN + NIP * NPE * (1 + RANK);
  Ids cur = fetch_ids(t_cur);
  issue(cur, 0);
  int t_nxt = t_cur + stride;
  Ids nxt = fetch_ids(t_nxt);
  int b = 0;
  double err_acc = 0.0;
  for (;;) {
    dma_wait_all();                                    // this wave's DMA of tile `cur` (issued a whole tile ago) has landed ...
    __syncthreads();                                   // ... and everybody's; the previous tile's phase 2 is over
    // the ids fetched a tile ago are "used" here, before any DMA is issued: the compiler then places its vmcnt wait for
    // them at this point and not (as a guard against overwriting a register with a load in flight) behind the DMA
    asm volatile("" : "+v"(nxt.n0), "+v"(nxt.n1), "+v"(nxt.ew));
    if (t_nxt < band1) issue(nxt, b ^ 1);
    const int t_nn = t_nxt + stride;
    const Ids nn = fetch_ids(t_nn);
    const char* B = lds + b * L.bytes;

    // ---- phase 1: element force vectors ----
    if (tid < cur.n_el) {
      const uint32_t ew = reinterpret_cast<const uint32_t*>(B + L.oEW)[tid];
      const uint32_t es = reinterpret_cast<const uint32_t*>(B + L.oES)[tid];
      const uint32_t c0 = reinterpret_cast<const uint32_t*>(B + L.oC0)[tid], c1 = reinterpret_cast<const uint32_t*>(B + L.oC1)[tid];
      const int64_t e = (int64_t)(ew & 0x7FFFFFFFu);
      const bool write = (ew >> 31) != 0;
      int li[4] = { (int)(c0 & 0xFFFFu), (int)(c0 >> 16), (int)(c1 & 0xFFFFu), (int)(c1 >> 16) };
      double H0v[NIP];
#pragma unroll
      for (int k = 0; k < NIP; k++) H0v[k] = 0.0;
      if (hist_model) {
        if constexpr (NIP == 4) {
          const double2 ha = reinterpret_cast<const double2*>(B + L.oH0)[tid], hb = reinterpret_cast<const double2*>(B + L.oH1)[tid];
          H0v[0] = ha.x; H0v[1] = ha.y; H0v[2] = hb.x; H0v[3] = hb.y;
        } else {
          H0v[0] = __hiloint2double(reinterpret_cast<const int*>(B + L.oH1)[tid], reinterpret_cast<const int*>(B + L.oH0)[tid]);
        }
      }
      EmatAcc<SH, FAM, false, false> A;
      A.zero();
#pragma unroll 1
      for (int ip = 0; ip < NIP; ip++) {
        asm volatile("" ::: "memory");                 // nodal values re-read per point (see assemble_fvec)
        ElemData<SH, PF> E;
#pragma unroll
        for (int a = 0; a < NPE; a++) {
          const double2 x = reinterpret_cast<const double2*>(B + L.oXL)[li[a]];
          E.x[a][0] = x.x; E.x[a][1] = x.y;
          if constexpr (PF) {
            const int* pa = reinterpret_cast<const int*>(B + L.oSA) + li[a] * 4;
            const int* pb = reinterpret_cast<const int*>(B + L.oSB) + li[a] * 4;
            E.u[a][0] = __hiloint2double(pa[1], pa[0]); E.u[a][1] = __hiloint2double(pb[0], pa[2]);
            E.phi[a] = __hiloint2double(pb[2], pb[1]); E.phi0[a] = 0.0; E.phiX[a] = 0.0;
          } else {
            const double2 u = reinterpret_cast<const double2*>(B + L.oSA)[li[a]];
            E.u[a][0] = u.x; E.u[a][1] = u.y;
          }
        }
        double dn[NPE][RANK], Nv[NPE];
#pragma unroll
        for (int a = 0; a < NPE; a++) {
          Nv[a] = tabN[ip * NPE + a];
#pragma unroll
          for (int d = 0; d < RANK; d++) dn[a][d] = tabD[(ip * NPE + a) * RANK + d];
        }
        double H0 = 0.0;
#pragma unroll
        for (int k = 0; k < NIP; k++) if (k == ip) H0 = H0v[k];
        IpRec<SH, FAM, false> r;
        ip_point_core<SH, FAM, false, false, SP>(P, e, ip, dn, Nv, tabW[ip], E, H0, write, r, err_acc);
        accum_emat<SH, FAM, false, false>(P, r, Nv, A);
      }
#pragma unroll
      for (int a = 0; a < NPE; a++) {
        if (li[a] < cur.n_own) {
          double* dst = sBuf + li[a] * SB + (int)((es >> (8 * a)) & 0xFFu) * NDOF;
#pragma unroll
          for (int i = 0; i < RANK; i++) dst[i] = A.fu[a][i];
          if constexpr (PF) dst[RANK] = A.fp[a];
        }
      }
    }
    __syncthreads();

    // ---- phase 2: f_int rows of the owned nodes, summed in element order ----
    for (int i = tid; i < cur.n_own * NDOF; i += BLK) {
      const int ia = i / NDOF, c = i - ia * NDOF;
      const int val = reinterpret_cast<const int*>(B + L.oNV)[ia];
      const double* src = sBuf + ia * SB + c;
      double s = 0.0;
      for (int k = 0; k < val; k++) s += src[k * NDOF];
      const int64_t q = (int64_t)reinterpret_cast<const int*>(B + L.oNID)[ia] * NDOF + c;
      P.f[q] = P.overwrite ? s : P.f[q] + s;
    }
    if (t_nxt >= band1) break;
    cur = nxt; nxt = nn; t_cur = t_nxt; t_nxt = t_nn; b ^= 1;
  }
}

template <class SH, int FAM, class SP>
static int launch_fvec_one(const Params& P, int grid, hipStream_t st) {
  const size_t shmem = (size_t)P.tile_lds;
  auto go = [&](auto kern, int blk) -> int {
    if (shmem > 65536 - 64) {
      hipError_t rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (rc != hipSuccess) return (int)rc;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(blk), shmem, st, P);
    return (int)hipGetLastError();
  };
  if constexpr (SH::NPE <= 4) {
    if (P.persistent) {
      if (P.tile_block == 192) return go(assemble_fvec_p<SH, FAM, SP, 192>, 192);
      if (P.tile_block == 256) return go(assemble_fvec_p<SH, FAM, SP, 256>, 256);
      return -1000;
    }
  } else if (P.persistent) return -1000;
  if (P.tile_block == 320) return go(assemble_fvec<SH, FAM, SP, 320>, 320);
  if (P.tile_block == 192) return go(assemble_fvec<SH, FAM, SP, 192>, 192);
  if (P.tile_block == 256) return go(assemble_fvec<SH, FAM, SP, 256>, 256);
  if (P.tile_block == 128) return go(assemble_fvec<SH, FAM, SP, 128>, 128);
  return -1000;
}

template <class SH>
static int launch_fvec_shape(const Params& P, int fam, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (fam == FAM_LE) return launch_fvec_one<SH, FAM_LE, SpecFixed<0, 0, 0>>(P, grid, st);
  if (fam == FAM_PF_ISO) {
    if (P.model == 1 && P.ptype == 0 && P.material == 1) return launch_fvec_one<SH, FAM_PF_ISO, SpecFixed<1, 0, 1>>(P, grid, st);
    if (P.model == 7) return gd_dispatch<SpecCanon>(P, [&](auto spec) -> int { return launch_fvec_one<SH, FAM_PF_ISO, decltype(spec)>(P, grid, st); });
    return launch_fvec_one<SH, FAM_PF_ISO, SpecCanon>(P, grid, st);
  }
  if (fam == FAM_PF_MIEHE) {
    if (is_micro_amor(P)) return launch_fvec_one<SH, FAM_PF_MIEHE, SpecMA<SpecCanon>>(P, grid, st);
    return launch_fvec_one<SH, FAM_PF_MIEHE, SpecCanon>(P, grid, st);
  }
  return -1000;
}

constexpr int kEmatBlock = 256;

template <class SH, int FAM, bool WANT_K, bool KEEP, class SP, int LPE>
static int launch_emat_geom(const Params& P, int grid, hipStream_t st) {
  if constexpr (SH::NIP % LPE != 0) return -1000;
  else {
    auto kern = assemble_emat<SH, FAM, WANT_K, KEEP, SP, kEmatBlock, LPE>;
    const size_t shmem = (size_t)P.tile_lds;
    if (shmem > 65536 - 64) {                            // per device and per context size: set on every launch (microseconds)
      hipError_t rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
      if (rc != hipSuccess) return (int)rc;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kEmatBlock), shmem, st, P);
    return (int)hipGetLastError();
  }
}

template <class SH, int FAM, bool WANT_K, bool KEEP, class SP>
static int launch_emat_one(const Params& P, int grid, hipStream_t st) {
  if (P.tile_block != kEmatBlock) return -1000;
  if (P.emat_lpe == 2) return launch_emat_geom<SH, FAM, WANT_K, KEEP, SP, 2>(P, grid, st);
  return launch_emat_geom<SH, FAM, WANT_K, KEEP, SP, 1>(P, grid, st);
}

template <class SH>
static int launch_emat_shape(const Params& P, int fam, bool want_K, bool keep, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (fam == FAM_LE)
    return want_K ? launch_emat_one<SH, FAM_LE, true, false, SpecFixed<0, 0, 0>>(P, grid, st)
                  : launch_emat_one<SH, FAM_LE, false, false, SpecFixed<0, 0, 0>>(P, grid, st);
  if (fam == FAM_PF_ISO) {
    const bool fixed = P.model == 1 && P.ptype == 0 && P.material == 1;    // PhaseFracture, p = 2, AmorPhase folded in
    if (!want_K) return fixed ? launch_emat_one<SH, FAM_PF_ISO, false, false, SpecFixed<1, 0, 1>>(P, grid, st)
                              : launch_emat_one<SH, FAM_PF_ISO, false, false, SpecCanon>(P, grid, st);
    if (keep) return launch_emat_one<SH, FAM_PF_ISO, true, true, SpecCanon>(P, grid, st);
    return fixed ? launch_emat_one<SH, FAM_PF_ISO, true, false, SpecFixed<1, 0, 1>>(P, grid, st)
                 : launch_emat_one<SH, FAM_PF_ISO, true, false, SpecCanon>(P, grid, st);
  }
  if (fam == FAM_PF_MIEHE) {
    if (!want_K) return launch_emat_one<SH, FAM_PF_MIEHE, false, false, SpecCanon>(P, grid, st);
    return keep ? launch_emat_one<SH, FAM_PF_MIEHE, true, true, SpecCanon>(P, grid, st)
                : launch_emat_one<SH, FAM_PF_MIEHE, true, false, SpecCanon>(P, grid, st);
  }
  return -1000;
}

}  // namespace fa
