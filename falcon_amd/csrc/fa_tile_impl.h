// fa_tile_impl.h -- tile kernel for Hex8 and the higher-order shapes.
//
// A tile is a compact set of owned nodes (a block of the spatially sorted node sequence, built in fa_api.hip)
// together with every element that touches one of them.  One workgroup per tile:
//   phase 0  stage    nodal coordinates / state of the tile's nodes, the shape tables and the local connectivity
//                     go to LDS
//   phase 1  points   one lane per (element, Gauss point): Jacobian, B u, degradation, material update, history
//                     (PhaseFractureModel.cpp:629-703).  The point leaves a short record in LDS -- J^-1, w det J and
//                     the constitutive coefficients -- and adds its f_int contributions for the owned nodes to an
//                     LDS image of the tile's f rows.  Each point of an element is evaluated once per tile that
//                     touches the element (1.5-2.5 times per pass instead of NPE times in the row kernel).
//   phase 2  pairs    one lane per (owned node a, neighbour b) -- per unordered pair when both nodes are owned --: the (rank+1)^2 block of K1..K4 summed over the
//                     elements that hold both nodes and over their Gauss points (:713-722), read from the LDS
//                     records, and written straight to its CSR entries.  Every entry of K has exactly one writer:
//                     no atomics, no LDS image of the rows, deterministic summation order (element-ascending).
//                     GET_INT_VECTOR passes (WANT_K = false) stop after phase 1 and keep no records.
// Lanes of phase 2 are sorted by their element count (8/4/2/1 for a structured Hex8 mesh) so that a wave does
// uniform work.  Registers stay low (16 accumulators), which is what the row kernel cannot offer in 3-D, where
// its 4 x 32 row block pins it at one wave per SIMD.
#pragma once
#include "fa_kernel_impl.h"

#ifndef FA_TILE_UNROLL_IPMAJ
#define FA_TILE_UNROLL_IPMAJ 1   /* 8 measured slower: cfg4 at 128^3 5.05 against 4.88 ms */
#endif
// Hex20 GET_INT_VECTOR passes, measured and left off (profiles/r03_batch29.txt, 64^3 Gauss3^3, 1.23 ms as is): point-major lanes
// as for Hex8 (FA_TILE_IPMAJ20=1) 1.29-1.33 ms, node loops unrolled by 4 / 20 (FA_TILE_UNROLL_F20) 1.27 / 1.33 ms, workgroups
// of 1024 (FA_TILE_FBLOCK) 1.35 ms -- the pass is bound by instruction issue (LDS reads + address arithmetic + FMAs), not by
// latency; what halves it is evaluating every point once (element tiles, fa_tiles.h)
#ifndef FA_TILE_ACC_TOUCH
#define FA_TILE_ACC_TOUCH 1
#endif
// FA_TILE_ACC_ATOMIC=1: K += of the ACC variants as atomic adds without return instead of load + add + store.  Measured
// (profiles/r03_batch34.txt): cfg4 at 128^3 40.3 against 8.3 ms -- 16 G scattered f64 atomics a second is all the L2s give.
#ifndef FA_TILE_ACC_ATOMIC
#define FA_TILE_ACC_ATOMIC 0
#endif
#ifndef FA_TILE_IPMAJ20
#define FA_TILE_IPMAJ20 0
#endif
#ifndef FA_TILE_UNROLL_F20
#define FA_TILE_UNROLL_F20 1
#endif
#ifndef FA_TILE_UNROLL_A
#define FA_TILE_UNROLL_A 1
#endif
#ifndef FA_TILE_UNROLL_IP
#define FA_TILE_UNROLL_IP 1
#endif

namespace fa {

template <class SH, int FAM, bool KEEP>
struct TileRec {                                        // offsets (doubles) inside one Gauss-point record
  static constexpr int R = SH::RANK, S = SH::S;
  static constexpr bool PF = (FAM != FAM_LE), GEN = (FAM == FAM_PF_MIEHE);
  static constexpr bool GREC = tile_grad_records(R, SH::NPE);           // gradients of all nodes instead of J^-1
  static constexpr int O_INV = 0, O_W = GREC ? SH::NPE * R : R * R, O_D = O_W + 1, NDC = GEN ? S * (S + 1) / 2 : (PF ? 3 : 0);   // Hooke: the tangent is a constant of the launch
  static constexpr int O_C1 = O_D + NDC, O_SP = O_C1 + (PF ? 1 : 0), W = O_SP + (KEEP ? S + 2 : 0);
  static_assert(W == tile_rec_doubles(R, FAM, KEEP, SH::NPE), "host and device disagree on the record width");
};

// ACC: the FA_ACCUMULATE variant of the GET_MATRIX0 kernels (the reference's `addBlock` +=, PhaseFractureModel.cpp:738-749).
// A block's old values are requested together -- all row chunks of the block in flight at once, one memory round trip per
// block instead of one per 32-byte chunk -- which costs registers the overwrite kernels must not pay (as a runtime branch
// it slowed them from 5.3 to 7.7 ms on cfg4 at 128^3), hence a compile-time variant.  ACC = false serves FA_OVERWRITE and,
// for the GET_INT_VECTOR passes, both modes.
// The 2-D general-tangent instantiations (129-136 VGPRs) sit a few registers above the 128 that two 512-thread workgroups
// per CU need, and ran one workgroup per CU because of them.  FA_TILE_CAP asks the compiler for four waves per SIMD there
// (second launch-bounds argument = waves per execution unit): 12-28 bytes of scratch per lane, measured Quad8 / MiehePhase
// 2.98 -> 2.45 ms, Triangle6 3.21 -> 2.15, Triangle3 Gauss3 3.46 -> 2.45 (the whole LDS for one workgroup instead: 2.91 /
// 2.73 / 3.18).  Not for the 3-D elastic kernels (Hex20 with Hooke, 134-136 VGPRs): capped, Hex20 Gauss2^3 at 64^3 went
// from 3.6 to 3.86 ms -- their grouped pair lanes lose more to the spills than the second workgroup brings.
#ifndef FA_TILE_CAP
#define FA_TILE_CAP 1
#endif
template <class SH, int FAM, int BLK>
constexpr int tile_min_waves() {
  return (FA_TILE_CAP && BLK == 512 && SH::RANK == 2 && FAM == FAM_PF_MIEHE) ? 4 : 1;
}
// global_load_lds_dword with a per-lane source address: 4 bytes per lane land at `lds_wave_base` + lane x 4 (wave-uniform base
// through M0, saved and restored inside the statement); issued as inline asm so that the compiler keeps no book on it (see
// glds in fa_emat_impl.h).  Used to pull the cache lines of a later read-modify-write into the L2 without a register.
typedef __attribute__((address_space(3))) char* tile_lds_cptr;
__device__ __forceinline__ void l2_touch(const void* src, void* lds_wave_base) {
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(tile_lds_cptr)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

template <class SH, int FAM, bool WANT_K, bool KEEP, class SP, int BLK, bool ACC = false>
__global__ __launch_bounds__(BLK, (tile_min_waves<SH, FAM, BLK>())) void assemble_tiles(const Params P) {
  constexpr int RANK = SH::RANK, NPE = SH::NPE, NIP = SH::NIP, S = SH::S, STR = SH::STR;
  constexpr bool PF = (FAM != FAM_LE);
  constexpr int NDOF = PF ? RANK + 1 : RANK;
  using TR = TileRec<SH, FAM, KEEP>;
  constexpr int RW = TR::W, SLOTW = tile_slot_doubles(NIP, RW), NODW = tile_node_doubles(RANK, PF);
  constexpr int TABN = NIP * NPE * (1 + RANK) + NIP;
  extern __shared__ double smem[];
  __shared__ double sErr[BLK / 64];

  const int tid = threadIdx.x;
#ifdef FA_TRACE
  long long trc[6]; trc[0] = __builtin_readcyclecounter();
#endif
  const int per = (int)gridDim.x >> 3;                 // XCD bands: neighbouring tiles share halo nodes in one L2
  const int tile0 = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (tile0 >= P.ntile3) return;
  const int tile = P.tile_list ? P.tile_list[tile0] : tile0;   // interface / interior part of a pass (fa_set_interface)
  const int32_t* __restrict__ ti = P.tinfo + (int64_t)tile * kTileInfo;
  // Node lists padded to a fixed stride per tile (P.tn_stride > 0): the address of a lane's first tile node depends on the
  // tile index only, so its id and then its coordinates / state / row offsets are requested before the descriptor has
  // arrived -- the staging chain descriptor -> node list -> nodal data loses one of its three memory round trips.
  // (Idle slots of the padded list hold node 0: a harmless read.)
  constexpr int NDOF0 = (FAM != FAM_LE) ? SH::RANK + 1 : SH::RANK;
  const bool pre = P.tn_stride > 0;
  int64_t n_pre = 0;
  double pre_x[SH::RANK], pre_u[NDOF0];
  long long pre_b0 = 0, pre_b1 = 0;
  if (pre) {
    n_pre = P.tnodes[(int64_t)tile * P.tn_stride + ((int)threadIdx.x < P.tn_stride ? (int)threadIdx.x : 0)];
#pragma unroll
    for (int d = 0; d < SH::RANK; d++) pre_x[d] = P.coords[n_pre * SH::RANK + d];
#pragma unroll
    for (int d = 0; d < NDOF0; d++) pre_u[d] = P.state[n_pre * NDOF0 + d];
    if constexpr (WANT_K) { pre_b0 = P.row_ptr[n_pre * NDOF0]; pre_b1 = P.row_ptr[n_pre * NDOF0 + 1]; }
  }
  const int node_off = ti[0], n_all = ti[1], n_own = ti[2], elem_off = ti[3], n_el = ti[4];
  const int pair_off = ti[5], n_pairs = ti[6], ent_off = ti[7];

  double* sRec = smem;
  double* sNod = sRec + (WANT_K ? n_el * SLOTW : 0);   // GET_INT_VECTOR passes keep no records
  double* sExt = sNod + n_all * NODW;                 // phi of state0 / state00 (stateExt_), Ext variants only
  const bool ext_state = PF && (SP::model(P) == 2 || SP::model(P) == 3 || (SP::model(P) == 5 || SP::model(P) == 6));
  double* sF   = sExt + (ext_state ? 2 * n_all : 0);
  const int n_rows = P.tile_sym ? n_all : n_own;      // cross-tile mirror lanes also write rows of halo nodes
  // grouped pair lanes (3-D elastic kernels): f image for every tile node -- the pair lanes add K_ba u_a for halo nodes b too
  const int n_f = (tile_group(RANK, FAM) > 1) ? n_all : n_own;
  long long* sRow = reinterpret_cast<long long*>(sF + n_f * NDOF);     // first CSR entry of every (owned) node's first row
  int* sDeg = reinterpret_cast<int*>(sRow + n_rows);                   // neighbours of every (owned) node
  double* sTab = reinterpret_cast<double*>(sRow + n_rows) + (n_rows + 1) / 2;
  uint16_t* sConn = (uint16_t*)(sTab + TABN);
  const double* tabN = sTab;
  const double* tabD = sTab + NIP * NPE;
  const double* tabW = sTab + NIP * NPE * (1 + RANK);

  // ---- phase 0: stage ----
  auto stage_ext = [&](int i, int64_t n) {
    if constexpr (PF) {
      const int64_t dp = n * NDOF + RANK;
      if (SP::model(P) == 2 || SP::model(P) == 3) { sExt[2 * i] = P.state0[dp]; sExt[2 * i + 1] = P.stateX[dp]; }
      else if ((SP::model(P) == 5 || SP::model(P) == 6)) { sExt[2 * i] = 0.0; sExt[2 * i + 1] = P.stateX[dp]; }
    }
  };
  if (pre && tid < n_all) {                             // the values requested in front of the descriptor
    double* nd = sNod + tid * NODW;
#pragma unroll
    for (int d = 0; d < RANK; d++) { nd[d] = pre_x[d]; nd[RANK + d] = pre_u[d]; }
    if constexpr (PF) nd[2 * RANK] = pre_u[RANK];
    stage_ext(tid, n_pre);
  }
  for (int i = pre ? tid + BLK : tid; i < n_all; i += BLK) {
    const int64_t n = P.tnodes[node_off + i];
    double* nd = sNod + i * NODW;
#pragma unroll
    for (int d = 0; d < RANK; d++) { nd[d] = P.coords[n * RANK + d]; nd[RANK + d] = P.state[n * NDOF + d]; }
    if constexpr (PF) nd[2 * RANK] = P.state[n * NDOF + RANK];
    stage_ext(i, n);
  }
  for (int i = tid; i < n_f * NDOF; i += BLK) sF[i] = 0.0;
  const int cnt0 = P.cnt[0];
  if (WANT_K && pre && tid < n_rows) { sRow[tid] = pre_b0; sDeg[tid] = (cnt0 > 0) ? (int)((pre_b1 - pre_b0) / cnt0) : 0; }
  for (int i = (WANT_K && pre) ? tid + BLK : tid; i < n_rows; i += BLK) {
    const int64_t r0 = (int64_t)P.tnodes[node_off + i] * NDOF;
    const long long b0 = P.row_ptr[r0], b1 = P.row_ptr[r0 + 1];
    sRow[i] = b0; sDeg[i] = (cnt0 > 0) ? (int)((b1 - b0) / cnt0) : 0;
  }
  for (int i = tid; i < TABN; i += BLK) sTab[i] = P.tab[i];
  for (int i = tid; i < n_el * NPE; i += BLK) sConn[i] = P.tconn[(int64_t)elem_off * NPE + i];
  __syncthreads();
#ifdef FA_TRACE
  trc[1] = __builtin_readcyclecounter();
#endif

  // Hooke with grouped pair lanes, GET_MATRIX0 passes without a strain cache: f_int = sum_b K_ab u_b is formed by the pair
  // lanes from the blocks they hold anyway (the elastic f_int IS K u: sigma = D B u), so the Gauss-point lanes skip
  // strain, stress and the f rows -- most of their work.  All f updates are then atomic adds to global memory (blocks
  // written for a neighbour tile's node add to that node's row); the host zeroes f first in overwrite mode.
  constexpr bool FU_CAP = WANT_K && !PF && tile_group(RANK, FAM) > 1 && !TR::GREC;
  const bool fu = FU_CAP && P.tile_fu != 0;
  // ---- phase 1: Gauss points ----
  double err_acc = 0.0;
  // Hex8: the lanes of a wave take ONE Gauss point of 64 different elements (point-major lane order) instead of the 8 points
  // of 8 elements: the point index is wave-uniform, so the shape tables come through scalar loads instead of per-lane LDS
  // reads, and the lanes of a wave add to the f rows of different nodes (on a structured mesh local node a of different
  // elements is never the same node)
  // (Hex20 GET_INT_VECTOR passes in the same order: FA_TILE_IPMAJ20, measured slower)
  constexpr bool IPMAJ = RANK == 3 && ((NPE == 8 && NIP == 8) || (FA_TILE_IPMAJ20 && NPE == 20 && (!WANT_K || FA_TILE_IPMAJ20 > 1)));
  const int nel_pad = (n_el + 63) & ~63;
  const const_dptr gN = (const_dptr)P.tab, gD = gN + NIP * NPE, gW = gN + NIP * NPE * (1 + RANK);
  auto TN = [&](int idx) -> double { if constexpr (IPMAJ) return gN[idx]; else return tabN[idx]; };
  auto TD = [&](int idx) -> double { if constexpr (IPMAJ) return gD[idx]; else return tabD[idx]; };
  auto TW = [&](int idx) -> double { if constexpr (IPMAJ) return gW[idx]; else return tabW[idx]; };
  for (int t = tid; t < (IPMAJ ? NIP * nel_pad : n_el * NIP); t += BLK) {
    int s, ip;
    if constexpr (IPMAJ) {
      ip = __builtin_amdgcn_readfirstlane(t / nel_pad); s = t - ip * nel_pad;
      if (s >= n_el) continue;
    } else { s = t / NIP; ip = t - s * NIP; }
    const uint32_t ew = (uint32_t)P.telems[elem_off + s];
    const int64_t e = (int64_t)(ew & 0x7FFFFFFFu);
    const bool write = (ew >> 31) != 0;                // this tile stores the element's history / strain / errExt_
    const uint16_t* lc = sConn + s * NPE;
    // One pass over the element's nodes: J = sum x_a (x) dN_a and, in the same reference-space form, F = sum u_a (x) dN_a,
    // sum phi_a dN_a and the interpolated phase fields.  The physical quantities follow from J^-1 afterwards
    // (du/dx = F J^-1, grad phi = J^-T sum phi_a dN_a): 27 + 9 multiply-adds once instead of 9 + 9 + 3 per node for
    // the physical gradient of every node.  (Not when the strain is not needed -- K-u force path -- or when the records
    // hold the nodal gradients.)
    constexpr bool FUSED = !TR::GREC;
    double J[RANK][RANK], Fm[RANK][RANK], gref[RANK], pf = 0.0, pf0 = 0.0, pfX = 0.0;
#pragma unroll
    for (int i = 0; i < RANK; i++) {
      gref[i] = 0.0;
#pragma unroll
      for (int k = 0; k < RANK; k++) { J[i][k] = 0.0; Fm[i][k] = 0.0; }
    }
    const bool lean = FU_CAP && fu;                      // only J is needed
    constexpr bool F20 = !WANT_K && NPE == 20;
#pragma unroll (F20 ? FA_TILE_UNROLL_F20 : (IPMAJ ? FA_TILE_UNROLL_IPMAJ : FA_TILE_UNROLL_A))
    for (int a = 0; a < NPE; a++) {
      const double* nd = sNod + (int)lc[a] * NODW;
#pragma unroll
      for (int k = 0; k < RANK; k++) {
        const double dn = TD((ip * NPE + a) * RANK + k);
#pragma unroll
        for (int i = 0; i < RANK; i++) J[i][k] += nd[i] * dn;
        if (FUSED && !lean) {
#pragma unroll
          for (int i = 0; i < RANK; i++) Fm[i][k] += nd[RANK + i] * dn;
          if constexpr (PF) gref[k] += nd[2 * RANK] * dn;
        }
      }
      if constexpr (FUSED && PF) {
        const double Na = TN(ip * NPE + a);
        pf += Na * nd[2 * RANK];
        if (SP::model(P) == 2 || SP::model(P) == 3) { pf0 += Na * sExt[2 * (int)lc[a]]; pfX += Na * sExt[2 * (int)lc[a] + 1]; }
        else if ((SP::model(P) == 5 || SP::model(P) == 6)) pfX += Na * sExt[2 * (int)lc[a] + 1];
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
    const double w = TW(ip) * det;
    if constexpr (FU_CAP) {
      if (fu) {                                        // f_int comes from the pair lanes (K u): the point only leaves J^-1 and w det J
        double* rec = sRec + s * SLOTW + ip * RW;
#pragma unroll
        for (int k = 0; k < RANK; k++)
#pragma unroll
          for (int i = 0; i < RANK; i++) rec[TR::O_INV + k * RANK + i] = inv[k][i];
        rec[TR::O_W] = w;
        continue;
      }
    }
    double eps[S], gphi[RANK];
#pragma unroll
    for (int c = 0; c < S; c++) eps[c] = 0.0;
#pragma unroll
    for (int i = 0; i < RANK; i++) gphi[i] = 0.0;
    if constexpr (FUSED) {
      double H[RANK][RANK];                              // H[i][j] = d u_i / d x_j
#pragma unroll
      for (int i = 0; i < RANK; i++)
#pragma unroll
        for (int j = 0; j < RANK; j++) {
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < RANK; k++) acc += Fm[i][k] * inv[k][j];
          H[i][j] = acc;
        }
      if constexpr (RANK == 2) {                         // strain = B u (ShapeUtils.cpp:73-104)
        eps[0] = H[0][0]; eps[1] = H[1][1]; eps[2] = H[0][1] + H[1][0];
      } else {                                           // ShapeUtils.cpp:114-145: xx yy zz xy yz zx
        eps[0] = H[0][0]; eps[1] = H[1][1]; eps[2] = H[2][2];
        eps[3] = H[0][1] + H[1][0]; eps[4] = H[1][2] + H[2][1]; eps[5] = H[2][0] + H[0][2];
      }
      if constexpr (PF) {
#pragma unroll
        for (int i = 0; i < RANK; i++) {
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < RANK; k++) acc += gref[k] * inv[k][i];
          gphi[i] = acc;
        }
      }
    } else
#pragma unroll FA_TILE_UNROLL_A
    for (int a = 0; a < NPE; a++) {
      const double* nd = sNod + (int)lc[a] * NODW;
      double ga[RANK], ua[RANK];
#pragma unroll
      for (int i = 0; i < RANK; i++) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < RANK; k++) acc += TD((ip * NPE + a) * RANK + k) * inv[k][i];
        ga[i] = acc; ua[i] = nd[RANK + i];
      }
      if constexpr (WANT_K && TR::GREC) {
#pragma unroll
        for (int i = 0; i < RANK; i++) sRec[s * SLOTW + ip * RW + TR::O_INV + a * RANK + i] = ga[i];
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
        const double Na = TN(ip * NPE + a), ph = nd[2 * RANK];
        pf += Na * ph;
#pragma unroll
        for (int i = 0; i < RANK; i++) gphi[i] += ga[i] * ph;
        if (SP::model(P) == 2 || SP::model(P) == 3) { pf0 += Na * sExt[2 * (int)lc[a]]; pfX += Na * sExt[2 * (int)lc[a] + 1]; }
        else if ((SP::model(P) == 5 || SP::model(P) == 6)) pfX += Na * sExt[2 * (int)lc[a] + 1];
      }
    }
    const int64_t ipoint = e * NIP + ip;               // initializeIPMPMap_, PhaseFractureModel.cpp:1070-1093
    if (P.strain != nullptr && write) {
      double* sc = P.strain + ipoint * (PF ? STR + 1 : STR);
      if constexpr (RANK == 2) { sc[0] = eps[0]; sc[1] = eps[1]; sc[2] = 0.0; sc[3] = eps[2]; }
      else {
#pragma unroll
        for (int c = 0; c < S; c++) sc[c] = eps[c];
      }
    }
    double H0 = 0.0;
    if constexpr (PF) {
      if (SP::model(P) == 1 || SP::model(P) >= 4) H0 = P.hist_pre[ipoint];
    }
    IpMat<SH, FAM, KEEP> m;
    ip_material<SH, FAM, KEEP, SP>(P, ipoint, eps, pf, pf0, pfX, H0, write, m, err_acc);

    if constexpr (WANT_K) {
    double* rec = sRec + s * SLOTW + ip * RW;
    if constexpr (!TR::GREC) {
#pragma unroll
      for (int k = 0; k < RANK; k++)
#pragma unroll
        for (int i = 0; i < RANK; i++) rec[TR::O_INV + k * RANK + i] = inv[k][i];
    }
    rec[TR::O_W] = w;
    if constexpr (FAM == FAM_PF_MIEHE) {
      int q = 0;
#pragma unroll
      for (int i = 0; i < S; i++)
#pragma unroll
        for (int j = i; j < S; j++) rec[TR::O_D + (q++)] = m.D[i][j];
    } else if constexpr (PF) {
      rec[TR::O_D] = m.d0; rec[TR::O_D + 1] = m.d1; rec[TR::O_D + 2] = m.d3;
    }
    if constexpr (PF) rec[TR::O_C1] = m.c1;
    if constexpr (KEEP) {
#pragma unroll
      for (int c = 0; c < S; c++) rec[TR::O_SP + c] = m.sp[c];
      rec[TR::O_SP + S] = m.dg; rec[TR::O_SP + S + 1] = m.dgl;
    }
    }

    // f1 += w B^T sigma, f2 (PhaseFractureModel.cpp:727-730) for the owned nodes.  The node loop starts at a
    // different node in every lane of an element so that its lanes do not add to one address at the same time.
    if constexpr (FUSED) {
      // f1_a = w B_a^T sigma = dN_a . T with T[k][i] = w sum_j J^-1[k][j] sigma_ij (27 multiply-adds once, 9 per owned
      // node); f2_a = w N_a cf + dN_a . G with G[k] = c2 w sum_j J^-1[k][j] grad phi_j
      double sm[RANK][RANK], T[RANK][RANK], G[RANK];
      if constexpr (RANK == 2) { sm[0][0] = m.sig[0]; sm[1][1] = m.sig[1]; sm[0][1] = sm[1][0] = m.sig[2]; }
      else {
        sm[0][0] = m.sig[0]; sm[1][1] = m.sig[1]; sm[2][2] = m.sig[2];
        sm[0][1] = sm[1][0] = m.sig[3]; sm[1][2] = sm[2][1] = m.sig[4]; sm[0][2] = sm[2][0] = m.sig[5];
      }
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
#pragma unroll (F20 ? FA_TILE_UNROLL_F20 : 1)
      for (int q = 0; q < NPE; q++) {
        int a = q + (ip % NPE); if (a >= NPE) a -= NPE;
        const int li = (int)lc[a];
        if (li >= n_own) continue;
        double dn[RANK];
#pragma unroll
        for (int k = 0; k < RANK; k++) dn[k] = TD((ip * NPE + a) * RANK + k);
#pragma unroll
        for (int i = 0; i < RANK; i++) {
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < RANK; k++) acc += dn[k] * T[k][i];
          atomicAdd(&sF[li * NDOF + i], acc);
        }
        if constexpr (PF) {
          double acc = wcf * TN(ip * NPE + a);
#pragma unroll
          for (int k = 0; k < RANK; k++) acc += dn[k] * G[k];
          atomicAdd(&sF[li * NDOF + RANK], acc);
        }
      }
    } else
#pragma unroll 1
    for (int q = 0; q < NPE; q++) {
      int a = q + (ip % NPE); if (a >= NPE) a -= NPE;
      const int li = (int)lc[a];
      if (li >= n_own) continue;
      double wga[RANK], tt[RANK];
#pragma unroll
      for (int i = 0; i < RANK; i++) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < RANK; k++) acc += TD((ip * NPE + a) * RANK + k) * inv[k][i];
        wga[i] = w * acc;
      }
      Bt_vec<RANK, S>(wga, m.sig, tt);
#pragma unroll
      for (int i = 0; i < RANK; i++) atomicAdd(&sF[li * NDOF + i], tt[i]);
      if constexpr (PF) {
        double gg = 0.0;
#pragma unroll
        for (int i = 0; i < RANK; i++) gg += wga[i] * gphi[i];
        atomicAdd(&sF[li * NDOF + RANK], (w * TN(ip * NPE + a)) * m.cf + P.c2 * gg);
      }
    }
  }
  __syncthreads();

#ifdef FA_TRACE
  trc[2] = __builtin_readcyclecounter();
#endif
  // ---- f_int rows of the owned nodes ----
  if (!fu)
    for (int i = tid; i < n_own * NDOF; i += BLK) {
      const int ia = i / NDOF;
      const int64_t q = (int64_t)P.tnodes[node_off + ia] * NDOF + (i - ia * NDOF);
      if (P.f_atomic) atomicAdd(&P.f[q], sF[i]);       // element tiles (fa_tiles.h): a node's row is summed over several tiles
      else P.f[q] = P.overwrite ? sF[i] : P.f[q] + sF[i];
    }

  // ---- phase 2: node-pair blocks of K1..K4 (PhaseFractureModel.cpp:713-722) ----
  if constexpr (WANT_K) {
#ifndef FA_TILE_TAB_GLOBAL
#define FA_TILE_TAB_GLOBAL 0
#endif
  // FA_TILE_TAB_GLOBAL=1 reads the shape tables of the pair lanes through the vector-memory path to take load off
  // the LDS: measured slower (Hex8 100^3 3.94 against 3.47 ms, Hex20 64^3 11.2 against 8.75 ms) -- divergent 8-byte loads.
  const double* __restrict__ tab2N = FA_TILE_TAB_GLOBAL ? P.tab : tabN;
  const double* __restrict__ tab2D = FA_TILE_TAB_GLOBAL ? P.tab + NIP * NPE : tabD;
  // wave-private staging rows for the coalesced write-out of row-ordered lanes (P.tile_stage)
  double* stg = reinterpret_cast<double*>(sConn + ((n_el * NPE + 3) & ~3)) + (tid >> 6) * (64 * NDOF + 64);
  long long* stg_base = reinterpret_cast<long long*>(stg + 64 * NDOF);
  const int lane = tid & 63;
  const uint4* __restrict__ prec = reinterpret_cast<const uint4*>(P.tpairs) + pair_off;
  uint4 pw_next = (tid < n_pairs) ? prec[tid] : make_uint4(0u, 0u, 0u, 0u);
  constexpr int GRP = tile_group(RANK, FAM);
  if constexpr (GRP > 1) {
    // ---- grouped pair lanes (TileOpts::group): one lane serves up to 4 neighbours b of its node a that share the same
    //      elements with a.  Per element and point: the record and grad N_a once, then per neighbour 3 table reads and
    //      18 multiply-adds (g_b and M_b += w g_a g_b^T).  Hooke only: K1 follows from M after the loops. ----
    static_assert(!PF && RANK == 3 && GRP <= 4, "grouped lanes: 3-D elastic kernels");
    for (int p0 = 0; p0 < n_pairs; p0 += BLK) {
      const int p = p0 + tid;
      const bool active = p < n_pairs;
      const uint4 pw = pw_next;
      pw_next = (p + BLK < n_pairs) ? prec[p + BLK] : make_uint4(0u, 0u, 0u, 0u);
      const int ia = (int)(pw.x & 0xFFFFu), nb = (int)((pw.x >> 16) & 0xFFu), cnt = (int)(pw.x >> 24);
      const uint32_t* __restrict__ ents = P.tents + (int64_t)ent_off + pw.y;
      double M[GRP][3][3];
#pragma unroll
      for (int j = 0; j < GRP; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int k = 0; k < 3; k++) M[j][i][k] = 0.0;
      uint32_t e0_next = pw.w, e1_next = cnt > 0 ? ents[nb + 1] : 0u;
#pragma unroll 1
      for (int t = 0; t < cnt; t++) {
        const uint32_t e0 = e0_next, e1 = e1_next;
        if (t + 1 < cnt) { e0_next = ents[nb + 2 * t + 2]; e1_next = ents[nb + 2 * t + 3]; }
        const int slot = (int)(e0 & 0xFFFFu), la = (int)((e0 >> 16) & 0xFFu);
#pragma unroll 1
        for (int ip = 0; ip < NIP; ip++) {
          const double* rec = sRec + slot * SLOTW + ip * RW;
          double inv[3][3];
#pragma unroll
          for (int k = 0; k < 3; k++)
#pragma unroll
            for (int i = 0; i < 3; i++) inv[k][i] = rec[TR::O_INV + k * 3 + i];
          const double w = rec[TR::O_W];
          double wga[3];
          {
            const double* da = tabD + (ip * NPE + la) * 3;
            const double d0 = da[0], d1 = da[1], d2 = da[2];
#pragma unroll
            for (int i = 0; i < 3; i++) wga[i] = w * (d0 * inv[0][i] + d1 * inv[1][i] + d2 * inv[2][i]);
          }
#pragma unroll
          for (int j = 0; j < GRP; j++) {
            if (j < nb) {
              const int lb = (int)((e1 >> (8 * j)) & 0xFFu);
              const double* db = tabD + (ip * NPE + lb) * 3;
              const double d0 = db[0], d1 = db[1], d2 = db[2];
#pragma unroll
              for (int k = 0; k < 3; k++) {
                const double gbk = d0 * inv[0][k] + d1 * inv[1][k] + d2 * inv[2][k];
#pragma unroll
                for (int i = 0; i < 3; i++) M[j][i][k] += wga[i] * gbk;
              }
            }
          }
        }
      }
      const int lg = (int)((pw.z >> 24) & 3u);
      if (__any(lg != 0)) {
        auto dpp = [](double v, auto ctl) {
          int lo = __double2loint(v), hi = __double2hiint(v);
          lo = __builtin_amdgcn_mov_dpp(lo, decltype(ctl)::value, 0xF, 0xF, true);
          hi = __builtin_amdgcn_mov_dpp(hi, decltype(ctl)::value, 0xF, 0xF, true);
          return __hiloint2double(hi, lo);
        };
#pragma unroll
        for (int j = 0; j < GRP; j++)
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
              double& v = M[j][i][k];
              const double p1 = dpp(v, std::integral_constant<int, 0xB1>());
              v += (lg >= 1) ? p1 : 0.0;
              const double p2 = dpp(v, std::integral_constant<int, 0x4E>());
              v += (lg >= 2) ? p2 : 0.0;
            }
      }
      if (!active || ((pw.z >> 26) & 3u) != 0) continue;
      const bool diag = (pw.z >> 30) & 1u;
      // the neighbour words are read here, not in front of the element loop: four registers less across it (scratch of the
      // 1024-thread Gauss3^3 kernel 20 -> 12 bytes, cfg5 35.7 -> 34.8 ms: profiles/r03_batch39.txt)
      uint32_t bw[GRP];
#pragma unroll
      for (int j = 0; j < GRP; j++) bw[j] = (j < nb) ? ents[j] : 0u;
#pragma unroll
      for (int j = 0; j < GRP; j++) {
        if (j >= nb) continue;
        const double tr = M[j][0][0] + M[j][1][1] + M[j][2][2];
        const int ib = (int)(bw[j] & 0xFFFFu), kpos2 = (int)((bw[j] >> 16) & 0xFFu), kpos = (int)(bw[j] >> 24);
        // canonical blocked layout, 3 dofs per node: row i of node io starts at sRow[io] + i * 3 * deg
#pragma unroll
        for (int i = 0; i < 3; i++) {
          double row[3], col[3];
#pragma unroll
          for (int k = 0; k < 3; k++) {
            row[k] = (i == k) ? (P.hd0 - P.hd3) * M[j][i][i] + P.hd3 * tr : P.hd1 * M[j][i][k] + P.hd3 * M[j][k][i];   // K_ab[i][k]
            col[k] = (i == k) ? row[k] : P.hd1 * M[j][k][i] + P.hd3 * M[j][i][k];                                        // K_ba[i][k] = K_ab[k][i]
          }
#ifdef FA_TILE_NOWRITE                                   // ablation build: no K stores (the blocks stay live through a never-true test)
          if (row[0] + row[1] + row[2] + col[0] + col[1] + col[2] == 1.2345e300) P.K[0] = 1.0;
#else
          // a 24-byte row chunk as one 16-byte and one 8-byte store (the address is 8-byte aligned only; global memory
          // takes unaligned vector stores)
          typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
          auto put3 = [&](long long b, const double (&v)[3]) {
            double* q = P.K + b;
            if (P.overwrite) { d2u t; t.x = v[0]; t.y = v[1]; *reinterpret_cast<d2u*>(q) = t; q[2] = v[2]; }
            else { const d2u o = *reinterpret_cast<const d2u*>(q); d2u t; t.x = o.x + v[0]; t.y = o.y + v[1]; const double o2 = q[2];
                   *reinterpret_cast<d2u*>(q) = t; q[2] = o2 + v[2]; }
          };
          if constexpr (ACC) {                           // both chunks' old values in flight together
            double* qa = P.K + (sRow[ia] + (long long)i * 3 * sDeg[ia] + (long long)kpos * 3);
            double* qb = diag ? qa : P.K + (sRow[ib] + (long long)i * 3 * sDeg[ib] + (long long)kpos2 * 3);
            const d2u oa = *reinterpret_cast<const d2u*>(qa); const double oa2 = qa[2];
            const d2u ob = *reinterpret_cast<const d2u*>(qb); const double ob2 = qb[2];
            d2u t; t.x = oa.x + row[0]; t.y = oa.y + row[1]; *reinterpret_cast<d2u*>(qa) = t; qa[2] = oa2 + row[2];
            if (!diag) { d2u u; u.x = ob.x + col[0]; u.y = ob.y + col[1]; *reinterpret_cast<d2u*>(qb) = u; qb[2] = ob2 + col[2]; }
          } else {
          put3(sRow[ia] + (long long)i * 3 * sDeg[ia] + (long long)kpos * 3, row);
          if (!diag) put3(sRow[ib] + (long long)i * 3 * sDeg[ib] + (long long)kpos2 * 3, col);
          }
#endif
          if constexpr (FU_CAP) {
            if (fu) {                                  // f_a += K_ab u_b, f_b += K_ba u_a
              const double* ub = sNod + (diag ? ia : ib) * NODW + RANK;
              const double* ua = sNod + ia * NODW + RANK;
              atomicAdd(&sF[ia * NDOF + i], row[0] * ub[0] + row[1] * ub[1] + row[2] * ub[2]);
              if (!diag) {
                const double v = col[0] * ua[0] + col[1] * ua[1] + col[2] * ua[2];
                atomicAdd(&sF[ib * NDOF + i], v);        // owned or halo: the image covers every tile node
              }
            }
          }
        }
      }
    }
    if constexpr (FU_CAP) {
      if (fu) {
        __syncthreads();
        for (int i = tid; i < n_all * NDOF; i += BLK) {
          const int io = i / NDOF;
          const double v = sF[i];
          if (v != 0.0) atomicAdd(&P.f[(int64_t)P.tnodes[node_off + io] * NDOF + (i - io * NDOF)], v);
        }
      }
    }
  } else
  for (int p0 = 0; p0 < n_pairs; p0 += BLK) {           // uniform trip count: every lane of a wave takes part in the write-out
    const int p = p0 + tid;
    const bool active = p < n_pairs;
    const uint4 pw = pw_next;
    pw_next = (p + BLK < n_pairs) ? prec[p + BLK] : make_uint4(0u, 0u, 0u, 0u);   // next pass's record in flight
    const uint32_t w0 = pw.x;
    const int ia = (int)(w0 & 0xFFFFu), kpos = (int)((w0 >> 16) & 0xFFu), cnt = (int)(w0 >> 24);
    const uint32_t* __restrict__ ents = P.tents + (int64_t)ent_off + pw.y;
    // b is owned by this tile as well: this lane also writes the block (b, a) -- K1 and K4 of it are the transposes
    const bool mirror = (pw.z >> 31) != 0;
    const int ib = (int)(pw.z & 0xFFFFu), kpos2 = (int)((pw.z >> 16) & 0xFFu);
    double Kuu[RANK][RANK], Kpp = 0.0, Kup[RANK], Kpu[RANK], Kup2[RANK], Kpu2[RANK];
#pragma unroll
    for (int i = 0; i < RANK; i++) {
      Kup[i] = 0.0; Kpu[i] = 0.0; Kup2[i] = 0.0; Kpu2[i] = 0.0;
#pragma unroll
      for (int j = 0; j < RANK; j++) Kuu[i][j] = 0.0;
    }
    // FA_ACCUMULATE (ACC), 4 x 4 blocks: the row chunks this lane is going to read-modify-write are requested now, through
    // LDS-DMA loads of 4 bytes each into a dead part of the LDS (the staged nodal data: phase 1 is over) -- no registers, and
    // nothing waits for them.  The element loop runs while the lines travel from HBM into the L2; the old values are then
    // read from there (cfg4 at 128^3 8.25 -> 7.81 ms, at 200^3 30.8 -> 29.5, keepOffDiags 10.2 -> 9.7: profiles/r03_batch38.txt;
    // requesting both blocks' old values together at the end on top of it costs 14 registers = a workgroup per CU: 7.97).
    if constexpr (FA_TILE_ACC_TOUCH && ACC && PF && RANK == 3) {
      const bool leader0 = !P.tile_split || ((pw.z >> 26) & 3u) == 0;
      if (active && leader0 && !P.tile_stage && n_all * NODW * 8 >= (BLK / 64) * 256 && P.cnt[0] == 4 && P.cnt[3] == 4) {
        char* sink = reinterpret_cast<char*>(sNod) + (tid >> 6) * 256;
        const double* q0 = P.K + sRow[ia] + (long long)kpos * 4;
        const long long rs = (long long)sDeg[ia] * 4;
#pragma unroll
        for (int i = 0; i < 4; i++) l2_touch(q0 + i * rs, sink);
        if (mirror) {
          const double* q1 = P.K + sRow[ib] + (long long)kpos2 * 4;
          const long long rs2 = (long long)sDeg[ib] * 4;
#pragma unroll
          for (int i = 0; i < 4; i++) l2_touch(q1 + i * rs2, sink);
        }
      }
    }
    uint32_t ent_next = pw.w;                            // the first entry travels in the pair record
#pragma unroll 1
    for (int t = 0; t < cnt; t++) {
      const uint32_t ent = ent_next;
      if (t + 1 < cnt) ent_next = ents[t + 1];          // in flight while this element's points are summed
      const int slot = (int)(ent & 0xFFFFu), la = (int)((ent >> 16) & 0xFFu), lb = (int)(ent >> 24);
#pragma unroll FA_TILE_UNROLL_IP
      for (int ip = 0; ip < NIP; ip++) {
        const double* rec = sRec + slot * SLOTW + ip * RW;
        const double w = rec[TR::O_W];
        double wga[RANK], gb[RANK], ga[RANK];
#pragma unroll
        for (int i = 0; i < RANK; i++) {               // same expression as phase 1: the gradients are bit-identical
          double acca = 0.0, accb = 0.0;
          if constexpr (TR::GREC) {
            acca = rec[TR::O_INV + la * RANK + i]; accb = rec[TR::O_INV + lb * RANK + i];
          } else {
#pragma unroll
            for (int k = 0; k < RANK; k++) {
              acca += tab2D[(ip * NPE + la) * RANK + k] * rec[TR::O_INV + k * RANK + i];
              accb += tab2D[(ip * NPE + lb) * RANK + k] * rec[TR::O_INV + k * RANK + i];
            }
          }
          wga[i] = w * acca; gb[i] = accb; ga[i] = acca;
        }
        if constexpr (FAM == FAM_PF_MIEHE) {
          double D[S][S];
          {
            int q = 0;
#pragma unroll
            for (int i = 0; i < S; i++)
#pragma unroll
              for (int j = i; j < S; j++) { const double v = rec[TR::O_D + q]; q++; D[i][j] = v; D[j][i] = v; }
          }
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
              for (int k = 0; k < S; k++) acc += Bi[k] * D[k][c];
              row[c] = acc;
            }
            Bt_vec<RANK, S>(gb, row, tt);
#pragma unroll
            for (int j = 0; j < RANK; j++) Kuu[i][j] += tt[j];
          }
        } else if constexpr (RANK == 2) {
          const double d0 = PF ? rec[TR::O_D] : P.hd0, d1 = PF ? rec[TR::O_D + 1] : P.hd1, d3 = PF ? rec[TR::O_D + 2] : P.hd3;
          Kuu[0][0] += (wga[0] * d0) * gb[0] + (wga[1] * d3) * gb[1];
          Kuu[0][1] += (wga[0] * d1) * gb[1] + (wga[1] * d3) * gb[0];
          Kuu[1][0] += (wga[1] * d1) * gb[0] + (wga[0] * d3) * gb[1];
          Kuu[1][1] += (wga[1] * d0) * gb[1] + (wga[0] * d3) * gb[0];
        } else if constexpr (!PF) {
          // Hooke: the tangent is a constant of the launch, so only M = sum w g_a g_b^T is accumulated per point
          // (9 multiply-adds); K1 = d1 M + d3 M^T + diag((d0 - d1 - d3) M_ii + d3 tr M) follows after the loops
#pragma unroll
          for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Kuu[i][j] += wga[i] * gb[j];
        } else {
          const double d0 = PF ? rec[TR::O_D] : P.hd0, d1 = PF ? rec[TR::O_D + 1] : P.hd1, d3 = PF ? rec[TR::O_D + 2] : P.hd3;
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
          const double Na = tab2N[ip * NPE + la], Nb = tab2N[ip * NPE + lb];
          const double wNa = w * Na;
          double gab = 0.0;
#pragma unroll
          for (int i = 0; i < RANK; i++) gab += wga[i] * gb[i];
          Kpp += (wNa * rec[TR::O_C1]) * Nb + P.c2 * gab;               // K4 (:714-716)
          if constexpr (KEEP) {
            double spv[S], tt[RANK];
#pragma unroll
            for (int c = 0; c < S; c++) spv[c] = rec[TR::O_SP + c];
            if (P.keep2) {                                              // K2 (:720)
              Bt_vec<RANK, S>(wga, spv, tt);
              const double dg = rec[TR::O_SP + S];
#pragma unroll
              for (int i = 0; i < RANK; i++) Kup[i] += (dg * tt[i]) * Nb;
              if (mirror) {                                             // K2 of (b, a)
                double wgb[RANK];
#pragma unroll
                for (int i = 0; i < RANK; i++) wgb[i] = w * gb[i];
                Bt_vec<RANK, S>(wgb, spv, tt);
#pragma unroll
                for (int i = 0; i < RANK; i++) Kup2[i] += (dg * tt[i]) * Na;
              }
            }
            if (P.keep3) {                                              // K3 (:721)
              Bt_vec<RANK, S>(gb, spv, tt);
              const double dgl = rec[TR::O_SP + S + 1];
#pragma unroll
              for (int j = 0; j < RANK; j++) Kpu[j] += (wNa * dgl) * tt[j];
              if (mirror) {                                             // K3 of (b, a)
                Bt_vec<RANK, S>(ga, spv, tt);
#pragma unroll
                for (int j = 0; j < RANK; j++) Kpu2[j] += ((w * Nb) * dgl) * tt[j];
              }
            }
          }
        }
      }
    }
    // lane groups (TileOpts::split): the 2 or 4 adjacent lanes that share a pair add their partial blocks with DPP
    // quad permutes; the first lane of the group goes on to write
    const int lg = P.tile_split ? (int)((pw.z >> 24) & 3u) : 0;
    if (P.tile_split && __any(lg != 0)) {
      auto dpp = [](double v, auto ctl) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __builtin_amdgcn_mov_dpp(lo, decltype(ctl)::value, 0xF, 0xF, true);
        hi = __builtin_amdgcn_mov_dpp(hi, decltype(ctl)::value, 0xF, 0xF, true);
        return __hiloint2double(hi, lo);
      };
      auto gsum = [&](double& v) {
        const double p1 = dpp(v, std::integral_constant<int, 0xB1>());      // quad_perm [1,0,3,2]
        v += (lg >= 1) ? p1 : 0.0;
        const double p2 = dpp(v, std::integral_constant<int, 0x4E>());      // quad_perm [2,3,0,1]
        v += (lg >= 2) ? p2 : 0.0;
      };
#pragma unroll
      for (int i = 0; i < RANK; i++) {
#pragma unroll
        for (int j = 0; j < RANK; j++) gsum(Kuu[i][j]);
      }
      if constexpr (PF) {
        gsum(Kpp);
        if constexpr (KEEP) {
#pragma unroll
          for (int i = 0; i < RANK; i++) { gsum(Kup[i]); gsum(Kpu[i]); gsum(Kup2[i]); gsum(Kpu2[i]); }
        }
      }
    }
    const bool leader = !P.tile_split || ((pw.z >> 26) & 3u) == 0;
    if constexpr (!PF && RANK == 3) {                  // Hooke in 3-D: K1 from M (see above)
      const double tr = Kuu[0][0] + Kuu[1][1] + Kuu[2][2];
      double Kn[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
          Kn[i][j] = (i == j) ? (P.hd0 - P.hd3) * Kuu[i][i] + P.hd3 * tr : P.hd1 * Kuu[i][j] + P.hd3 * Kuu[j][i];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Kuu[i][j] = Kn[i][j];
    }
    // ---- the block's CSR entries: rows of a node, columns of its kp-th neighbour (canonical blocked layout) ----
    // first entry of row i of owned node io: the rows of a node follow each other, deg * cnt[i'] entries each
    auto row_base = [&](int io, int i) -> long long {
      long long b = sRow[io];
      const int deg = sDeg[io];
#pragma unroll
      for (int r = 0; r < NDOF; r++) if (r < i) b += (long long)deg * P.cnt[r];
      return b;
    };
    auto emit = [&](int io, int kp, const double (&blk)[NDOF][NDOF]) {
#ifdef FA_TILE_NOWRITE                                   // ablation build: no K stores (the block stays live through a never-true test)
      double sum = 0.0;
#pragma unroll
      for (int i = 0; i < NDOF; i++)
#pragma unroll
        for (int j = 0; j < NDOF; j++) sum += blk[i][j];
      if (sum == 1.2345e300) P.K[sRow[io] + kp] = sum;
      return;
#endif
      // full 4x4 block: each row chunk is 32 bytes at a multiple of 32 bytes from K (two 16-byte stores if K itself is aligned)
      if (NDOF == 4 && P.cnt[0] == 4 && P.cnt[3] == 4 && (reinterpret_cast<uintptr_t>(P.K) & 15) == 0) {
        if constexpr (NDOF == 4) {
          const long long b0 = sRow[io] + (long long)kp * 4, rs = (long long)sDeg[io] * 4;
          if constexpr (ACC && FA_TILE_ACC_ATOMIC) {     // K += as 16 atomic adds without return: the L2 does the read-modify-write,
                                                         // the lane waits for nothing (every entry still has one writer)
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
              for (int j = 0; j < 4; j++) atomicAdd(P.K + b0 + i * rs + j, blk[i][j]);
          } else if constexpr (ACC) {                    // the four row chunks' old values in flight together
            double2 o0[4], o1[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { const double2* q = reinterpret_cast<const double2*>(P.K + b0 + i * rs); o0[i] = q[0]; o1[i] = q[1]; }
#pragma unroll
            for (int i = 0; i < 4; i++) {
              double2* q = reinterpret_cast<double2*>(P.K + b0 + i * rs);
              q[0] = make_double2(o0[i].x + blk[i][0], o0[i].y + blk[i][1]); q[1] = make_double2(o1[i].x + blk[i][2], o1[i].y + blk[i][3]);
            }
          } else {
#pragma unroll
          for (int i = 0; i < 4; i++) {
            double2* q = reinterpret_cast<double2*>(P.K + b0 + i * rs);
            double2 v0 = make_double2(blk[i][0], blk[i][1]), v1 = make_double2(blk[i][2], blk[i][3]);
            if (!P.overwrite) { const double2 o0 = q[0], o1 = q[1]; v0.x += o0.x; v0.y += o0.y; v1.x += o1.x; v1.y += o1.y; }
            q[0] = v0; q[1] = v1;
          }
          }
        }
      } else if constexpr (ACC) {                        // every old value of the block first, then the stores
        double old[NDOF][NDOF];
#pragma unroll
        for (int i = 0; i < NDOF; i++) {
          const long long bi = row_base(io, i) + (long long)kp * P.cnt[i];
#pragma unroll
          for (int j = 0; j < NDOF; j++) { const int o = P.off[i][j]; old[i][j] = (o >= 0) ? P.K[bi + o] : 0.0; }
        }
#pragma unroll
        for (int i = 0; i < NDOF; i++) {
          const long long bi = row_base(io, i) + (long long)kp * P.cnt[i];
#pragma unroll
          for (int j = 0; j < NDOF; j++) { const int o = P.off[i][j]; if (o >= 0) P.K[bi + o] = old[i][j] + blk[i][j]; }
        }
      } else {
#pragma unroll
        for (int i = 0; i < NDOF; i++) {
          const long long bi = row_base(io, i) + (long long)kp * P.cnt[i];
#pragma unroll
          for (int j = 0; j < NDOF; j++) {
            const int o = P.off[i][j];
            if (o < 0) continue;
            P.K[bi + o] = P.overwrite ? blk[i][j] : P.K[bi + o] + blk[i][j];
          }
        }
      }
    };
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
    if (P.tile_stage) {
      // Lanes are in (node, neighbour) order: the chunks of one CSR row held by consecutive lanes are consecutive in
      // memory.  Row by row they pass through LDS so that a store instruction writes 64 consecutive doubles
      // (LDS operations of one wave complete in order; the wave barriers only pin the compiler's schedule).
#pragma unroll
      for (int i = 0; i < NDOF; i++) {
        const int ci = P.cnt[i];
        if (ci == 0) continue;
        if (active) {
#pragma unroll
          for (int j = 0; j < NDOF; j++) { const int o = P.off[i][j]; if (o >= 0) stg[lane * ci + o] = blk[i][j]; }
          stg_base[lane] = row_base(ia, i) + (long long)kpos * ci;
        } else {
          stg_base[lane] = -1;
        }
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
      continue;
    }
    if (!active || !leader) continue;
    emit(ia, kpos, blk);
    if (mirror) {
#pragma unroll
      for (int i = 0; i < RANK; i++) {
#pragma unroll
        for (int j = 0; j < RANK; j++) blk[i][j] = Kuu[j][i];
        if constexpr (PF) {
          blk[i][RANK] = (KEEP && P.keep2) ? Kup2[i] : 0.0;
          blk[RANK][i] = (KEEP && P.keep3) ? Kpu2[i] : 0.0;
        }
      }
      emit(ib, kpos2, blk);
    }
  }
  }
#ifdef FA_TRACE
  trc[3] = __builtin_readcyclecounter();
  __builtin_amdgcn_s_waitcnt(0);
  trc[4] = __builtin_readcyclecounter();
  if ((tid & 63) == 0 && (tid >> 6) < 4 && P.dbg != nullptr) {
    long long* out = (long long*)P.dbg + ((int64_t)tile * 4 + (tid >> 6)) * 8;
    for (int q = 0; q < 5; q++) out[q] = trc[q];
    out[5] = n_pairs; out[6] = n_el; out[7] = n_all;
  }
#endif
  if (PF && P.err_part != nullptr) reduce_err<BLK>(err_acc, sErr, &P.err_part[tile]);
}

#ifndef FA_TILE_ACC
#define FA_TILE_ACC 1
#endif
template <class SH, int FAM, bool WANT_K, bool KEEP, class SP, int BLK>
static int launch_tiles_geom(const Params& P, int grid, hipStream_t st) {
  auto kern = assemble_tiles<SH, FAM, WANT_K, KEEP, SP, BLK, false>;
  if constexpr (WANT_K && FA_TILE_ACC) {
    if (!P.overwrite && !P.tile_stage) kern = assemble_tiles<SH, FAM, WANT_K, KEEP, SP, BLK, true>;
  }
  const size_t shmem = (size_t)P.tile_lds;
  if (shmem > 65536 - 64) {                            // per device and per context size: set on every launch (microseconds)
    hipError_t rc = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    if (rc != hipSuccess) return (int)rc;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(BLK), shmem, st, P);
  return (int)hipGetLastError();
}

template <class SH, int FAM, bool WANT_K, bool KEEP, class SP>
static int launch_tiles_one(const Params& P, int grid, hipStream_t st) {
  if constexpr (SH::RANK == 3) {                       // the big geometry exists for the 3-D shapes only
    if (P.tile_block == kTileBlockBig) return launch_tiles_geom<SH, FAM, WANT_K, KEEP, SP, kTileBlockBig>(P, grid, st);
  }
  if (P.tile_block != kTileBlock) return -1000;
  return launch_tiles_geom<SH, FAM, WANT_K, KEEP, SP, kTileBlock>(P, grid, st);
}

// getStress_ (PhaseFractureModel.cpp:1165-1271): one lane per node over its incident elements; per Gauss point the
// material is updated from the cached strain and the cached degradation (row STR of the cache), all STR components
// (2-D: xx, yy, zz, xy), and N_node(ip) * stress is summed; per element first, then into the node, like the reference.
template <int RANK>
__global__ void stress_table_kernel(const Params P, int npe, int nip, int plane_stress, double* __restrict__ table,
                                    double* __restrict__ weights) {
  constexpr int S = (RANK == 2) ? 3 : 6, STR = (RANK == 2) ? 4 : 6;
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P.nnode) return;
  const bool pf = P.model != 0;
  const int srows = pf ? STR + 1 : STR;
  double acc[STR], w = 0.0;
#pragma unroll
  for (int c = 0; c < STR; c++) acc[c] = 0.0;
  for (int32_t t = P.n2e_ptr[n]; t < P.n2e_ptr[n + 1]; t++) {
    const int32_t ent = P.n2e_ent[t];
    const int64_t e = ent / npe; const int la = ent - (int32_t)e * npe;
    double el[STR], ew = 0.0;
#pragma unroll
    for (int c = 0; c < STR; c++) el[c] = 0.0;
#pragma unroll 1
    for (int ip = 0; ip < nip; ip++) {
      const double N = P.tab[ip * npe + la];
      const double* sc = P.strain + (e * nip + ip) * srows;
      double eps[S], sig[S], sp[S], Psi, d0 = 0.0, d1 = 0.0, d3 = 0.0, szz = 0.0;
      if constexpr (RANK == 2) { eps[0] = sc[0]; eps[1] = sc[1]; eps[2] = sc[3]; }
      else {
#pragma unroll
        for (int c = 0; c < S; c++) eps[c] = sc[c];
      }
      const double g = pf ? sc[STR] : 1.0;
      if (pf && P.material == 3) {
        double D[S][S];
        miehe_update<RANK, S>(P, eps, g, D, sig, sp, Psi);
        if constexpr (RANK == 2) {                       // the zero zz eigenvalue sits in the negative part
          const double tr = eps[0] + eps[1];
          szz = (tr < 0.0) ? P.lmda * tr : g * (P.lmda * tr);
        }
      } else {
        iso_update<RANK, S, SpecRuntime>(P, eps, g, d0, d1, d3, sig, sp, Psi);
        if constexpr (RANK == 2) szz = (P.material != 1 && plane_stress) ? 0.0 : d1 * eps[0] + d1 * eps[1];
      }
      if constexpr (RANK == 2) { el[0] += N * sig[0]; el[1] += N * sig[1]; el[2] += N * szz; el[3] += N * sig[2]; }
      else {
#pragma unroll
        for (int c = 0; c < S; c++) el[c] += N * sig[c];
      }
      ew += N;
    }
#pragma unroll
    for (int c = 0; c < STR; c++) acc[c] += el[c];
    w += ew;
  }
#pragma unroll
  for (int c = 0; c < STR; c++) table[n * STR + c] += acc[c];
  weights[n] += w;
}

template <int RANK>
static int launch_stress_table_rank(const Params& P, int npe, int nip, int plane_stress, double* table, double* weights, void* stream) {
  hipLaunchKernelGGL(stress_table_kernel<RANK>, dim3((unsigned)((P.nnode + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                     P, npe, nip, plane_stress, table, weights);
  return (int)hipGetLastError();
}

// ---- GradientDamageModel's tables (GradientDamageModel.cpp:956-1365) ----
// DamageMaterial::update (DamageMaterial.cpp:356-437) at one Gauss point from the cached strain: `local` = the caller
// hands the material a strain of stress size, so the equivalent strain is computed from the strain (:370-373,
// getOutputData_), otherwise it is the cached nonlocal one in row STR (:374-377, getStress_).  sig: all STR components
// (2-D: xx, yy, zz, xy; the cached zz strain is 0).  hnew: newHist_ as :434-435 leave it (kappa, negative = loading).
template <int RANK>
__device__ __forceinline__ void gd_point_update(const Params& P, const double* __restrict__ sc, const double H0, const bool local,
                                                const int plane_stress, double* sig, double& hnew) {
  constexpr int S = (RANK == 2) ? 3 : 6, STR = (RANK == 2) ? 4 : 6;
  double eps[S], e0v[S], psi2 = 0.0;
  if constexpr (RANK == 2) { eps[0] = sc[0]; eps[1] = sc[1]; eps[2] = sc[3]; }
  else {
#pragma unroll
    for (int c = 0; c < S; c++) eps[c] = sc[c];
  }
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
  double eq = sc[STR];
  if (local) {
    double deq[S];
    eq = 0.5 * psi2;                                     // getEquiStrain (:602-652)
    if (P.eqv == 1) mazars_strain<RANK, S>(P, eps, eq, deq);
    else if (P.eqv == 2) vonmises_strain<S>(P, eps, eq, deq);
  }
  const double kappa0 = fabs(H0);
  const int load0 = (H0 < 0.0) ? 1 : 0;
  const double fl = eq - kappa0;
  double kappa; int load;
  if (fl == 0.0)     { kappa = kappa0; load = load0; }
  else if (fl > 0.0) { kappa = eq; load = (kappa < P.kI) ? 0 : 1; }
  else               { kappa = kappa0; load = 0; }
  double omega, domega;
  damage_law(P, kappa, omega, domega, P.soft);
  const double s1 = 1.0 - omega;
  if constexpr (RANK == 2) {
    sig[0] = s1 * e0v[0]; sig[1] = s1 * e0v[1]; sig[3] = s1 * e0v[2];
    sig[2] = plane_stress ? 0.0 : (s1 * P.hd1) * eps[0] + (s1 * P.hd1) * eps[1];
  } else {
#pragma unroll
    for (int c = 0; c < S; c++) sig[c] = s1 * e0v[c];
  }
  hnew = load ? -kappa : kappa;
}

// One lane per node over its incident elements, per element first, like the reference's element loops.  mode 0:
// getStress_ (:956-1063), NC = STR columns; mode 1: getDamage_ (:1184-1241), one column, omega of newHist_ weighed with
// |N| (values and weights, :1233-1234); mode 2: getOutputData_ (:1254-1365), the 2 STR + 1 point values
// [strain.tensor | stress.tensor (local update) | damage (giveOmega() before the loop, :1316)].  out is ADDED to.
template <int RANK, int MODE>
__global__ void gd_table_kernel(const Params P, int npe, int nip, int plane_stress, double* __restrict__ out,
                                double* __restrict__ weights) {
  constexpr int STR = (RANK == 2) ? 4 : 6;
  constexpr int NC = (MODE == 0) ? STR : (MODE == 1) ? 1 : 2 * STR + 1;
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= P.nnode) return;
  double acc[NC], w = 0.0;
#pragma unroll
  for (int c = 0; c < NC; c++) acc[c] = 0.0;
  for (int32_t t = P.n2e_ptr[n]; t < P.n2e_ptr[n + 1]; t++) {
    const int32_t ent = P.n2e_ent[t];
    const int64_t e = ent / npe; const int la = ent - (int32_t)e * npe;
    double el[NC], ew = 0.0;
#pragma unroll
    for (int c = 0; c < NC; c++) el[c] = 0.0;
#pragma unroll 1
    for (int ip = 0; ip < nip; ip++) {
      const int64_t ipoint = e * nip + ip;
      double N = P.tab[ip * npe + la];
      if constexpr (MODE == 1) {
        N = fabs(N);
        double omega, domega;
        damage_law(P, fabs(P.hist_new[ipoint]), omega, domega, P.soft);       // giveOmega (DamageMaterial.cpp:533-545)
        el[0] += N * omega;
      } else {
        const double* sc = P.strain + ipoint * (STR + 1);
        double sig[STR], hnew;
        gd_point_update<RANK>(P, sc, P.hist_pre[ipoint], MODE == 2, plane_stress, sig, hnew);
        if constexpr (MODE == 0) {
#pragma unroll
          for (int c = 0; c < STR; c++) el[c] += N * sig[c];
        } else {
          double omega, domega;
          damage_law(P, fabs(P.hist_new[ipoint]), omega, domega, P.soft);
#pragma unroll
          for (int c = 0; c < STR; c++) { el[c] += N * sc[c]; el[STR + c] += N * sig[c]; }
          el[2 * STR] += N * omega;
        }
      }
      ew += N;
    }
#pragma unroll
    for (int c = 0; c < NC; c++) acc[c] += el[c];
    w += ew;
  }
#pragma unroll
  for (int c = 0; c < NC; c++) out[n * NC + c] += acc[c];
  weights[n] += w;
}

// the history side of the same update (DamageMaterial.cpp:434-435): one lane per Gauss point, after the table kernel
template <int RANK>
__global__ void gd_hist_update_kernel(const Params P, int64_t npoint, int local, int plane_stress) {
  constexpr int STR = (RANK == 2) ? 4 : 6;
  const int64_t ipoint = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ipoint >= npoint) return;
  double sig[STR], hnew;
  gd_point_update<RANK>(P, P.strain + ipoint * (STR + 1), P.hist_pre[ipoint], local != 0, plane_stress, sig, hnew);
  P.hist_new[ipoint] = hnew;
}

template <int RANK>
static int launch_gd_table_rank(const Params& P, int npe, int nip, int plane_stress, int mode, double* out, double* weights,
                                void* stream) {
  const dim3 grid((unsigned)((P.nnode + 127) / 128)), blk(128);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0)      hipLaunchKernelGGL((gd_table_kernel<RANK, 0>), grid, blk, 0, st, P, npe, nip, plane_stress, out, weights);
  else if (mode == 1) hipLaunchKernelGGL((gd_table_kernel<RANK, 1>), grid, blk, 0, st, P, npe, nip, plane_stress, out, weights);
  else                hipLaunchKernelGGL((gd_table_kernel<RANK, 2>), grid, blk, 0, st, P, npe, nip, plane_stress, out, weights);
  int rc = (int)hipGetLastError();
  if (rc != 0 || mode == 1) return rc;
  const int64_t npoint = P.nel * nip;
  if (npoint > 0)
    hipLaunchKernelGGL(gd_hist_update_kernel<RANK>, dim3((unsigned)((npoint + 255) / 256)), dim3(256), 0, st, P, npoint,
                       mode == 2 ? 1 : 0, plane_stress);
  return (int)hipGetLastError();
}

// getArcFunc_ (PhaseFractureModel.cpp:844-1063): one lane per node over its incident elements (ascending, like the
// reference's element loop reaches the node); per Gauss point the strain, the current and old degradation, the material
// update with the current one (Psi = givePsi(), sigma+ = giveDerivPsi(), no history maximum: :1033-1038), then
//   jac10[disp dofs of the node] += wip (g0 - g) B_a^t sigma+,   jac10[phi dof] -= wip N_a g' Psi          (:1048-1049)
// summed per element first.  The lane of an element's local node 0 also adds the element's share of FE_DISSIPATION
// (:1046); per-workgroup partial sums in part[], reduced afterwards in a fixed order.  No atomics.
template <int RANK>
__global__ __launch_bounds__(128) void arc_func_kernel(const Params P, int npe, int nip, const double* __restrict__ state0,
                                                       double* __restrict__ jac10, double* __restrict__ part) {
  constexpr int S = (RANK == 2) ? 3 : 6, ND = RANK + 1;
  __shared__ double sfv[128];
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  auto dof = [&](int64_t node, int t) -> int64_t { return P.node_dofs ? P.node_dofs[node * ND + t] : node * ND + t; };
  double acc[ND], fv = 0.0;
#pragma unroll
  for (int c = 0; c < ND; c++) acc[c] = 0.0;
  if (n < P.nnode) {
    const double* tabN = P.tab;
    const double* tabD = P.tab + (size_t)nip * npe;
    const double* tabW = tabD + (size_t)nip * npe * RANK;
    for (int32_t t = P.n2e_ptr[n]; t < P.n2e_ptr[n + 1]; t++) {
      const int32_t ent = P.n2e_ent[t];
      const int64_t e = ent / npe; const int la = ent - (int32_t)e * npe;
      const int32_t* cn = P.conn + e * npe;
      double el[ND], efv = 0.0;
#pragma unroll
      for (int c = 0; c < ND; c++) el[c] = 0.0;
#pragma unroll 1
      for (int ip = 0; ip < nip; ip++) {
        const double* dn = tabD + (size_t)ip * npe * RANK;
        double J[RANK][RANK], inv[RANK][RANK], det;
#pragma unroll
        for (int i = 0; i < RANK; i++)
#pragma unroll
          for (int k = 0; k < RANK; k++) J[i][k] = 0.0;
        for (int b = 0; b < npe; b++) {
          const double* xb = P.coords + (int64_t)cn[b] * RANK;
#pragma unroll
          for (int i = 0; i < RANK; i++)
#pragma unroll
            for (int k = 0; k < RANK; k++) J[i][k] += xb[i] * dn[b * RANK + k];
        }
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
        double eps[S], pf = 0.0, pf0 = 0.0, ga[RANK];
#pragma unroll
        for (int c = 0; c < S; c++) eps[c] = 0.0;
        for (int b = 0; b < npe; b++) {
          double g[RANK], u[RANK];
#pragma unroll
          for (int i = 0; i < RANK; i++) {
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < RANK; k++) sacc += dn[b * RANK + k] * inv[k][i];
            g[i] = sacc; u[i] = P.state[dof(cn[b], i)];
          }
          if (b == la) {
#pragma unroll
            for (int i = 0; i < RANK; i++) ga[i] = g[i];
          }
          if constexpr (RANK == 2) {
            eps[0] += g[0] * u[0]; eps[1] += g[1] * u[1];
            eps[2] += g[1] * u[0]; eps[2] += g[0] * u[1];
          } else {
            eps[0] += g[0] * u[0]; eps[1] += g[1] * u[1]; eps[2] += g[2] * u[2];
            eps[3] += g[1] * u[0]; eps[3] += g[0] * u[1];
            eps[4] += g[2] * u[1]; eps[4] += g[1] * u[2];
            eps[5] += g[0] * u[2]; eps[5] += g[2] * u[0];
          }
          const double Nb = tabN[ip * npe + b];
          const int64_t pd = dof(cn[b], RANK);
          pf += Nb * P.state[pd]; pf0 += Nb * state0[pd];
        }
        double gphi, dgphi, ddg, gphi0, t1, t2;
        degrade<SpecRuntime>(P, pf, gphi, dgphi, ddg);
        degrade<SpecRuntime>(P, pf0, gphi0, t1, t2);
        double sig[S], sp[S], Psi;
        if (P.material == 3) {
          double D[S][S];
          miehe_update<RANK, S>(P, eps, gphi, D, sig, sp, Psi);
        } else {
          double d0, d1, d3;
          iso_update<RANK, S, SpecRuntime>(P, eps, gphi, d0, d1, d3, sig, sp, Psi);
        }
        const double wip = tabW[ip] * det;
        double bt[RANK];
        Bt_vec<RANK, S>(ga, sp, bt);
#pragma unroll
        for (int i = 0; i < RANK; i++) el[i] += wip * (gphi0 - gphi) * bt[i];
        el[RANK] -= wip * tabN[ip * npe + la] * dgphi * Psi;
        if (la == 0) efv += wip * (gphi0 - gphi) * Psi;
      }
#pragma unroll
      for (int c = 0; c < ND; c++) acc[c] += el[c];
      fv += efv;
    }
#pragma unroll
    for (int c = 0; c < ND; c++) jac10[dof(n, c)] = acc[c];
  }
  sfv[threadIdx.x] = fv;
  __syncthreads();
  for (int w = 64; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) sfv[threadIdx.x] += sfv[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sfv[0];
}

// MicroPhaseFracture* with AmorPhase: the reference's split material update writes stiffN only at Gauss points with
// trace(strain) <= 0 (AmorPhaseMaterial.cpp:256-263) into an array that lives outside the model's element loop
// (MicroPhaseFractureModel.cpp:535), so every later point of the serial loop in tension finds K PVol there.  One lane per
// element: the smallest serial index e * nip + ip of a Gauss point with tr <= 0 goes into `first` by a 64-bit atomic
// minimum; elements behind the minimum found so far leave at once (workgroups start roughly in order), so the pass costs
// a fraction of a sweep unless the whole mesh is in tension.  This is the ONE computation that decides the first
// compression point of the call: the assembly kernels, whose strain arithmetic differs from path to path, take every
// point before it as tension and the point itself as compression whatever the last bits of their own trace say
// (ip_material, SpecMA).  `mask` (optional, nullptr in the product path): bit ip of mask[e] set when tr <= 0 at point ip --
// with a mask no element may leave early.
template <int RANK>
__global__ __launch_bounds__(256) void first_comp_kernel(const Params P, int npe, int nip, unsigned long long* __restrict__ first,
                                                         uint32_t* __restrict__ mask) {
  constexpr int ND = RANK + 1;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P.nel) return;
  if (mask == nullptr && (unsigned long long)(e * nip) >= *(volatile unsigned long long*)first) return;
  auto dof = [&](int64_t node, int k) -> int64_t { return P.node_dofs ? P.node_dofs[node * ND + k] : node * ND + k; };
  const int32_t* cn = P.conn + e * npe;
  uint32_t m = 0;
  for (int ip = 0; ip < nip; ip++) {
    const double* dn = P.tab + (size_t)nip * npe + (size_t)ip * npe * RANK;
    double J[RANK][RANK], inv[RANK][RANK];
#pragma unroll
    for (int i = 0; i < RANK; i++)
#pragma unroll
      for (int k = 0; k < RANK; k++) J[i][k] = 0.0;
    for (int b = 0; b < npe; b++) {
      const double* xb = P.coords + (int64_t)cn[b] * RANK;
#pragma unroll
      for (int i = 0; i < RANK; i++)
#pragma unroll
        for (int k = 0; k < RANK; k++) J[i][k] += xb[i] * dn[b * RANK + k];
    }
    if constexpr (RANK == 2) {
      const double rd = 1.0 / (J[0][0] * J[1][1] - J[0][1] * J[1][0]);
      inv[0][0] = J[1][1] * rd;  inv[0][1] = -J[0][1] * rd;
      inv[1][0] = -J[1][0] * rd; inv[1][1] = J[0][0] * rd;
    } else {
      const double c00 = J[1][1] * J[2][2] - J[1][2] * J[2][1];
      const double c01 = J[1][2] * J[2][0] - J[1][0] * J[2][2];
      const double c02 = J[1][0] * J[2][1] - J[1][1] * J[2][0];
      const double rd = 1.0 / (J[0][0] * c00 + J[0][1] * c01 + J[0][2] * c02);
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
    double en[RANK];                                   // the normal strains, summed over the nodes as ip_point_core does
#pragma unroll
    for (int i = 0; i < RANK; i++) en[i] = 0.0;
    for (int b = 0; b < npe; b++) {
#pragma unroll
      for (int i = 0; i < RANK; i++) {
        double g = 0.0;
#pragma unroll
        for (int k = 0; k < RANK; k++) g += dn[b * RANK + k] * inv[k][i];
        en[i] += g * P.state[dof(cn[b], i)];
      }
    }
    const double tr = (RANK == 3) ? (en[0] + en[1] + en[RANK - 1]) : (en[0] + en[1] + 0.0);
    if (tr <= 0.0) { m |= 1u << ip; if (mask == nullptr) break; }
  }
  if (mask != nullptr) mask[e] = m;
  if (m != 0u) {
    const unsigned long long t = (unsigned long long)(e * nip + (__ffs((int)m) - 1));
    if (t < *(volatile unsigned long long*)first) atomicMin(first, t);
  }
}

template <int RANK>
static int launch_first_comp_rank(const Params& P, int npe, int nip, unsigned long long* first, uint32_t* mask, void* stream) {
  if (P.nel <= 0) return 0;
  hipLaunchKernelGGL(first_comp_kernel<RANK>, dim3((unsigned)((P.nel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, npe, nip, first, mask);
  return (int)hipGetLastError();
}

template <int RANK>
static int launch_arc_func_rank(const Params& P, int npe, int nip, const double* state0, double* jac10, double* part, void* stream) {
  hipLaunchKernelGGL(arc_func_kernel<RANK>, dim3((unsigned)((P.nnode + 127) / 128)), dim3(128), 0, (hipStream_t)stream,
                     P, npe, nip, state0, jac10, part);
  return (int)hipGetLastError();
}

template <class SH>
static int launch_tiles_shape(const Params& P, int fam, bool want_K, bool keep, int grid, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (fam == FAM_LE)
    return want_K ? launch_tiles_one<SH, FAM_LE, true, false, SpecFixed<0, 0, 0>>(P, grid, st)
                  : launch_tiles_one<SH, FAM_LE, false, false, SpecFixed<0, 0, 0>>(P, grid, st);
  if (fam == FAM_PF_ISO && P.model == 7) {              // GradientDamage: all four blocks
    return gd_dispatch<SpecCanon>(P, [&](auto spec) -> int {
      using SPG = decltype(spec);
      if (!want_K) return launch_tiles_one<SH, FAM_PF_ISO, false, false, SPG>(P, grid, st);
      return keep ? launch_tiles_one<SH, FAM_PF_ISO, true, true, SPG>(P, grid, st) : -1000;
    });
  }
  if (fam == FAM_PF_ISO && P.model == 1 && P.ptype == 0 && P.material == 1) {   // PhaseFracture, AT1 / AT2, AmorPhase folded in (cfg4)
    using SPF = SpecFixed<1, 0, 1>;
    if (!want_K) return launch_tiles_one<SH, FAM_PF_ISO, false, false, SPF>(P, grid, st);
    return keep ? launch_tiles_one<SH, FAM_PF_ISO, true, true, SPF>(P, grid, st)
                : launch_tiles_one<SH, FAM_PF_ISO, true, false, SPF>(P, grid, st);
  }
  if (fam == FAM_PF_ISO) {
    if (!want_K) return launch_tiles_one<SH, FAM_PF_ISO, false, false, SpecCanon>(P, grid, st);
    return keep ? launch_tiles_one<SH, FAM_PF_ISO, true, true, SpecCanon>(P, grid, st)
                : launch_tiles_one<SH, FAM_PF_ISO, true, false, SpecCanon>(P, grid, st);
  }
  if (fam == FAM_PF_MIEHE) {
    if (is_micro_amor(P)) {                             // MicroPhaseFracture* + AmorPhase (SpecMA)
      using SPM = SpecMA<SpecCanon>;
      if (!want_K) return launch_tiles_one<SH, FAM_PF_MIEHE, false, false, SPM>(P, grid, st);
      return keep ? launch_tiles_one<SH, FAM_PF_MIEHE, true, true, SPM>(P, grid, st)
                  : launch_tiles_one<SH, FAM_PF_MIEHE, true, false, SPM>(P, grid, st);
    }
    if (!want_K) return launch_tiles_one<SH, FAM_PF_MIEHE, false, false, SpecCanon>(P, grid, st);
    return keep ? launch_tiles_one<SH, FAM_PF_MIEHE, true, true, SpecCanon>(P, grid, st)
                : launch_tiles_one<SH, FAM_PF_MIEHE, true, false, SpecCanon>(P, grid, st);
  }
  return -1000;
}

}  // namespace fa
