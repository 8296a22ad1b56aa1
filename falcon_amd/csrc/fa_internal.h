// fa_internal.h -- shared between the host-side C ABI (fa_api.cpp) and the gfx950 kernels.
#pragma once
#include <cstddef>
#include <stdint.h>

namespace fa {

// sets the calling thread's fa_last_error() text and returns `code` (defined in fa_api.hip)
int set_error(int code, const char* fmt, ...);


// Workgroup geometry per space dimension: threads per workgroup and capacity (in doubles) of the LDS image
// of the CSR rows.  2D: 128 threads + 20 KiB => 7-8 workgroups per CU; 3D row blocks are ~5x larger.
#ifndef FA_BLOCK2D
#define FA_BLOCK2D 256
#endif
#ifndef FA_LDS2D
#define FA_LDS2D 5120
#endif
#ifndef FA_WAVES2D
#define FA_WAVES2D 1
#endif
constexpr int block_for(int rank) { return rank == 2 ? FA_BLOCK2D : 256; }
constexpr int lds_for(int rank)   { return rank == 2 ? FA_LDS2D : 12288; }

// patch kernel geometry
#ifndef FA_PBLOCK
#define FA_PBLOCK 256
#endif
constexpr int kMaxOwn = 256;                          // owned nodes per patch (LDS tables)
constexpr int pblock_for(int npe) { return (FA_PBLOCK / npe) * npe; }   // threads: whole NPE-lane groups
constexpr size_t patch_shmem_bytes(int plds, int ndofpn) {     // K image, f image, global offsets, 4 int tables
  return sizeof(double) * ((size_t)plds + (size_t)kMaxOwn * ndofpn + kMaxOwn) + sizeof(int) * (4 * (size_t)kMaxOwn + 2);
}
// per patch-element record (int32 words): element id, its NPE nodes, NPE owned indices (u16, 0xFFFF = node of
// another patch), NPE rows of NPE column positions (u8)
constexpr int prec_own_words(int npe) { return (npe + 1) / 2; }
constexpr int prec_pos_words(int npe) { return (npe + 3) / 4; }
constexpr int prec_stride(int npe) { return 1 + npe + prec_own_words(npe) + npe * prec_pos_words(npe); }
// row kernel: one record per lane (= per (node, incident element) pair, in n2e order), so that a lane needs a single
// coalesced read before its nodal gathers: [0] e*npe + la, [1..npe] element nodes, [npe+1] LDS offset of the node's
// first row | (node - n0) << 16 | neighbour count << 24, then npe bytes of column positions; padded to 16 bytes.
constexpr int lrec_stride(int npe) { return ((2 + npe + (npe + 3) / 4) + 3) / 4 * 4; }
// pair kernel: the same record plus a flag word in the last slot (bit 0 owner, bit 1 half, bits 2.. store masks)
constexpr int lrec_stride2(int npe) { return ((3 + npe + (npe + 3) / 4) + 3) / 4 * 4; }
// per workgroup: n0, n1, first lane t0 (lo), lane count, CSR base (lo, hi), image length, t0 (hi)
constexpr int kWgInfo = 8;
enum { PATH_PATCH = 0, PATH_ROWS = 1, PATH_GENERIC = 2, PATH_ROWS_ORDERED = 3, PATH_COOP = 4, PATH_ROWS_PAIR = 5, PATH_TILES = 6, PATH_EMAT = 7, PATH_FVEC = 8 };

// kernel family: which compute core is instantiated
enum { FAM_LE = 0, FAM_PF_ISO = 1, FAM_PF_MIEHE = 2 };

// tile kernel (fa_tile_impl.h): workgroup size, LDS budget of one tile, descriptor words per tile
// (node offset, nodes, owned nodes, element offset, elements, pair offset, pairs, entry offset)
#ifndef FA_TILE_BLOCK
#define FA_TILE_BLOCK 512
#endif
#ifndef FA_TILE_LDS
#define FA_TILE_LDS 81856
#endif
constexpr int kTileBlock = FA_TILE_BLOCK, kTileLds = FA_TILE_LDS, kTileInfo = 8;
// second geometry for element slots too large for a good 80 KB tile (27-point Hex20, general tangents in 3-D): one
// 1024-thread workgroup per CU with (almost) all of its LDS
constexpr int kTileBlockBig = 1024, kTileLdsBig = 163000, kTileBigSlot = 200;
// doubles of one Gauss-point record: J^-1, w det J, tangent (3 isotropic coefficients or the packed symmetric D;
// nothing for Hooke, whose tangent is a constant),
// c1 for phase-field families, sigma+ / g' / loading g' when the off-diagonal blocks are kept
#ifndef FA_TILE_GREC
#define FA_TILE_GREC 0
#endif
// Hex8: the record holds the NPE physical shape gradients instead of J^-1 (the pair lanes then read 2 x 3 doubles per
// point instead of 9 + 2 x 3 and skip the 18 multiply-adds that rebuild the gradients)
constexpr bool tile_grad_records(int rank, int npe) { return FA_TILE_GREC && rank == 3 && npe == 8; }
constexpr int tile_rec_doubles(int rank, int fam, bool keep, int npe = 0) {
  const int S = (rank == 2) ? 3 : 6;
  return (tile_grad_records(rank, npe) ? npe * rank : rank * rank) + 1 + (fam == FAM_PF_MIEHE ? S * (S + 1) / 2 : (fam != FAM_LE ? 3 : 0)) + (fam != FAM_LE ? 1 : 0) + (keep ? S + 2 : 0);
}
// Grouped pair lanes (TileOpts::group, fa_tile_impl.h): neighbours per lane for the 3-D elastic kernels (Hex20 / Hex8 with
// Hooke), 1 = off.  Host and kernel must agree: the lane records differ.
#ifndef FA_TILE_GROUP
#define FA_TILE_GROUP 4
#endif
constexpr int tile_group(int rank, int fam) { return (rank == 3 && fam == FAM_LE) ? FA_TILE_GROUP : 1; }
// doubles of one element slot (nip records): odd, so that lanes reading the same field of different slots spread
// over the LDS banks
constexpr int tile_slot_doubles(int nip, int rec_doubles) { return (nip * rec_doubles) | 1; }
// bytes of the wave-private staging rows of the row-ordered write-out (64 chunks of a row + 64 row bases per wave)
constexpr int tile_stage_bytes(int ndofpn, int block) { return (block / 64) * (64 * ndofpn + 64) * 8; }
// doubles staged per tile node: x, u and (phase-field families) phi; the Ext variants add phi of state0 and of
// state00 / stateExt_ in an array of their own
constexpr int tile_node_doubles(int rank, bool pf) { return 2 * rank + (pf ? 1 : 0); }

// element-matrix kernel (fa_emat_impl.h): doubles of one element slot -- the node-pair blocks a <= b of K1 (+ the K4
// entry), f1 / f2, and K2 / K3 in full when the off-diagonal blocks are kept; odd
constexpr int emat_slot_doubles(int rank, int npe, bool pf, bool want_K, bool keep) {
  return ((want_K ? npe * (npe + 1) / 2 * (rank * rank + (pf ? 1 : 0)) : 0) + npe * (rank + (pf ? 1 : 0)) +
          ((want_K && keep) ? 2 * npe * npe * rank : 0)) | 1;
}

// persistent kernels: bytes of one LDS staging buffer (fa_emat_impl.h, StageLayout): per node 16 (coordinates) + 32 or 16
// (state: two 12-byte pieces at 16 bytes each, or one 16-byte piece) + 4 (valence) + 4 (id), per element two history pieces + 8 (connectivity) + 4 (element word) + 4 (rank word)
constexpr int stage_bytes_host(bool pf, int nip, int maxn, int maxe, bool hist) {
  return maxn * (16 + (pf ? 32 : 16) + 8) + maxe * ((hist ? 2 * (nip == 4 ? 16 : 4) : 0) + 16);
}

// cooperative element kernel (higher-order shapes): elements per workgroup so that the LDS slots of the largest
// family (Miehe with off-diagonal blocks) stay below 64 KB
constexpr int coop_slot_doubles(int rank, int npe, int nip) {
  const int S = (rank == 2) ? 3 : 6;
  return (npe + 1) / 2 + npe * (2 * rank + 3) + nip * (1 + npe * rank + rank + (3 + S * S + 2 * S + 4)) + 4;
}
constexpr int coop_epb(int rank, int npe, int nip) {
  const int n = 8000 / coop_slot_doubles(rank, npe, nip);
  return n < 1 ? 1 : (n > 32 ? 32 : n);
}

struct Params {
  // ---- mesh / topology (device) ----
  const int32_t* conn;       // nel*npe
  const double*  coords;     // nnode*rank
  const int64_t* row_ptr;    // ndof+1
  const int32_t* col_idx;    // nnz (generic kernel only)
  const int64_t* node_dofs;  // nnode*ndofpn or nullptr (generic kernel only)
  const int32_t* n2e_ptr;    // nnode+1   node -> incident (element, local node) pairs, element-ascending
  const int32_t* n2e_ent;    // nel*npe   e*npe + la
  const uint8_t* pos;        // nel*npe*npe: position of node conn[e][lb] in the sorted neighbour list of conn[e][la]
                             // (nullptr when the caller's pattern is not the canonical blocked layout)
  const int32_t* wg_start;   // nwg+1     first node of every workgroup (row kernel)
  const int32_t* lrec;       // nel*npe * lrec_stride(npe)  lane records of the row kernel
  const int32_t* wginfo;     // nwg * kWgInfo
  const int32_t* lrec2;      // pair kernel: lane records / workgroup descriptors of its own tiling
  const int32_t* wginfo2;
  // patch kernel: node patches with their halo of elements
  const int32_t* pn_ptr;     // npatch+1  owned nodes of patch p: pnode[pn_ptr[p] .. pn_ptr[p+1]), ascending ids
  const int32_t* pe_ptr;     // npatch+1  elements touching patch p: pelem[pe_ptr[p] .. pe_ptr[p+1])
  const int32_t* pr_ptr;     // npatch+1  runs of consecutive node ids
  const int32_t* pnode;
  const int32_t* poff;       // nnode+npatch  LDS offset of every owned node's row block, +1 sentinel per patch
  const int32_t* prec;       // per patch element: record of prec_stride(npe) words (see above)
  const int32_t* prun;       // first owned index | count << 16
  // tile kernel: compact node tiles with their halo of elements (fa_tile_impl.h)
  const int32_t*  tinfo;     // ntile3 * kTileInfo
  const int32_t*  tnodes;    // tile nodes, owned ones first
  const int32_t*  telems;    // tile elements, ascending; bit 31: this tile stores the element's per-point outputs
  const uint16_t* tconn;     // per tile element: npe indices into the tile's node list
  const uint32_t* tpairs;    // 4 words per pair lane: owned index | position in the row << 16 | elements << 24, entry offset, mirror, pad
  const uint32_t* tents;     // per (pair, shared element): slot | local node a << 16 | local node b << 24
  const uint32_t* eslot;     // force-vector tiles: per tile element, byte a = rank of the element at its local node a
  const uint32_t* nval;      // force-vector tiles: per tile node, incident elements (owned nodes)
  const int32_t*  tile_list; // nullptr, or the tiles this launch runs (interface / interior part of a pass): tile = tile_list[i]
  const double*  tab;        // shape tables: N[nip][npe], dN[nip][npe][rank], w[nip]
  // ---- vectors (device) ----
  const double*  state;
  const double*  state0;
  const double*  stateX;     // state00 (Ext) or stateExt_ (ExtIt)
  const double*  hist_pre;
  double*        hist_new;
  double*        strain;     // strain_ cache or nullptr
  void*          dbg;        // FA_TRACE builds: per-wave phase time stamps
  double*        err_part;   // per-workgroup partial sums of (pf-pfEx)^2 (ExtIt) or nullptr
  double*        K;
  double*        f;
  // ---- sizes / switches ----
  int64_t nnode, nel;
  int32_t model, material, ptype, keep2, keep3, overwrite, max_valence, ndofpn, col_sorted, plds, canon, ntile, ntile2, ntile3, tile_lds, tile_stage, tile_block, tile_sym, tile_split, tile_fu, fvec_atomic, f_atomic, rec_padded, pair_pers, tn_stride, te_stride, emat_lpe, fvec_val, conn_stride, st_maxn, st_maxe, persistent;
  int32_t cnt[4];            // entries per neighbour node in a row of dof type i
  int32_t off[4][4];         // offset of column type j inside that group, -1 if not in the pattern
  // ---- constants ----
  double Kb, G, lmda;        // bulk, shear, Miehe's literal lambda
  double hd0, hd1, hd3;      // Hooke D: normal diagonal, normal off-diagonal, shear
  double gcl, c2;            // gc/(cw*l0), 2*gc*l0/cw
  double eta, a1, a2, a3, psi0, beta, dt, dt0;
  double alpha;              // micromorphic penalty beta * gc / l0 (MicroPhaseFractureModel.cpp:579)
  // gradient damage (model 7): softening law and its parameters (DamageMaterial.cpp:157-268); c2 holds the length scale c
  int32_t soft, eqv;         // softening law, equivalent strain (0 Energy, 1 Mazars, 2 vonMises)
  int32_t k2sig, pstress;    // K2 from the stress record (sig / (1 - omega) = D0 eps) because sp holds d eq / d eps; plane stress
  double kI, kC, sA, sB, poiD;   // poiD = nu / (nu - 1) (DamageMaterial.cpp:971)
  double vmA, vmB, vmC;          // VonMisesStrain::init (VonMisesStrain.cpp:34-43)
  // MicroPhaseFracture* with AmorPhase: serial index (e * nip + ip) of the first Gauss point with trace(strain) <= 0 in this
  // call, found by first_comp_kernel before the assembly kernel runs; nullptr for every other model / material
  const unsigned long long* first_comp;
  // ... and, per element, the bit mask of its Gauss points with trace(strain) <= 0 from the same pre-pass: the one
  // classification every assembly kernel uses (their own strain arithmetic differs from path to path)
  const uint32_t* comp_mask;
  // MicroPhaseFracture*: counter of Gauss points that took the `d > 1` branch of MicroPhaseFractureModel.cpp:772-776, where
  // the reference keeps a dphidStrain left over from an earlier point of its serial loop and the kernels use 0
  unsigned long long* stale_hits;
};

// launchers, one translation unit per shape (fa_kernels_<shape>.hip)
typedef int (*launch_fn)(const Params& P, int fam, bool want_K, bool keep, int path, int grid, void* stream);
int launch_tri3_g1 (const Params&, int, bool, bool, int, int, void*);
int launch_quad4_g2(const Params&, int, bool, bool, int, int, void*);
int launch_hex8_g2 (const Params&, int, bool, bool, int, int, void*);
int launch_tri3_g3 (const Params&, int, bool, bool, int, int, void*);
int launch_tri6_g3 (const Params&, int, bool, bool, int, int, void*);
int launch_quad8_g3(const Params&, int, bool, bool, int, int, void*);
int launch_quad8_g2(const Params&, int, bool, bool, int, int, void*);
int launch_hex20_g3(const Params&, int, bool, bool, int, int, void*);
int launch_hex20_g2(const Params&, int, bool, bool, int, int, void*);

// element-matrix tile kernel (fa_kernels_emat.hip)
int launch_emat_quad4_g2(const Params&, int fam, bool want_K, bool keep, int grid, void* stream);
int launch_emat_tri3_g1 (const Params&, int fam, bool want_K, bool keep, int grid, void* stream);
int launch_fvec_quad4_g2(const Params&, int fam, int grid, void* stream);
int launch_fvec_tri3_g1 (const Params&, int fam, int grid, void* stream);
int launch_fvec_hex8_g2 (const Params&, int fam, int grid, void* stream);

// getStress_ table kernels, one per space dimension (fa_kernels_quad4.hip / fa_kernels_hex8.hip)
// GradientDamageModel's tables: mode 0 getStress_, 1 getDamage_, 2 getOutputData_ point values (fa_tile_impl.h)
int launch_gd_table_2d(const Params&, int npe, int nip, int plane_stress, int mode, double* out, double* weights, void* stream);
int launch_gd_table_3d(const Params&, int npe, int nip, int plane_stress, int mode, double* out, double* weights, void* stream);
int launch_stress_table_2d(const Params&, int npe, int nip, int plane_stress, double* table, double* weights, void* stream);
int launch_stress_table_3d(const Params&, int npe, int nip, int plane_stress, double* table, double* weights, void* stream);

// MicroPhaseFracture* + AmorPhase: first compression point of the call (same translation units)
int launch_first_comp_2d(const Params&, int npe, int nip, unsigned long long* first, uint32_t* mask, void* stream);
int launch_first_comp_3d(const Params&, int npe, int nip, unsigned long long* first, uint32_t* mask, void* stream);

// getArcFunc_ kernels (same translation units)
int launch_arc_func_2d(const Params&, int npe, int nip, const double* state0, double* jac10, double* part, void* stream);
int launch_arc_func_3d(const Params&, int npe, int nip, const double* state0, double* jac10, double* part, void* stream);

}  // namespace fa
