// fa_tiles.h -- host-side construction of the node tiles of the tile kernel (fa_tile_impl.h).
#pragma once
#include <stdint.h>

#include <vector>

namespace fa {

struct TileSet {
  int ntile = 0;
  int lds_bytes = 0;                     // largest LDS footprint of a tile (dynamic shared memory of the launch)
  int lds_bytes_alt = 0;                 // the same with TileOpts::own_doubles_alt doubles per owned node
  std::vector<int32_t> tinfo;            // kTileInfo words per tile
  std::vector<int32_t> tnodes;           // nodes of every tile, owned ones first
  std::vector<int32_t> telems;           // elements of every tile, ascending; bit 31 = tile stores the per-point outputs
  std::vector<uint16_t> tconn;           // npe local node indices per tile element
  std::vector<uint32_t> tpairs;          // 4 words per pair lane: owned index | row position << 16 | elements << 24, entry offset,
                                         // mirror (second owned index | its row position << 16 | bit 31) or 0, pad
  std::vector<uint32_t> tents;           // one word per (pair, shared element)
  std::vector<uint32_t> eslot;           // fvec mode: per tile element (npe + 3) / 4 words, byte a = rank of the element among the
                                         // incident elements of its local node a (element-ascending)
  std::vector<uint32_t> nval;            // fvec mode: per tile node, the number of incident elements (owned nodes only)
  int max_all = 0, max_el = 0, max_own = 0;   // largest node / element / owned-node count of a tile
  int conn_stride = 0;                   // entries of tconn per tile element (npe, or 4 when padded)
  double point_redundancy = 0.0;         // tile elements / elements: how often a Gauss point is evaluated per pass
  // K blocks by destination (statistics of the pair lanes): rows of the lane's own node, mirror into the rows of another
  // owned node of the tile, mirror into the rows of a node another tile owns
  int64_t blocks_direct = 0, blocks_mirror_own = 0, blocks_mirror_halo = 0;
  int max_pairs = 0;                     // most pair lanes of a tile (the row-image write-out needs one pass)
};

// What the element-matrix kernel (fa_emat_impl.h) asks of the tiling on top of the tile kernel's:
struct TileOpts {
  int slot_doubles = 0;     // > 0: doubles of one element slot (instead of nip Gauss-point records of rec_doubles)
  int max_el = 0;           // > 0: at most this many elements per tile (phase 1 of the element-matrix kernel is one pass)
  int max_own = 0;          // > 0: at most this many owned nodes per tile
  bool diag_only = false;   // pair lanes only for (a, a): force-only passes sum f_int over a node's elements
  int order = -1;           // pair-lane order: 0 count-sorted with (a,b)/(b,a) sharing a lane, 1 row order with the
                            // diagonal blocks first, 2 plain (node, neighbour) order; -1: from `row_order`
  bool no_tab = false;      // no shape tables in LDS
  bool no_f_image = false;  // no f image in LDS
  int own_doubles = 0;      // extra doubles of LDS per owned node
  int own_doubles_alt = -1; // >= 0: a second, smaller per-node footprint the same tiles are also launched with (lds_bytes_alt)
  int max_all = 0;          // > 0: at most this many nodes (owned + halo) per tile
  bool conn_pad4 = false;   // tconn rows padded to 4 entries (8 bytes per element: the persistent kernels fetch them with
                            // two 4-byte LDS-DMA loads)
  bool sym_all = false;     // order 0 only: every unordered node pair {a, b} gets ONE lane, in the tile that owns the node
                            // with the smaller id; the lane also writes the block (b, a) into b's rows, whether or not
                            // this tile owns b (the tile holds every element with both nodes, so its sum is complete and
                            // every entry of K still has exactly one writer).  Halves the pair work; the kernel then
                            // needs the row bases of all tile nodes, not only the owned ones (12 bytes of LDS per node).
  bool sort_local = false;  // owned nodes of a tile in ascending node id instead of Morton order: on meshes numbered row by
                            // row the lanes of a wave then read consecutive LDS node slots (no bank conflicts) and the
                            // staging reads / f writes run over consecutive global addresses
  int split = 0;            // order 0 only, > 0: a pair that spans more than `split` elements is shared by 2 or 4 adjacent
                            // lanes (its elements dealt out in runs), which add their partial blocks with DPP row
                            // operations before the first of them writes: the lanes of a wave then do (nearly) equal
                            // work instead of 8 / 4 / 2 / 1 elements.  Pair word 2: bits 24-25 log2(lanes of the
                            // group), bits 26-27 index of this lane in its group.
  int group = 0;            // order 0 with sym_all, > 1: GROUPED pair lanes.  The neighbours b of an owned node a that share
                            // exactly the same elements with a (e.g. the 4 nodes of a Hex20 that lie on no face through
                            // its corner a) go to one lane, up to `group` of them: the lane reads a Gauss-point record and
                            // the gradient of a once per point and serves all its b with them.  Lane record (4 words):
                            //   ia | nb << 16 | elements << 24;  entry offset;  flags (bits 24-27 as with `split`, bit 30
                            //   diagonal);  first element word
                            // entries of a lane: nb words  ib | position of a in b's row << 16 | position of b in a's row
                            // << 24, then per element 2 words  slot | la << 16  and  lb_0 | lb_1 << 8 | lb_2 << 16 | lb_3 << 24
  int cell[3] = { 0, 0, 0 };// cell[0] > 0: tiles are boxes of cell[0] x cell[1] x cell[2] grid-quantised nodes (walked along a
                            // Morton curve of the boxes) instead of power-of-two Morton cells: 3 x 3 x 3 nodes of a Hex8 mesh
                            // bring 4 x 4 x 4 = 64 elements, i.e. exactly 512 (element, Gauss point) lanes
  int max_deg_sum = 0;      // > 0: a tile's owned nodes have at most this many neighbours in total (bounds its pair lanes)
  int lane_rows = 0;        // order 0 with split: 1 = the pair lanes of a tile are ordered by their owned node (inside a node: 4-lane
                            // groups, 2-lane groups, single lanes; every node's lanes start at a multiple of 4, padded with idle
                            // lanes), so that a store instruction of a wave lands in the row blocks of 3-4 nodes instead of 64
                            // (scattered 32-byte chunks reach 0.9-1.9 TB/s on MI355X, compact ones 5 TB/s:
                            // scripts/calib/store_patterns.hip); 0 = sorted by lane-group size over the whole tile
  bool no_pair_share = false; // order 0: no lane writes a mirror block at all (every directed pair has its own lane, all stores
                            // go to the rows of the lane's own node); overrides sym_all
  int fvec = 0;             // > 0: force-vector mode (fa_emat_impl.h, assemble_fvec): no element slots, no pair lanes; per
                            // element the slot ranks of its nodes, per owned node its valence (<= fvec)
};

// Nodes are sorted along a Morton curve of their grid-quantised coordinates and cut into tiles whose staged data
// (Gauss-point records of `rec_doubles` each, `node_doubles` per node, f image, shape tables, local connectivity)
// fit `lds_budget` bytes (`block` = threads of the workgroup that will run a tile: it sizes the staging rows).  nb_ptr / nb: sorted neighbour lists of the nodes (the column blocks of a node's rows).
// row_order: pair lanes in (node, neighbour) order (coalesced row chunks; for the 3-wide blocks that cannot use
// 32-byte stores) instead of sorted by element count with (a,b)/(b,a) sharing a lane.
// Returns false when the mesh cannot be tiled (a single node's elements exceed the budget, counters overflow).
bool build_tiles(int64_t nnode, int64_t nel, int npe, int nip, int rank, int ndofpn, const double* coords,
                 const int32_t* conn, const int32_t* n2e_ptr, const int32_t* n2e_ent, const int64_t* nb_ptr,
                 const int32_t* nb, int rec_doubles, int node_doubles, int lds_budget, int block, bool row_order, TileSet& out,
                 const TileOpts& opts = TileOpts());

// ELEMENT tiles (GET_INT_VECTOR passes of the 3-D shapes): the elements, sorted along a Morton curve of their centroids, are cut
// into runs of at most `max_el` elements / `max_all` nodes whose nodal data fit `lds_budget`; a tile's nodes are the nodes of
// its elements, all of them "owned" (tinfo: n_own = n_all).  Every element belongs to exactly one tile -- no Gauss point is
// evaluated twice (the node tiles above evaluate a Hex8 point 1.95 times, a Hex20 point 2.2 times) -- and the kernels add a
// tile's f rows to f_int with global atomics, since a node's row is then summed over several tiles.
// with_tab: the tile kernel's layout (shape tables and row bases in LDS); otherwise the force-vector kernel's one-row-per-node
// layout (eslot / nval are filled with zeros: that kernel's early loads read them).
bool build_element_tiles(int64_t nnode, int64_t nel, int npe, int nip, int rank, int ndofpn, const double* coords,
                         const int32_t* conn, int node_doubles, int lds_budget, int max_el, int max_all, bool with_tab,
                         bool conn_pad4, TileSet& out);

}  // namespace fa
