// fa_tiles.h -- host-side construction of the node tiles of the tile kernel (fa_tile_impl.h).
#pragma once
#include <stdint.h>

#include <vector>

namespace fa {

struct TileSet {
  int ntile = 0;
  int lds_bytes = 0;                     // largest LDS footprint of a tile (dynamic shared memory of the launch)
  std::vector<int32_t> tinfo;            // kTileInfo words per tile
  std::vector<int32_t> tnodes;           // nodes of every tile, owned ones first
  std::vector<int32_t> telems;           // elements of every tile, ascending; bit 31 = tile stores the per-point outputs
  std::vector<uint16_t> tconn;           // npe local node indices per tile element
  std::vector<uint32_t> tpairs;          // 4 words per pair lane: owned index | row position << 16 | elements << 24, entry offset,
                                         // mirror (second owned index | its row position << 16 | bit 31) or 0, pad
  std::vector<uint32_t> tents;           // one word per (pair, shared element)
  double point_redundancy = 0.0;         // tile elements / elements: how often a Gauss point is evaluated per pass
};

// Nodes are sorted along a Morton curve of their grid-quantised coordinates and cut into tiles whose staged data
// (Gauss-point records of `rec_doubles` each, `node_doubles` per node, f image, shape tables, local connectivity)
// fit `lds_budget` bytes (`block` = threads of the workgroup that will run a tile: it sizes the staging rows).  nb_ptr / nb: sorted neighbour lists of the nodes (the column blocks of a node's rows).
// row_order: pair lanes in (node, neighbour) order (coalesced row chunks; for the 3-wide blocks that cannot use
// 32-byte stores) instead of sorted by element count with (a,b)/(b,a) sharing a lane.
// Returns false when the mesh cannot be tiled (a single node's elements exceed the budget, counters overflow).
bool build_tiles(int64_t nnode, int64_t nel, int npe, int nip, int rank, int ndofpn, const double* coords,
                 const int32_t* conn, const int32_t* n2e_ptr, const int32_t* n2e_ent, const int64_t* nb_ptr,
                 const int32_t* nb, int rec_doubles, int node_doubles, int lds_budget, int block, bool row_order, TileSet& out);

}  // namespace fa
