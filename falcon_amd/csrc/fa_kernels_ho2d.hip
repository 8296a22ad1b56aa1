// Triangle3/Gauss3, Triangle6/Gauss3, Quad8/Gauss3*Gauss3 and Quad8/Gauss2*Gauss2 on the cooperative element kernel
#include "fa_coop_impl.h"
#include "fa_tile_impl.h"
namespace fa {
int launch_tri3_g3 (const Params& P, int fam, bool want_K, bool keep, int path, int grid, void* st) { if (path == PATH_TILES) return launch_tiles_shape<Shape<2, 3, 3>>(P, fam, want_K, keep, grid, st); return launch_coop_shape<Shape<2, 3, 3>>(P, fam, want_K, keep, path, grid, st); }
int launch_tri6_g3 (const Params& P, int fam, bool want_K, bool keep, int path, int grid, void* st) { if (path == PATH_TILES) return launch_tiles_shape<Shape<2, 6, 3>>(P, fam, want_K, keep, grid, st); return launch_coop_shape<Shape<2, 6, 3>>(P, fam, want_K, keep, path, grid, st); }
int launch_quad8_g3(const Params& P, int fam, bool want_K, bool keep, int path, int grid, void* st) { if (path == PATH_TILES) return launch_tiles_shape<Shape<2, 8, 9>>(P, fam, want_K, keep, grid, st); return launch_coop_shape<Shape<2, 8, 9>>(P, fam, want_K, keep, path, grid, st); }
int launch_quad8_g2(const Params& P, int fam, bool want_K, bool keep, int path, int grid, void* st) { if (path == PATH_TILES) return launch_tiles_shape<Shape<2, 8, 4>>(P, fam, want_K, keep, grid, st); return launch_coop_shape<Shape<2, 8, 4>>(P, fam, want_K, keep, path, grid, st); }
}  // namespace fa
