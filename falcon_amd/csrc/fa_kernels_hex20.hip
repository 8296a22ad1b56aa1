// Hex20 (serendipity) with Gauss3^3 and Gauss2^3 on the cooperative element kernel
#include "fa_coop_impl.h"
#include "fa_tile_impl.h"
namespace fa {
int launch_hex20_g3(const Params& P, int fam, bool want_K, bool keep, int path, int grid, void* st) { if (path == PATH_TILES) return launch_tiles_shape<Shape<3, 20, 27>>(P, fam, want_K, keep, grid, st); return launch_coop_shape<Shape<3, 20, 27>>(P, fam, want_K, keep, path, grid, st); }
int launch_hex20_g2(const Params& P, int fam, bool want_K, bool keep, int path, int grid, void* st) { if (path == PATH_TILES) return launch_tiles_shape<Shape<3, 20, 8>>(P, fam, want_K, keep, grid, st); return launch_coop_shape<Shape<3, 20, 8>>(P, fam, want_K, keep, path, grid, st); }
}  // namespace fa
