// gfx950 kernels instantiated for one shape (one translation unit per shape keeps builds parallel).
#include "fa_tile_impl.h"

namespace fa {
int launch_tri3_g1(const Params& P, int fam, bool want_K, bool keep, int path, int grid, void* stream) {
  if (path == PATH_FVEC) return launch_fvec_tri3_g1(P, fam, grid, stream);
  if (path == PATH_EMAT) return launch_emat_tri3_g1(P, fam, want_K, keep, grid, stream);
  if (path == PATH_TILES) return launch_tiles_shape<Shape<2,3,1>>(P, fam, want_K, keep, grid, stream);
  return launch_shape<Shape<2,3,1>>(P, fam, want_K, keep, path, grid, stream);
}
}  // namespace fa
