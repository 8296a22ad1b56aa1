// element-matrix tile kernel (fa_emat_impl.h) instantiated for Quad4 and Triangle3
#include "fa_emat_impl.h"

namespace fa {
int launch_emat_quad4_g2(const Params& P, int fam, bool want_K, bool keep, int grid, void* stream) {
  return launch_emat_shape<Shape<2, 4, 4>>(P, fam, want_K, keep, grid, stream);
}
int launch_emat_tri3_g1(const Params& P, int fam, bool want_K, bool keep, int grid, void* stream) {
  return launch_emat_shape<Shape<2, 3, 1>>(P, fam, want_K, keep, grid, stream);
}
int launch_fvec_quad4_g2(const Params& P, int fam, int grid, void* stream) { return launch_fvec_shape<Shape<2, 4, 4>>(P, fam, grid, stream); }
int launch_fvec_tri3_g1(const Params& P, int fam, int grid, void* stream) { return launch_fvec_shape<Shape<2, 3, 1>>(P, fam, grid, stream); }
int launch_fvec_hex8_g2(const Params& P, int fam, int grid, void* stream) { return launch_fvec_shape<Shape<3, 8, 8>>(P, fam, grid, stream); }
}  // namespace fa
