// gfx950 kernels instantiated for one shape (one translation unit per shape keeps builds parallel).
#include "fa_tile_impl.h"

namespace fa {
int launch_quad4_g2(const Params& P, int fam, bool want_K, bool keep, int path, int grid, void* stream) {
  if (path == PATH_FVEC) return launch_fvec_quad4_g2(P, fam, grid, stream);
  if (path == PATH_EMAT) return launch_emat_quad4_g2(P, fam, want_K, keep, grid, stream);
  if (path == PATH_TILES) return launch_tiles_shape<Shape<2,4,4>>(P, fam, want_K, keep, grid, stream);
  return launch_shape<Shape<2,4,4>>(P, fam, want_K, keep, path, grid, stream);
}
int launch_first_comp_2d(const Params& P, int npe, int nip, unsigned long long* first, uint32_t* mask, void* stream) {
  return launch_first_comp_rank<2>(P, npe, nip, first, mask, stream);
}
int launch_arc_func_2d(const Params& P, int npe, int nip, const double* state0, double* jac10, double* part, void* stream) {
  return launch_arc_func_rank<2>(P, npe, nip, state0, jac10, part, stream);
}

int launch_gd_table_2d(const Params& P, int npe, int nip, int plane_stress, int mode, double* out, double* weights, void* stream) {
  return launch_gd_table_rank<2>(P, npe, nip, plane_stress, mode, out, weights, stream);
}
int launch_stress_table_2d(const Params& P, int npe, int nip, int plane_stress, double* table, double* weights, void* stream) {
  return launch_stress_table_rank<2>(P, npe, nip, plane_stress, table, weights, stream);
}
}  // namespace fa
