// fa_api.hip -- host side of libfalcon_assemble.so: the C ABI declared in include/falcon_assemble.h.
//
// Set-up (fa_create) replaces what the reference's model constructors obtain from Jive
// (PhaseFractureModel.cpp:82-267): element group connectivity, DOF space, "FEM" matrix pattern;
// plus the topology the row-gather kernel needs (inverse connectivity, neighbour lists, per-element
// column positions, workgroup partition).  Everything lives in HBM afterwards.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/falcon_assemble.h"
#include "fa_internal.h"
#include "fa_patches.h"
#include "fa_tiles.h"

namespace {

thread_local std::string g_err;

fa_status fail(fa_status code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}

}  // namespace

// error text of the calling thread, for the other translation units of the library
int fa::set_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}

namespace {

#define HIP_TRY(expr)                                                                              \
  do { hipError_t _e = (expr);                                                                     \
       if (_e != hipSuccess) return fail(FA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } while (0)

// Debug / experiment switches.  They are read from the environment ONCE PER fa_create (so a test can build contexts
// on different kernel paths in one process) and nowhere else; the defaults are the measured best.  Documented in
// INTEGRATION.md ("Environment switches").
struct DebugOptions {
  int  tiles = -1;        // FA_TILES=0|1: tile kernel off / on for every shape (-1: per-shape default)
  int  tile_order = -1;   // FA_TILE_ORDER=0|1: pair lanes count-sorted with shared (a,b)/(b,a) lanes | row order
  int  tile_sym = -1;     // FA_TILE_SYM=0|1: one pair lane per unordered node pair across tiles off / on
  int  tile_split = -1;   // FA_TILE_SPLIT=n: pairs spanning more than n elements are shared by 2 or 4 lanes (0: off)
  int  tile_sort = -1;    // FA_TILE_SORT=0|1: owned nodes of a tile in Morton order | ascending node id
  int  tile_fu = -1;      // FA_TILE_FU=0|1: 3-D elastic GET_MATRIX0 passes take f_int from the pair lanes (K u) off / on
  int  tile_lanes = -1;   // FA_TILE_LANES=0|1|2|3: pair lanes sorted over the tile | by owned node | ... without cross-tile pair lanes | ... without any shared pair lanes
  int  elem_tiles = -1;   // FA_ELEM_TILES=0|1|2: GET_INT_VECTOR passes on element tiles (fa_tiles.h) off / force-vector kernel and 3-D shapes of the tile kernel / 2-D shapes of the tile kernel too
  int  elem_tile_el = 0;   // FA_ELEM_TILE_EL=n: elements per element tile of the tile kernel (0: default)
  int  tile_fblock = -1;  // FA_TILE_FBLOCK=512|1024: workgroup size of those passes (3-D shapes)
  int  tile_fset = -1;    // FA_TILE_FSET=0|1|n: GET_INT_VECTOR passes of the tile kernel on tiles of their own off / on / on with <= n elements per tile
  int  tile_budget = 0;   // FA_TILE_BUDGET=n: LDS bytes the tiles are cut for (experiments; 0: per geometry)
  int  tile_big = -1;     // FA_TILE_BIG=0|1: two 512-thread workgroups per CU | one of 1024 threads
  int  tile_cell = -1;    // FA_TILE_CELL=abc: tiles are boxes of a x b x c nodes (3-D shapes; 0: Morton cells; -1: per-shape default)
  int  emat = -1;         // FA_EMAT=0|1: element-matrix tile kernel (Quad4 / Triangle3) off / on
  int  fvec = -1;         // FA_FVEC=0|1: force-vector kernel (GET_INT_VECTOR passes of Quad4 / Triangle3) off / on
  int  persist = -1;      // FA_PERSIST=0|1: persistent LDS-DMA kernels off / on
  int  fvec_atomic = -1;  // FA_FVEC_ATOMIC=0|1: force-vector kernel with ordered per-element rows | one atomic row per node
  int  fvec_blk = 0;      // FA_FVEC_BLK=192|256|320: its workgroup size = elements per tile (0: default)
  int  fvec_cell = -1;    // FA_FVEC_CELL=abc: Hex8 force-vector tiles are boxes of a x b x c nodes (0 / -1: Morton cells)
  int  emat_lds = 0;      // FA_EMAT_LDS=n: LDS bytes of one element-matrix tile (0: default)
  int  emat_lpe = 0;      // FA_EMAT_LPE=1|2: lanes per element of its GET_MATRIX0 passes (0: default)
  int  emat_lpe_f = 0;    // FA_EMAT_LPE_F=1|2: the same for GET_INT_VECTOR passes
  bool no_pair = false;   // FA_NO_PAIR=1: no lane pairs for Quad4
  bool pair_3d = false;   // FA_PAIR_3D=1: lane pairs for Hex8 as well
  int  pad_rec = 1;       // FA_PAD_REC=0: lane records not padded per tile
  int  pad_tnodes = -1;   // FA_PAD_TNODES=0: node lists of the tile kernel not padded per tile (no early nodal loads)
  bool use_patch = false; // FA_USE_PATCH=1: experimental patch kernel
  int  patch_lds = 0;     // FA_PATCH_LDS=n: doubles of LDS for its CSR image
  bool verbose = false;   // FA_VERBOSE=1: tiling statistics on stderr
  int  pair_pers = -1;    // FA_PAIR_PERSIST=0|n: lane-pair kernel, GET_MATRIX0 passes: one workgroup per tile | n persistent workgroups (default: one per 16 tiles)
  int  micro_mask = 0;    // FA_MICRO_AMOR_MASK=1: Micro + AmorPhase, the pre-pass classifies EVERY Gauss point (see fa_create)
  static DebugOptions from_env() {
    DebugOptions o;
    auto geti = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
    o.tiles = geti("FA_TILES", -1); o.tile_order = geti("FA_TILE_ORDER", -1); o.tile_big = geti("FA_TILE_BIG", -1); o.tile_sym = geti("FA_TILE_SYM", -1); o.tile_fu = geti("FA_TILE_FU", -1); o.tile_sort = geti("FA_TILE_SORT", -1); o.tile_split = geti("FA_TILE_SPLIT", -1); o.tile_budget = geti("FA_TILE_BUDGET", 0); o.tile_lanes = geti("FA_TILE_LANES", -1); o.tile_fset = geti("FA_TILE_FSET", -1); o.tile_fblock = geti("FA_TILE_FBLOCK", -1); o.elem_tiles = geti("FA_ELEM_TILES", -1); o.elem_tile_el = geti("FA_ELEM_TILE_EL", 0);
    o.emat = geti("FA_EMAT", -1); o.emat_lds = geti("FA_EMAT_LDS", 0); o.emat_lpe = geti("FA_EMAT_LPE", 0);
    o.emat_lpe_f = geti("FA_EMAT_LPE_F", 0); o.fvec = geti("FA_FVEC", -1); o.fvec_blk = geti("FA_FVEC_BLK", 0); o.fvec_cell = geti("FA_FVEC_CELL", -1); o.fvec_atomic = geti("FA_FVEC_ATOMIC", -1);
    o.persist = geti("FA_PERSIST", -1); o.tile_cell = geti("FA_TILE_CELL", -1);
    o.no_pair = getenv("FA_NO_PAIR") != nullptr; o.pair_3d = getenv("FA_PAIR_3D") != nullptr;
    o.pad_rec = geti("FA_PAD_REC", 1); o.pad_tnodes = geti("FA_PAD_TNODES", -1); o.use_patch = getenv("FA_USE_PATCH") != nullptr;
    o.micro_mask = geti("FA_MICRO_AMOR_MASK", 0); o.pair_pers = geti("FA_PAIR_PERSIST", -1); o.patch_lds = geti("FA_PATCH_LDS", 0); o.verbose = getenv("FA_VERBOSE") != nullptr;
    return o;
  }
};

// The entry points run on the context's device and leave the caller's current device as they found it.
struct DeviceGuard {
  int prev = -1; bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
    if (prev == dev) prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define FA_ON_DEVICE(dev)                                                                          \
  DeviceGuard _guard(dev);                                                                         \
  if (!_guard.ok) return fail(FA_ERR_HIP, "hipSetDevice(%d) failed", (int)(dev))

template <class F>
void parallel_for(int64_t n, F&& body) {
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, n / 4096));
  if (nt <= 1) { body(0, n); return; }
  std::vector<std::thread> th;
  const int64_t chunk = (n + nt - 1) / nt;
  for (int t = 0; t < nt; t++) {
    const int64_t b = t * chunk, e = std::min(n, b + chunk);
    if (b >= e) break;
    th.emplace_back([&body, b, e] { body(b, e); });
  }
  for (auto& x : th) x.join();
}

// ---- shape tables (IShapeFactory stand-in): standard isoparametric elements, tensor Gauss rules ----
struct ShapeTab { int rank = 0, npe = 0, nip = 0; std::vector<double> N, dN, w; };

void gauss_line(int n, double* x, double* w) {
  if (n == 1) { x[0] = 0; w[0] = 2; }
  else if (n == 2) { x[0] = -std::sqrt(1.0 / 3.0); x[1] = -x[0]; w[0] = w[1] = 1; }
  else { x[0] = -std::sqrt(3.0 / 5.0); x[1] = 0; x[2] = -x[0]; w[0] = w[2] = 5.0 / 9.0; w[1] = 8.0 / 9.0; }
}

// shapes served by the cooperative element kernel instead of the lane-per-(node, element) kernels
bool coop_shape(int id) { return id != FA_TRI3_G1 && id != FA_QUAD4_G2 && id != FA_HEX8_G2; }

bool make_shape(int id, ShapeTab& s) {
  auto alloc = [&](int rank, int npe, int nip) {
    s.rank = rank; s.npe = npe; s.nip = nip;
    s.N.assign((size_t)nip * npe, 0.0); s.dN.assign((size_t)nip * npe * rank, 0.0); s.w.assign(nip, 0.0);
  };
  double gx[3], gw[3];
  switch (id) {
    case FA_TRI3_G1: {
      alloc(2, 3, 1);
      const double xi = 1.0 / 3.0, et = 1.0 / 3.0;
      s.N = { 1 - xi - et, xi, et };
      s.dN = { -1, -1, 1, 0, 0, 1 };
      s.w = { 0.5 };
      return true;
    }
    case FA_QUAD4_G2: {
      alloc(2, 4, 4);
      gauss_line(2, gx, gw);
      const int sx[4] = { -1, 1, 1, -1 }, sy[4] = { -1, -1, 1, 1 };
      for (int iy = 0; iy < 2; iy++) for (int ix = 0; ix < 2; ix++) {
        const int ip = iy * 2 + ix;
        for (int a = 0; a < 4; a++) {
          s.N[ip * 4 + a] = 0.25 * (1 + sx[a] * gx[ix]) * (1 + sy[a] * gx[iy]);
          s.dN[(ip * 4 + a) * 2 + 0] = 0.25 * sx[a] * (1 + sy[a] * gx[iy]);
          s.dN[(ip * 4 + a) * 2 + 1] = 0.25 * sy[a] * (1 + sx[a] * gx[ix]);
        }
        s.w[ip] = gw[ix] * gw[iy];
      }
      return true;
    }
    case FA_HEX8_G2: {
      alloc(3, 8, 8);
      gauss_line(2, gx, gw);
      const int sx[8] = { -1, 1, 1, -1, -1, 1, 1, -1 }, sy[8] = { -1, -1, 1, 1, -1, -1, 1, 1 },
                sz[8] = { -1, -1, -1, -1, 1, 1, 1, 1 };
      for (int iz = 0; iz < 2; iz++) for (int iy = 0; iy < 2; iy++) for (int ix = 0; ix < 2; ix++) {
        const int ip = (iz * 2 + iy) * 2 + ix;
        for (int a = 0; a < 8; a++) {
          const double fx = 1 + sx[a] * gx[ix], fy = 1 + sy[a] * gx[iy], fz = 1 + sz[a] * gx[iz];
          s.N[ip * 8 + a] = 0.125 * fx * fy * fz;
          s.dN[(ip * 8 + a) * 3 + 0] = 0.125 * sx[a] * fy * fz;
          s.dN[(ip * 8 + a) * 3 + 1] = 0.125 * sy[a] * fx * fz;
          s.dN[(ip * 8 + a) * 3 + 2] = 0.125 * sz[a] * fx * fy;
        }
        s.w[ip] = gw[ix] * gw[iy] * gw[iz];
      }
      return true;
    }
    case FA_TRI3_G3: case FA_TRI6_G3: {
      // 3-point rule at the "edge-interior" points, weight 1/6 each; Triangle6 in perimeter order
      // corner-mid-corner-mid-corner-mid (Jive's order)
      const int npe = (id == FA_TRI3_G3) ? 3 : 6;
      alloc(2, npe, 3);
      const double px[3] = { 1.0 / 6.0, 2.0 / 3.0, 1.0 / 6.0 }, py[3] = { 1.0 / 6.0, 1.0 / 6.0, 2.0 / 3.0 };
      for (int ip = 0; ip < 3; ip++) {
        const double L[3] = { 1.0 - px[ip] - py[ip], px[ip], py[ip] };
        const double dL[3][2] = { { -1, -1 }, { 1, 0 }, { 0, 1 } };
        double* N = &s.N[ip * npe]; double* d = &s.dN[ip * npe * 2];
        if (npe == 3) {
          for (int a = 0; a < 3; a++) { N[a] = L[a]; d[a * 2] = dL[a][0]; d[a * 2 + 1] = dL[a][1]; }
        } else {
          for (int c = 0; c < 3; c++) {
            const int n = (c + 1) % 3;                 // corner c -> node 2c; edge (c, c+1) -> node 2c+1
            N[2 * c] = L[c] * (2 * L[c] - 1);
            N[2 * c + 1] = 4 * L[c] * L[n];
            for (int k = 0; k < 2; k++) {
              d[(2 * c) * 2 + k] = (4 * L[c] - 1) * dL[c][k];
              d[(2 * c + 1) * 2 + k] = 4 * (L[n] * dL[c][k] + L[c] * dL[n][k]);
            }
          }
        }
        s.w[ip] = 1.0 / 6.0;
      }
      return true;
    }
    case FA_QUAD8_G3: case FA_QUAD8_G2: {
      // serendipity Quad8 in perimeter order corner-mid-corner-mid...
      const int n = (id == FA_QUAD8_G3) ? 3 : 2;
      alloc(2, 8, n * n);
      gauss_line(n, gx, gw);
      const int cx[8] = { -1, 0, 1, 1, 1, 0, -1, -1 }, cy[8] = { -1, -1, -1, 0, 1, 1, 1, 0 };
      for (int iy = 0; iy < n; iy++) for (int ix = 0; ix < n; ix++) {
        const int ip = iy * n + ix;
        const double x = gx[ix], y = gx[iy];
        for (int a = 0; a < 8; a++) {
          const double xa = cx[a], ya = cy[a];
          double N, dx, dy;
          if (cx[a] != 0 && cy[a] != 0) {
            N = 0.25 * (1 + xa * x) * (1 + ya * y) * (xa * x + ya * y - 1);
            dx = 0.25 * xa * (1 + ya * y) * (2 * xa * x + ya * y);
            dy = 0.25 * ya * (1 + xa * x) * (xa * x + 2 * ya * y);
          } else if (cx[a] == 0) {
            N = 0.5 * (1 - x * x) * (1 + ya * y); dx = -x * (1 + ya * y); dy = 0.5 * ya * (1 - x * x);
          } else {
            N = 0.5 * (1 + xa * x) * (1 - y * y); dx = 0.5 * xa * (1 - y * y); dy = -y * (1 + xa * x);
          }
          s.N[ip * 8 + a] = N; s.dN[(ip * 8 + a) * 2] = dx; s.dN[(ip * 8 + a) * 2 + 1] = dy;
        }
        s.w[ip] = gw[ix] * gw[iy];
      }
      return true;
    }
    case FA_HEX20_G3: case FA_HEX20_G2: {
      // serendipity Hex20: bottom-face perimeter (8), vertical mid-edges (4), top-face perimeter (8)
      // (Jive's Hex20 order is not evidenced by any shipped mesh; generator and table agree)
      const int n = (id == FA_HEX20_G3) ? 3 : 2;
      alloc(3, 20, n * n * n);
      gauss_line(n, gx, gw);
      int c[20][3];
      {
        const int px[8] = { -1, 0, 1, 1, 1, 0, -1, -1 }, py[8] = { -1, -1, -1, 0, 1, 1, 1, 0 };
        for (int a = 0; a < 8; a++) { c[a][0] = px[a]; c[a][1] = py[a]; c[a][2] = -1; c[12 + a][0] = px[a]; c[12 + a][1] = py[a]; c[12 + a][2] = 1; }
        for (int a = 0; a < 4; a++) { c[8 + a][0] = px[2 * a]; c[8 + a][1] = py[2 * a]; c[8 + a][2] = 0; }
      }
      for (int iz = 0; iz < n; iz++) for (int iy = 0; iy < n; iy++) for (int ix = 0; ix < n; ix++) {
        const int ip = (iz * n + iy) * n + ix;
        const double g[3] = { gx[ix], gx[iy], gx[iz] };
        for (int a = 0; a < 20; a++) {
          int mid = -1;                                 // direction along which the node sits at 0
          for (int k = 0; k < 3; k++) if (c[a][k] == 0) mid = k;
          double f[3];
          for (int k = 0; k < 3; k++) f[k] = 1 + c[a][k] * g[k];
          double N, d[3];
          if (mid < 0) {
            const double sum = c[a][0] * g[0] + c[a][1] * g[1] + c[a][2] * g[2] - 2;
            N = 0.125 * f[0] * f[1] * f[2] * sum;
            for (int k = 0; k < 3; k++) d[k] = 0.125 * c[a][k] * f[(k + 1) % 3] * f[(k + 2) % 3] * (sum + f[k]);
          } else {
            const int p = (mid + 1) % 3, q = (mid + 2) % 3;
            const double q2 = 1 - g[mid] * g[mid];
            N = 0.25 * q2 * f[p] * f[q];
            d[mid] = -0.5 * g[mid] * f[p] * f[q];
            d[p] = 0.25 * c[a][p] * q2 * f[q];
            d[q] = 0.25 * c[a][q] * q2 * f[p];
          }
          s.N[ip * 20 + a] = N;
          for (int k = 0; k < 3; k++) s.dN[(ip * 20 + a) * 3 + k] = d[k];
        }
        s.w[ip] = gw[ix] * gw[iy] * gw[iz];
      }
      return true;
    }
  }
  return false;
}

fa::launch_fn launcher_for(int shape) {
  switch (shape) {
    case FA_TRI3_G1:  return fa::launch_tri3_g1;
    case FA_QUAD4_G2: return fa::launch_quad4_g2;
    case FA_HEX8_G2:  return fa::launch_hex8_g2;
    case FA_TRI3_G3:  return fa::launch_tri3_g3;
    case FA_TRI6_G3:  return fa::launch_tri6_g3;
    case FA_QUAD8_G3: return fa::launch_quad8_g3;
    case FA_QUAD8_G2: return fa::launch_quad8_g2;
    case FA_HEX20_G3: return fa::launch_hex20_g3;
    case FA_HEX20_G2: return fa::launch_hex20_g2;
  }
  return nullptr;
}

template <class T>
fa_status upload(T** dst, const T* src, size_t n) {
  HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
  if (n && src) HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return FA_OK;
}

__global__ void fill_kernel(double* p, int64_t n, double v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// deterministic sum of the per-workgroup partials of errExt_
__global__ void reduce_err_kernel(const double* part, int n, double* out) {
  __shared__ double s[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *out = s[0];
}

// ---- scratch of the allocation-free entry point (fa_assemble_noalloc) ----
// Per-workgroup partial sums of errExt_ live in this file-scope device array instead of a hipMalloc'ed buffer.  Every
// GPU the library runs on holds its own copy (kNoAllocMaxParts doubles = 1 MiB) until the process ends.  In host code
// the name `scratch` is only a stand-in: the array's device address is resolved once per device with
// hipGetSymbolAddress (with that device current) and kept in g_scratch_ptr; host code uses that pointer everywhere.
constexpr int kNoAllocMaxParts = 1 << 17;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) double scratch[kNoAllocMaxParts];
double* g_scratch_ptr[kMaxDevices] = {};

// the caller has made `dev` current
fa_status resolve_scratch(int dev) {
  if (dev < 0 || dev >= kMaxDevices) return fail(FA_ERR_ARG, "device ordinal %d out of range", dev);
  if (g_scratch_ptr[dev]) return FA_OK;
  void* ptr = nullptr;
  hipError_t rc = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
  if (rc != hipSuccess || !ptr) return fail(FA_ERR_HIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(rc));
  g_scratch_ptr[dev] = static_cast<double*>(ptr);
  return FA_OK;
}

}  // namespace

// one tile set on the device (element-matrix kernel: one for GET_MATRIX0, one for GET_INT_VECTOR passes)
struct TileDev {
  bool ok = false;
  int ntile = 0, lds = 0, lpe = 1, block = 256;
  int lds_alt = 0;                                              // force-vector tiles: LDS bytes with one f row per owned node (atomic form)
  int tn_stride = 0, te_stride = 0;       // node / element lists padded to a fixed stride per tile (0: packed)
  int conn_stride = 0, maxn = 0, maxe = 0, plds = 0, pgrid = 0;   // persistent form: staging capacities, LDS bytes, grid
  bool persistent = false;
  double redundancy = 0.0;
  uint32_t* eslot = nullptr; uint32_t* nval = nullptr;
  int32_t *tinfo = nullptr, *tnodes = nullptr, *telems = nullptr; uint16_t* tconn = nullptr;
  uint32_t *tpairs = nullptr, *tents = nullptr;
};

// tiles of one tile set split into the ones that own an interface node ([0]) and the others ([1]): fa_set_interface
struct PartLists { int32_t* d[2] = { nullptr, nullptr }; int n[2] = { 0, 0 }; };

struct fa_ctx {
  fa_problem prob{};
  int rank = 0, npe = 0, nip = 0, str = 0, ndofpn = 0, fam = 0, srows = 0;
  int64_t nnode = 0, nel = 0, ndof = 0, nnz = 0;
  bool rows_path = false, own_pattern = false, coop = false;
  int ngrid_coop = 0;
  int nwg = 0, max_valence = 0, ngrid_generic = 0, block = 0, npatch = 0, plds = 0;
  bool patch_path = false, pair_path = false, tile_path = false, tile_force_only = false;
  int ncu = 256;
  int ntile2 = 0, ntile3 = 0, max_parts = 0;   // max_parts: workgroups of the largest launch (errExt_ partial sums)
  double lane_redundancy = 1.0;
  fa::launch_fn launch = nullptr;
  fa::Params P{};
  // device
  int32_t* d_conn = nullptr; double* d_coords = nullptr; int64_t* d_row_ptr = nullptr; int32_t* d_col_idx = nullptr;
  int64_t* d_node_dofs = nullptr; int32_t* d_n2e_ptr = nullptr; int32_t* d_n2e_ent = nullptr; uint8_t* d_pos = nullptr;
  int32_t* d_wg_start = nullptr; double* d_tab = nullptr; int32_t* d_lrec = nullptr; int32_t* d_wginfo = nullptr; int32_t* d_lrec2 = nullptr; int32_t* d_wginfo2 = nullptr;
  int32_t *d_tinfo = nullptr, *d_tnodes = nullptr, *d_telems = nullptr; uint16_t* d_tconn = nullptr;
  uint32_t *d_tpairs = nullptr, *d_tents = nullptr;
  int32_t *d_pn_ptr = nullptr, *d_pe_ptr = nullptr, *d_pr_ptr = nullptr, *d_pnode = nullptr, *d_poff = nullptr,
          *d_prec = nullptr, *d_prun = nullptr;
  double* d_hist[2] = { nullptr, nullptr };   // [0] preHist_, [1] newHist_
  double* d_strain = nullptr; double* d_err_part = nullptr;
  unsigned long long* d_first_comp = nullptr;   // MicroPhaseFracture* + AmorPhase: first compression point of the call
  uint32_t* d_comp_mask = nullptr;              // ... and the compression bits of every element's Gauss points (same pre-pass)
  bool d_first_comp_needed = false;
  unsigned long long* d_stale = nullptr;        // MicroPhaseFracture*: Gauss points of the last call in the `d > 1` branch
  hipEvent_t done_ev = nullptr;                 // recorded at the end of every fa_assemble: what the host getters wait for
  int tn_stride = 0, tn_strideF = 0;    // padded node lists of the matrix / force-only tile sets (0: not padded)
  double* d_out_tmp = nullptr;         // fa_output_data: strain and stress tables + weights before the column filter
  double* d_arc_part = nullptr;        // fa_arc_func: per-workgroup partial sums of FE_DISSIPATION
  int32_t* d_row_ptr32 = nullptr;      // fa_pattern_device32: the offsets as 32-bit integers, made on first request
  // staging for fa_assemble_host
  double *s_state = nullptr, *s_state0 = nullptr, *s_state00 = nullptr, *s_K = nullptr, *s_f = nullptr, *s_err = nullptr;
  std::vector<double> h_tmp;
  TileDev ematK, ematF;                // element-matrix kernel (Quad4 / Triangle3)
  TileDev fvec;                        // force-vector kernel (Quad4 / Triangle3)
  TileDev tileF;                       // tile kernel, GET_INT_VECTOR passes: a tile set of its own (no Gauss-point records: larger tiles)
  TileDev fvecU, tileU;                // the same two kernels on ELEMENT tiles (fa_tiles.h: every element in one tile, f_int by global atomics)
  bool parts_set = false;              // fa_set_interface has been called
  PartLists part_rows, part_pair, part_tiles, part_fvec, part_tileF;
  DebugOptions dbg;
};

extern "C" const char* fa_last_error(void) { return g_err.c_str(); }

extern "C" void fa_destroy(fa_ctx* c) {
  if (!c) return;
  DeviceGuard guard(c->prob.device);
  void* ptrs[] = { c->d_conn, c->d_coords, c->d_row_ptr, c->d_col_idx, c->d_node_dofs, c->d_n2e_ptr, c->d_n2e_ent,
                   c->d_pos, c->d_wg_start, c->d_lrec, c->d_wginfo, c->d_lrec2, c->d_wginfo2, c->d_tab, c->d_hist[0], c->d_hist[1], c->d_strain, c->d_err_part, c->d_first_comp, c->d_comp_mask, c->d_stale, c->d_row_ptr32, c->d_arc_part, c->d_out_tmp,
                   c->s_state, c->s_state0, c->s_state00, c->s_K, c->s_f, c->s_err,
                   c->d_pn_ptr, c->d_pe_ptr, c->d_pr_ptr, c->d_pnode, c->d_poff, c->d_prec, c->d_prun,
                   c->d_tinfo, c->d_tnodes, c->d_telems, c->d_tconn, c->d_tpairs, c->d_tents,
                   c->ematK.tinfo, c->ematK.tnodes, c->ematK.telems, c->ematK.tconn, c->ematK.tpairs, c->ematK.tents,
                   c->ematF.tinfo, c->ematF.tnodes, c->ematF.telems, c->ematF.tconn, c->ematF.tpairs, c->ematF.tents,
                   c->fvec.tinfo, c->fvec.tnodes, c->fvec.telems, c->fvec.tconn, c->fvec.tpairs, c->fvec.tents, c->fvec.eslot, c->fvec.nval,
                   c->tileF.tinfo, c->tileF.tnodes, c->tileF.telems, c->tileF.tconn, c->tileF.tpairs, c->tileF.tents,
                   c->fvecU.tinfo, c->fvecU.tnodes, c->fvecU.telems, c->fvecU.tconn, c->fvecU.tpairs, c->fvecU.tents, c->fvecU.eslot, c->fvecU.nval,
                   c->tileU.tinfo, c->tileU.tnodes, c->tileU.telems, c->tileU.tconn, c->tileU.tpairs, c->tileU.tents };
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (PartLists* pl : { &c->part_rows, &c->part_pair, &c->part_tiles, &c->part_fvec, &c->part_tileF })
    for (int k = 0; k < 2; k++) if (pl->d[k]) (void)hipFree(pl->d[k]);
  if (c->done_ev) (void)hipEventDestroy(c->done_ev);
  delete c;
}

// Node lists of a tile set re-laid with a fixed stride per tile (idle slots: node 0), so that the kernel can address a
// lane's first tile node from the tile index alone (assemble_tiles, "pre").  Returns the stride, 0 when it is not worth
// it (a tile with more than 1024 nodes would make every list that long).
static int pad_tile_nodes(fa::TileSet& ts) {
  int mx = 0;
  for (int t = 0; t < ts.ntile; t++) mx = std::max(mx, ts.tinfo[(size_t)t * fa::kTileInfo + 1]);
  if (ts.ntile == 0 || mx > 1024) return 0;
  const int stride = (mx + 15) & ~15;
  std::vector<int32_t> padded((size_t)ts.ntile * stride, 0);
  for (int t = 0; t < ts.ntile; t++) {
    int32_t* ti = ts.tinfo.data() + (size_t)t * fa::kTileInfo;
    std::copy(ts.tnodes.begin() + ti[0], ts.tnodes.begin() + ti[0] + ti[1], padded.begin() + (size_t)t * stride);
    ti[0] = t * stride;
  }
  ts.tnodes.swap(padded);
  if (!ts.nval.empty()) {                               // force-vector tiles: the valences run parallel to the node lists
    std::vector<uint32_t> nv((size_t)ts.ntile * stride, 0);
    size_t old_off = 0;
    for (int t = 0; t < ts.ntile; t++) {
      const int n_all = ts.tinfo[(size_t)t * fa::kTileInfo + 1];
      std::copy(ts.nval.begin() + old_off, ts.nval.begin() + old_off + n_all, nv.begin() + (size_t)t * stride);
      old_off += (size_t)n_all;
    }
    ts.nval.swap(nv);
  }
  return stride;
}

// The same for the element lists of the force-vector tiles (element words, rank words, local connectivity): a lane's
// element word and with it the element's history are requested without waiting for the descriptor.
static int pad_tile_elems(fa::TileSet& ts, int esw, int cs) {
  int mx = 0;
  for (int t = 0; t < ts.ntile; t++) mx = std::max(mx, ts.tinfo[(size_t)t * fa::kTileInfo + 4]);
  if (ts.ntile == 0 || mx > 1024) return 0;
  const int stride = (mx + 15) & ~15;
  std::vector<int32_t> te((size_t)ts.ntile * stride, 0);
  std::vector<uint32_t> es((size_t)ts.ntile * stride * esw, 0);
  std::vector<uint16_t> tc((size_t)ts.ntile * stride * cs, 0);
  for (int t = 0; t < ts.ntile; t++) {
    int32_t* ti = ts.tinfo.data() + (size_t)t * fa::kTileInfo;
    const size_t o = (size_t)ti[3], n = (size_t)ti[4], d = (size_t)t * stride;
    std::copy(ts.telems.begin() + o, ts.telems.begin() + o + n, te.begin() + d);
    std::copy(ts.eslot.begin() + o * esw, ts.eslot.begin() + (o + n) * esw, es.begin() + d * esw);
    std::copy(ts.tconn.begin() + o * cs, ts.tconn.begin() + (o + n) * cs, tc.begin() + d * cs);
    ti[3] = (int32_t)d;
  }
  ts.telems.swap(te); ts.eslot.swap(es); ts.tconn.swap(tc);
  return stride;
}

static fa_status create_impl(const fa_problem* p, fa_ctx* c) {
  c->prob = *p;
  c->dbg = DebugOptions::from_env();
  const DebugOptions& dbg = c->dbg;
  ShapeTab sh;
  if (!make_shape(p->shape, sh))
    return fail(FA_ERR_ARG, "unknown shape id %d", p->shape);
  c->launch = launcher_for(p->shape);
  c->rank = sh.rank; c->npe = sh.npe; c->nip = sh.nip; c->str = (sh.rank == 2) ? 4 : 6;
  if (p->model < FA_LINEAR_ELASTICITY || p->model > FA_GRADIENT_DAMAGE) return fail(FA_ERR_ARG, "invalid model %d", p->model);
  const bool gd = p->model == FA_GRADIENT_DAMAGE;
  const bool pf = p->model != FA_LINEAR_ELASTICITY;
  c->ndofpn = pf ? sh.rank + 1 : sh.rank;
  c->srows = pf ? c->str + 1 : c->str;
  // material <-> model pairing (PhaseFractureModel.cpp:193-194 dynamicCast<PhaseFractureMaterial>)
  if (!pf && p->material != FA_HOOKE) return fail(FA_ERR_ARG, "LinearElasticity needs material Hooke");
  if (pf && !gd && p->material == FA_HOOKE) return fail(FA_ERR_ARG, "phase-field models need a PhaseFractureMaterial");
  if (p->material < FA_HOOKE || p->material > FA_DAMAGE) return fail(FA_ERR_ARG, "Invalid material: %d", p->material);
  // GradientDamageModel.cpp:205-206 dynamicCast<DamageMaterial>
  if (gd != (p->material == FA_DAMAGE)) return fail(FA_ERR_ARG, "GradientDamage needs material Damage, and only GradientDamage takes it");
  if (gd) {
    if (p->softening < FA_SOFT_PERFECT || p->softening > FA_SOFT_MULTILINEAR)
      return fail(FA_ERR_ARG, "unexpected definition of softening law!!! (0 ... 10, the order of DamageMaterial's softeningLawDefs)");
    if (!(p->kappa_i > 0.0)) return fail(FA_ERR_ARG, "Damage: kappaI (or ft / young) must be positive");
    if (p->eqv_strain < FA_EQV_ENERGY || p->eqv_strain > FA_EQV_VONMISES)
      return fail(FA_ERR_ARG, "unexpected definition of equivalent strain!!! Available ones are: Mazars, vonMises, Energy");
    if (p->eqv_strain == FA_EQV_VONMISES) {
      // DamageMaterial.cpp:677-683 returns a 6-vector whatever the strain size; GradientDamageModel.cpp:548 multiplies it
      // with a 4-row B matrix in 2-D
      if (sh.rank != 3) return fail(FA_ERR_UNSUPPORTED, "eqvStrain = vonMises is available for 3-D shapes only");
      if (!(p->eta > 0.0)) return fail(FA_ERR_ARG, "eqvStrain = vonMises needs eta > 0");
    }
    const int sl = p->softening;
    if ((sl == FA_SOFT_LINEAR1 || sl == FA_SOFT_LINEAR2 || sl == FA_SOFT_EXPONENTIAL2 || sl == FA_SOFT_POWER || sl == FA_SOFT_POLYNOMIAL) &&
        !(p->kappa_c > p->kappa_i))
      return fail(FA_ERR_ARG, "Damage: kappaC must exceed kappaI");
    if (sl == FA_SOFT_HYPERBOLIC && !(p->soft_b >= 0.0)) return fail(FA_ERR_ARG, "Damage: b must not be negative");
    if (sl == FA_SOFT_EXPOENERGY && !(p->kappa_c > 0.0 && p->soft_s >= 0.0)) return fail(FA_ERR_ARG, "Damage: expoEnergy needs kappaC > 0 and s >= 0");
    if (sl == FA_SOFT_MULTILINEAR && !(p->kappa_p >= 0.0 && p->kappa_c > p->kappa_p)) return fail(FA_ERR_ARG, "Damage: multilinear needs kappaC > kappaP >= 0");
  }
  c->fam = !pf ? fa::FAM_LE : (p->material == FA_MIEHE_PHASE ? fa::FAM_PF_MIEHE : fa::FAM_PF_ISO);
  if (p->model >= FA_MICRO_PHASE_FRACTURE && p->model <= FA_MICRO_PHASE_FRACTURE_EXTIT) {
    // AmorPhase's split update leaves stiffN untouched in tension (AmorPhaseMaterial.cpp:239-268): a point in tension finds
    // what the last compression point of the serial loop left there.  That is K PVol from the first compression point of
    // the call on -- first_comp_kernel finds it before the assembly kernel runs -- and the array's initial contents before
    // (Jem does not initialise it; zero is assumed here and in the checker).
    c->fam = fa::FAM_PF_MIEHE;                          // general-tangent family (rank-one term of elemMat1)
    c->d_first_comp_needed = (p->material == FA_AMOR_PHASE);
  }
  if (sh.rank == 2 && p->state != FA_PLANE_STRAIN && p->state != FA_PLANE_STRESS)
    return fail(FA_ERR_ARG, "unexpected rank/state");     // HookeMaterial.cpp:262-265
  if (p->nnode <= 0 || p->nel < 0 || !p->coords || (!p->conn && p->nel)) return fail(FA_ERR_ARG, "empty mesh arrays");
  if (p->nnode > 2000000000LL || p->nel * sh.npe > 2000000000LL) return fail(FA_ERR_UNSUPPORTED, "mesh too large for 32-bit topology");
  const int64_t nnode = p->nnode, nel = p->nel;
  const int npe = sh.npe, nd = c->ndofpn, rank = sh.rank;
  c->nnode = nnode; c->nel = nel;
  for (int64_t k = 0; k < nel * npe; k++)
    if (p->conn[k] < 0 || p->conn[k] >= nnode) return fail(FA_ERR_ARG, "connectivity entry %lld out of range", (long long)k);

  fa::Params& P = c->P;
  // ---- fracture constants (PhaseFractureModel.cpp:296-355) ----
  {
    const double E = p->young, gc = p->gc, l0 = p->l0, sT = p->sigma_t;
    double cw = 2, eta = 0, pe = 2, a1 = 2, a2 = -0.5, a3 = 0, psi0 = 0;
    if (pf && !gd) {
      switch (p->frac_type) {
        case FA_AT1: cw = 8.0 / 3.0; eta = 1.0; psi0 = (3. * gc) / (16. * l0); break;
        case FA_AT2: break;
        case FA_CZM_LINEAR: cw = 3.14159265359; eta = 2.0; a1 = (4.0 * E * gc) / (cw * l0 * sT * sT);
                            psi0 = 0.5 * (sT * sT) / E; break;
        case FA_CZM_EXPONENTIAL: cw = 3.14159265359; eta = 2.0; pe = 2.5; a1 = (4.0 * E * gc) / (cw * l0 * sT * sT);
                            a2 = ::pow(2.0, 5.0 / 3.0) - 3.0; psi0 = 0.5 * (sT * sT) / E; break;
        case FA_CZM_CORNELISSEN: cw = 3.14159265359; eta = 2.0; a1 = (4.0 * E * gc) / (cw * l0 * sT * sT);
                            a2 = 1.3868; a3 = 0.6567; psi0 = 0.5 * (sT * sT) / E; break;
        case FA_CZM_HYPERBOLIC: cw = 3.14159265359; eta = 2.0; pe = 4.0; a1 = (4.0 * E * gc) / (cw * l0 * sT * sT);
                            a2 = ::pow(2.0, 7.0 / 3.0) - 4.5; psi0 = 0.5 * (sT * sT) / E; break;
        default: return fail(FA_ERR_ARG, "unexpected definition of fracture type!!! Supported types include: "
                                         "at1, at2, czmLinear, czmExponential, czmCornelissen, czmHyperbolic");
      }
    }
    P.ptype = (pe == 2.0) ? 0 : (pe == 2.5 ? 1 : 2);
    P.gcl = gc / (cw * l0); P.c2 = 2.0 * gc * l0 / cw;
    P.eta = eta; P.a1 = a1; P.a2 = a2; P.a3 = a3; P.psi0 = psi0; P.beta = p->beta;
    P.alpha = p->beta * gc / l0;
    // elastic constants (AmorPhaseMaterial.cpp:85-86, MiehePhaseMaterial.cpp:97-99, HookeMaterial.cpp:202-267)
    const double nu = p->poisson;
    P.Kb = E / (3.0 * (1.0 - 2.0 * nu)); P.G = E / (2.0 * (1.0 + nu)); P.lmda = P.Kb - 0.666666667 * P.G;
    if (rank == 3 || p->state == FA_PLANE_STRAIN) {
      const double a = E / ((1.0 + nu) * (1.0 - 2.0 * nu));
      P.hd0 = a * (1.0 - nu); P.hd1 = a * nu; P.hd3 = a * (0.5 * (1.0 - 2.0 * nu));
    } else {
      const double a = E / (1.0 - nu * nu);
      P.hd0 = a; P.hd1 = a * nu; P.hd3 = a * 0.5 * (1.0 - nu);
    }
  }
  P.model = p->model; P.material = p->material; P.ndofpn = nd; P.nnode = nnode; P.nel = nel;
  const bool plain = p->model == FA_PHASE_FRACTURE || p->model == FA_MICRO_PHASE_FRACTURE;
  P.keep2 = plain && p->keep_off_diags;
  P.keep3 = plain ? (p->keep_off_diags != 0) : (pf ? 1 : 0);
  if (gd) {                                            // all four blocks (GradientDamageModel.cpp:577-583); c2 = c (:438, :563)
    P.keep2 = 1; P.keep3 = 1; P.c2 = p->length_scale; P.psi0 = 0.0; P.gcl = 0.0;
    P.soft = p->softening; P.kI = p->kappa_i; P.kC = p->kappa_c; P.sA = p->soft_alpha; P.sB = p->soft_beta;
    // the one-parameter laws borrow sA: hyperbolic b, expoEnergy s, multilinear kappaP
    if (p->softening == FA_SOFT_HYPERBOLIC) P.sA = p->soft_b;
    if (p->softening == FA_SOFT_EXPOENERGY) P.sA = p->soft_s;
    if (p->softening == FA_SOFT_MULTILINEAR) P.sA = p->kappa_p;
    P.eqv = p->eqv_strain; P.k2sig = (p->eqv_strain != FA_EQV_ENERGY) ? 1 : 0;
    P.pstress = (sh.rank == 2 && p->state == FA_PLANE_STRESS) ? 1 : 0; P.poiD = p->poisson / (p->poisson - 1.0);
    if (p->eqv_strain == FA_EQV_VONMISES) {             // VonMisesStrain::init (eta_, elasticMat_->getPoisson ()), :310
      const double k = p->eta, nu = p->poisson;
      P.vmA = (k - 1.0) / (1.0 - 2.0 * nu); P.vmB = 0.5 / k; P.vmC = 12.0 * k / (1.0 + 2.0 * nu + nu * nu);
    }
  }

  // ---- inverse connectivity, element-ascending per node ----
  std::vector<int32_t> n2e_ptr(nnode + 1, 0), n2e_ent((size_t)nel * npe);
  for (int64_t k = 0; k < nel * npe; k++) n2e_ptr[p->conn[k] + 1]++;
  int maxval = 0;
  for (int64_t n = 0; n < nnode; n++) { maxval = std::max(maxval, n2e_ptr[n + 1]); n2e_ptr[n + 1] += n2e_ptr[n]; }
  {
    std::vector<int32_t> fill(n2e_ptr.begin(), n2e_ptr.end() - 1);
    for (int64_t k = 0; k < nel * npe; k++) n2e_ent[fill[p->conn[k]]++] = (int32_t)k;   // k = e*npe + la
  }
  c->max_valence = maxval;

  // ---- sorted neighbour lists ----
  std::vector<int64_t> nb_ptr(nnode + 1, 0);
  {
    std::vector<int32_t> cnt(nnode, 0);
    std::atomic<bool> overflow{false};
    parallel_for(nnode, [&](int64_t b, int64_t e) {
      std::vector<int32_t> tmp;
      for (int64_t n = b; n < e; n++) {
        tmp.clear();
        for (int32_t k = n2e_ptr[n]; k < n2e_ptr[n + 1]; k++) {
          const int32_t* cn = p->conn + (int64_t)(n2e_ent[k] / npe) * npe;
          tmp.insert(tmp.end(), cn, cn + npe);
        }
        std::sort(tmp.begin(), tmp.end());
        cnt[n] = (int32_t)(std::unique(tmp.begin(), tmp.end()) - tmp.begin());
        if (cnt[n] > 255) overflow.store(true, std::memory_order_relaxed);
      }
    });
    if (overflow.load()) return fail(FA_ERR_UNSUPPORTED, "a node has more than 255 neighbours");
    for (int64_t n = 0; n < nnode; n++) nb_ptr[n + 1] = nb_ptr[n] + cnt[n];
  }
  std::vector<int32_t> nb((size_t)nb_ptr[nnode]);
  parallel_for(nnode, [&](int64_t b, int64_t e) {
    std::vector<int32_t> tmp;
    for (int64_t n = b; n < e; n++) {
      tmp.clear();
      for (int32_t k = n2e_ptr[n]; k < n2e_ptr[n + 1]; k++) {
        const int32_t* cn = p->conn + (int64_t)(n2e_ent[k] / npe) * npe;
        tmp.insert(tmp.end(), cn, cn + npe);
      }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      std::copy(tmp.begin(), tmp.end(), nb.begin() + nb_ptr[n]);
    }
  });
  std::vector<uint8_t> pos((size_t)nel * npe * npe);
  parallel_for(nel, [&](int64_t b, int64_t e) {
    for (int64_t el = b; el < e; el++) {
      const int32_t* cn = p->conn + el * npe;
      for (int la = 0; la < npe; la++) {
        const int32_t* lo = nb.data() + nb_ptr[cn[la]];
        const int32_t* hi = nb.data() + nb_ptr[cn[la] + 1];
        for (int lb = 0; lb < npe; lb++)
          pos[(el * npe + la) * npe + lb] = (uint8_t)(std::lower_bound(lo, hi, cn[lb]) - lo);
      }
    }
  });

  // ---- canonical pattern: dof = node*ndofpn + type, rows column-sorted; type-pair mask ----
  auto canonical = [&](int kind, int (&cnt)[4], int (&off)[4][4], std::vector<int64_t>& rp, std::vector<int32_t>* ci) {
    for (int i = 0; i < 4; i++) { cnt[i] = 0; for (int j = 0; j < 4; j++) off[i][j] = -1; }
    for (int i = 0; i < nd; i++) for (int j = 0; j < nd; j++) {
      bool on = true;
      if (kind == FA_PATTERN_WRITTEN && pf) {
        const bool iu = i < rank, ju = j < rank;
        if (iu && !ju) on = P.keep2; else if (!iu && ju) on = P.keep3;
      }
      if (on) off[i][j] = cnt[i]++;
    }
    rp.assign((size_t)nnode * nd + 1, 0);
    for (int64_t n = 0; n < nnode; n++)
      for (int i = 0; i < nd; i++) rp[n * nd + i + 1] = rp[n * nd + i] + (nb_ptr[n + 1] - nb_ptr[n]) * cnt[i];
    if (ci) {
      ci->resize((size_t)rp.back());
      parallel_for(nnode, [&](int64_t b, int64_t e) {
        for (int64_t n = b; n < e; n++)
          for (int i = 0; i < nd; i++) {
            int32_t* out = ci->data() + rp[n * nd + i];
            for (int64_t k = nb_ptr[n]; k < nb_ptr[n + 1]; k++)
              for (int j = 0; j < nd; j++) if (off[i][j] >= 0) *out++ = nb[k] * nd + j;
          }
      });
    }
  };

  std::vector<int64_t> rp; std::vector<int32_t> ci;
  int cnt[4] = { 0, 0, 0, 0 }, off[4][4] = { { 0 } };
  bool canonical_dofs = true;
  c->ndof = p->ndof > 0 ? p->ndof : nnode * nd;
  if (p->node_dofs) {
    for (int64_t k = 0; k < nnode * nd; k++) {
      if (p->node_dofs[k] < 0 || p->node_dofs[k] >= c->ndof)
        return fail(FA_ERR_ARG, "node_dofs[%lld] = %lld is outside [0, %lld)", (long long)k, (long long)p->node_dofs[k], (long long)c->ndof);
      if (p->node_dofs[k] != k) canonical_dofs = false;
    }
  } else if (c->ndof < nnode * nd) {
    return fail(FA_ERR_ARG, "ndof = %lld is smaller than nnode * dofs per node = %lld", (long long)c->ndof, (long long)(nnode * nd));
  }
  const int64_t* h_rp = nullptr; const int32_t* h_ci = nullptr;
  bool rows_ok = canonical_dofs && (c->ndof == nnode * nd);
  bool col_sorted = true;
  if (!p->row_ptr) {
    if (!canonical_dofs || c->ndof != nnode * nd)
      return fail(FA_ERR_ARG, "a custom dof map needs the caller's CSR pattern (row_ptr/col_idx)");
    if (p->pattern_kind != FA_PATTERN_FULL && p->pattern_kind != FA_PATTERN_WRITTEN) return fail(FA_ERR_ARG, "invalid pattern_kind");
    canonical(p->pattern_kind, cnt, off, rp, &ci);
    h_rp = rp.data(); h_ci = ci.data(); c->own_pattern = true;
  } else {
    if (!p->col_idx) return fail(FA_ERR_ARG, "row_ptr given without col_idx");
    h_rp = p->row_ptr; h_ci = p->col_idx;
    // the caller's arrays are not trusted: offsets start at 0 and never decrease, columns and dofs are in range
    if (h_rp[0] != 0) return fail(FA_ERR_ARG, "row_ptr[0] is %lld, not 0", (long long)h_rp[0]);
    for (int64_t r = 0; r < c->ndof; r++)
      if (h_rp[r + 1] < h_rp[r]) return fail(FA_ERR_ARG, "row_ptr decreases at row %lld", (long long)r);
    const int64_t nnz = h_rp[c->ndof];
    {
      std::atomic<int64_t> bad{-1};
      parallel_for(nnz, [&](int64_t b, int64_t e) {
        for (int64_t k = b; k < e; k++) if (h_ci[k] < 0 || h_ci[k] >= c->ndof) { bad.store(k, std::memory_order_relaxed); return; }
      });
      if (bad.load() >= 0) return fail(FA_ERR_ARG, "col_idx[%lld] = %d is outside [0, %lld)", (long long)bad.load(), h_ci[bad.load()], (long long)c->ndof);
    }
    if (rows_ok) {                       // is it one of the canonical blocked layouts?
      rows_ok = false;
      for (int kind = 0; kind < 2 && !rows_ok; kind++) {
        canonical(kind, cnt, off, rp, nullptr);
        if (rp.back() != nnz || std::memcmp(rp.data(), h_rp, sizeof(int64_t) * rp.size()) != 0) continue;
        canonical(kind, cnt, off, rp, &ci);
        rows_ok = std::memcmp(ci.data(), h_ci, sizeof(int32_t) * (size_t)nnz) == 0;
      }
    }
    if (!rows_ok) {
      for (int64_t r = 0; r < c->ndof && col_sorted; r++)
        for (int64_t k = h_rp[r] + 1; k < h_rp[r + 1]; k++) if (h_ci[k - 1] >= h_ci[k]) { col_sorted = false; break; }
      // Every entry the model writes (MatrixBuilder::addBlock of K1, K4 and, per keepOffDiags / model, K2 and K3:
      // PhaseFractureModel.cpp:736-746) must exist in the caller's pattern: a missing one is an error here, so that
      // the scatter kernels never have anything to drop (the reference's builder would add the entry or throw).
      std::atomic<int64_t> miss_row{-1}, miss_col{-1};
      auto dofof = [&](int64_t n, int i) -> int64_t { return p->node_dofs ? p->node_dofs[n * nd + i] : n * nd + i; };
      parallel_for(nnode, [&](int64_t b, int64_t e) {
        for (int64_t n = b; n < e && miss_row.load(std::memory_order_relaxed) < 0; n++)
          for (int i = 0; i < nd; i++) {
            const int64_t row = dofof(n, i);
            const int32_t* lo = h_ci + h_rp[row]; const int32_t* hi = h_ci + h_rp[row + 1];
            for (int64_t k = nb_ptr[n]; k < nb_ptr[n + 1]; k++)
              for (int j = 0; j < nd; j++) {
                const bool iu = i < rank, ju = j < rank;
                const bool written = (iu && ju) || (!iu && !ju) || (iu && !ju && P.keep2) || (!iu && ju && P.keep3);
                if (!written) continue;
                const int64_t col = dofof(nb[k], j);
                const bool found = col_sorted ? std::binary_search(lo, hi, (int32_t)col) : (std::find(lo, hi, (int32_t)col) != hi);
                if (!found) { miss_row.store(row, std::memory_order_relaxed); miss_col.store(col, std::memory_order_relaxed); return; }
              }
          }
      });
      if (miss_row.load() >= 0)
        return fail(FA_ERR_ARG, "the CSR pattern has no entry (%lld, %lld), which the model writes (a half-stored / symmetric "
                    "pattern is not supported: pass the full structure)", (long long)miss_row.load(), (long long)miss_col.load());
    }
  }
  c->nnz = h_rp[c->ndof];
  for (int i = 0; i < 4; i++) { P.cnt[i] = cnt[i]; for (int j = 0; j < 4; j++) P.off[i][j] = off[i][j]; }
  P.col_sorted = col_sorted;

  // ---- workgroup partition of the node range (row-gather kernel) ----
  std::vector<int32_t> wg_start;
  const int kBlock = fa::block_for(rank), kLds = fa::lds_for(rank);
  c->block = kBlock;
  P.canon = rows_ok ? 1 : 0;                        // canonical blocked layout: entry index from the pos table
  c->coop = coop_shape(p->shape);
  c->ngrid_coop = (int)((nel + fa::coop_epb(rank, npe, sh.nip) - 1) / fa::coop_epb(rank, npe, sh.nip));
  if (c->coop) rows_ok = false;
  if (rows_ok) {
    wg_start.push_back(0);
    int64_t thr = 0, lds = 0, nn = 0;
    for (int64_t n = 0; n < nnode; n++) {
      const int64_t v = n2e_ptr[n + 1] - n2e_ptr[n];
      const int64_t l = h_rp[(n + 1) * nd] - h_rp[n * nd];
      if (v > kBlock || l > kLds) { rows_ok = false; break; }
      if (thr + v > kBlock || lds + l > kLds || nn + 1 > kBlock) {
        wg_start.push_back((int32_t)n); thr = lds = nn = 0;
      }
      thr += v; lds += l; nn++;
    }
    wg_start.push_back((int32_t)nnode);
  }
  c->rows_path = rows_ok;
  // ---- lane records / workgroup descriptors of the row kernel ----
  std::vector<int32_t> lrec, wginfo;
  if (rows_ok) {
    const int LS = fa::lrec_stride(npe);
    const int64_t nwg = (int64_t)wg_start.size() - 1;
    lrec.assign(((size_t)nel * npe + 1) * LS, 0);          // + one padding record (idle lanes of an empty workgroup)
    wginfo.assign((size_t)nwg * fa::kWgInfo, 0);
    parallel_for(nwg, [&](int64_t wb, int64_t we) {
      for (int64_t w = wb; w < we; w++) {
        const int64_t n0 = wg_start[w], n1 = wg_start[w + 1];
        const int64_t base = h_rp[n0 * nd];
        int32_t* wi = wginfo.data() + w * fa::kWgInfo;
        wi[0] = (int32_t)n0; wi[1] = (int32_t)n1; wi[2] = n2e_ptr[n0]; wi[3] = n2e_ptr[n1] - n2e_ptr[n0];
        wi[4] = (int32_t)(uint32_t)(base & 0xFFFFFFFFLL); wi[5] = (int32_t)(base >> 32);
        wi[6] = (int32_t)(h_rp[n1 * nd] - base);
        for (int64_t n = n0; n < n1; n++) {
          const uint32_t w0 = (uint32_t)(h_rp[n * nd] - base) | ((uint32_t)(n - n0) << 16) |
                              ((uint32_t)(nb_ptr[n + 1] - nb_ptr[n]) << 24);
          for (int32_t t = n2e_ptr[n]; t < n2e_ptr[n + 1]; t++) {
            int32_t* r = lrec.data() + (size_t)t * LS;
            const int32_t ent = n2e_ent[t];
            const int64_t el = ent / npe; const int la = ent - (int32_t)el * npe;
            r[0] = ent;
            for (int b = 0; b < npe; b++) r[1 + b] = p->conn[el * npe + b];
            r[1 + npe] = (int32_t)w0;
            for (int b = 0; b < npe; b++)
              r[2 + npe + (b >> 2)] |= (int32_t)((uint32_t)pos[((size_t)el * npe + la) * npe + b] << (8 * (b & 3)));
          }
        }
      }
    });
  }
  // ---- pair tiling (row kernel with Gauss points shared by lane pairs; Quad4 / Hex8) ----
  std::vector<int32_t> lrec2, wginfo2;
  if (rows_ok && nel > 0 && sh.nip % 2 == 0 && (rank == 2 || dbg.pair_3d) && !dbg.no_pair) {   // Hex8: the pair kernel needs > 512 registers, slower
    const int LS = fa::lrec_stride2(npe), HALF = sh.nip / 2;
    // greedy tiles: an element with k nodes in the tile takes 2*ceil(k/2) lanes
    std::vector<int32_t> stamp(nel, -1), cnt_in(nel, 0), tstart;
    std::vector<int64_t> lane0;                       // first lane of every tile
    int64_t lanes_total = 0; int tile = 0; int64_t thr = 0, lds = 0, nn = 0;
    bool ok = true;
    tstart.push_back(0); lane0.push_back(0);
    for (int64_t n = 0; n < nnode && ok; n++) {
      const int64_t l = h_rp[(n + 1) * nd] - h_rp[n * nd];
      auto delta = [&](int t) {
        int64_t d = 0;
        for (int32_t k = n2e_ptr[n]; k < n2e_ptr[n + 1]; k++) {
          const int32_t el = n2e_ent[k] / npe;
          const int c0 = (stamp[el] == t) ? cnt_in[el] : 0;
          if ((c0 & 1) == 0) d += 2;
        }
        return d;
      };
      int64_t d = delta(tile);
      if (thr + d > kBlock || lds + l > kLds || nn + 1 > kBlock) {
        lanes_total += thr; tile++; tstart.push_back((int32_t)n); lane0.push_back(lanes_total);
        thr = lds = nn = 0;
        d = delta(tile);
        if (d > kBlock || l > kLds) { ok = false; break; }
      }
      for (int32_t k = n2e_ptr[n]; k < n2e_ptr[n + 1]; k++) {
        const int32_t el = n2e_ent[k] / npe;
        if (stamp[el] != tile) { stamp[el] = tile; cnt_in[el] = 0; }
        cnt_in[el]++;
      }
      thr += d; lds += l; nn++;
    }
    if (ok) {
      lanes_total += thr; tstart.push_back((int32_t)nnode); lane0.push_back(lanes_total);
      const int64_t nt = (int64_t)tstart.size() - 1;
      std::vector<int32_t> nlane2((size_t)nt);
      for (int64_t w = 0; w < nt; w++) nlane2[w] = (int32_t)(lane0[w + 1] - lane0[w]);
      // records padded to one workgroup's worth per tile: a lane's record address depends on blockIdx only, so its
      // load does not wait for the tile descriptor
      P.rec_padded = dbg.pad_rec;
      // persistent workgroups for the GET_MATRIX0 passes: every workgroup walks ~16 tiles and requests the lane records of the
      // next one before it streams the current image out (cfg3: 4.27 -> 3.94 ms; 31 tiles each 3.96, 4 tiles 4.18, 496 tiles
      // 4.18 -- few workgroups run in lockstep; meshes below 32 k tiles keep one workgroup per tile: 1000^2 0.303 against 0.317)
      P.pair_pers = dbg.pair_pers >= 0 ? (dbg.pair_pers & ~7) : (nt >= 32768 ? (int)(((nt / 16) + 7) & ~7) : 0);
      if (P.rec_padded) { for (int64_t w = 0; w <= nt; w++) lane0[w] = w * kBlock; lanes_total = nt * kBlock; }
      lrec2.assign((size_t)(lanes_total + 2) * LS, 0);
      wginfo2.assign((size_t)nt * fa::kWgInfo, 0);
      parallel_for(nt, [&](int64_t wb, int64_t we) {
        struct Ent { int32_t el, la, node; };
        std::vector<Ent> ents;
        for (int64_t w = wb; w < we; w++) {
          const int64_t n0 = tstart[w], n1 = tstart[w + 1];
          const int64_t base = h_rp[n0 * nd];
          ents.clear();
          for (int64_t n = n0; n < n1; n++)
            for (int32_t t = n2e_ptr[n]; t < n2e_ptr[n + 1]; t++) ents.push_back({ n2e_ent[t] / npe, n2e_ent[t] % npe, (int32_t)n });
          std::stable_sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.el < b.el; });
          int32_t* wi = wginfo2.data() + w * fa::kWgInfo;
          int64_t lane = lane0[w];
          wi[0] = (int32_t)n0; wi[1] = (int32_t)n1; wi[2] = (int32_t)(uint32_t)(lane & 0xFFFFFFFFLL); wi[7] = (int32_t)(lane >> 32);
          wi[3] = nlane2[w];
          wi[4] = (int32_t)(uint32_t)(base & 0xFFFFFFFFLL); wi[5] = (int32_t)(base >> 32);
          wi[6] = (int32_t)(h_rp[n1 * nd] - base);
          auto emit = [&](const Ent& en, bool owner, int half, uint32_t wmask) {
            int32_t* r = lrec2.data() + (size_t)lane * LS;
            const int64_t n = en.node;
            r[0] = en.el * npe + en.la;
            for (int b = 0; b < npe; b++) r[1 + b] = p->conn[(int64_t)en.el * npe + b];
            r[1 + npe] = (int32_t)((uint32_t)(h_rp[n * nd] - base) | ((uint32_t)(n - n0) << 16) |
                                   ((uint32_t)(nb_ptr[n + 1] - nb_ptr[n]) << 24));
            for (int b = 0; b < npe; b++)
              r[2 + npe + (b >> 2)] |= (int32_t)((uint32_t)pos[((size_t)en.el * npe + en.la) * npe + b] << (8 * (b & 3)));
            r[LS - 1] = (int32_t)((owner ? 1u : 0u) | ((uint32_t)half << 1) | (wmask << 2));
            lane++;
          };
          for (size_t i = 0; i < ents.size();) {
            size_t j = i;
            while (j < ents.size() && ents[j].el == ents[i].el) j++;
            // Gauss point ip is stored by the pair that holds local node ip % npe (every (element, node) lane exists
            // exactly once over all tiles), by its lane of half ip / HALF
            for (size_t q = i; q < j; q += 2) {
              const bool two = q + 1 < j;
              const Ent& A = ents[q]; const Ent& B = two ? ents[q + 1] : ents[q];
              uint32_t wm[2] = { 0, 0 };
              for (int ip = 0; ip < sh.nip; ip++) {
                const int lw = ip % npe;
                if (lw == A.la || (two && lw == B.la)) wm[ip / HALF] |= 1u << (ip % HALF);
              }
              emit(A, true, 0, wm[0]);
              emit(B, two, 1, wm[1]);
            }
            i = j;
          }
        }
      });
      c->pair_path = true; c->ntile2 = (int)nt;
    }
  }
  // ---- node patches (patch kernel) ----
  fa::PatchSet ps;
  if (rows_ok && nel > 0 && c->fam != fa::FAM_PF_MIEHE && dbg.use_patch && !gd) {               // experimental, opt-in
    int cap = (rank == 2) ? 6144 : 16384;                       // doubles of LDS for the CSR image
    if (dbg.patch_lds > 0) cap = std::max(64, dbg.patch_lds);
    const size_t lim = 160 * 1024;
    while (fa::patch_shmem_bytes(cap, nd) > lim) cap -= 256;
    c->patch_path = fa::build_patches(nnode, nel, npe, rank, p->coords, p->conn, n2e_ptr.data(), n2e_ent.data(),
                                      pos.data(), h_rp, nd, cap, fa::kMaxOwn, ps);
    if (c->patch_path) { c->npatch = ps.npatch; c->plds = ps.max_lds; c->lane_redundancy = ps.lane_redundancy; }
    if (c->patch_path && dbg.verbose)
      fprintf(stderr, "[falcon_assemble] %d patches, %.1f nodes / %.1f elements / %.2f runs per patch, LDS image <= %d doubles, "
              "Gauss-point redundancy %.3f\n", ps.npatch, (double)nnode / ps.npatch, (double)ps.nrec / ps.npatch,
              (double)ps.prun.size() / ps.npatch, ps.max_lds, ps.lane_redundancy);
  }
  // ---- node tiles (tile kernel: GET_MATRIX0 passes of the 3-D shapes) ----
  fa::TileSet ts;
  {
    // default: the higher-order shapes (they have no row kernel) and Hex8.
    // FA_TILES=0 / 1 turns the tile kernel off / on for every shape (experiments).
    const bool env = dbg.tiles >= 0;
    // (Hex8 with a general tangent -- MiehePhase, the micromorphic models -- as well since the 3 x 3 x 3 box tiles: its 32-double
    // records take the 1024-thread geometry, 6.55 ms against 8.8 ms for the row kernel on cfg4 / MiehePhase at 96^3)
    const bool wanted = env ? (dbg.tiles != 0) : (c->coop || p->shape == FA_HEX8_G2);

    if (P.canon && wanted && nel > 0) {
      const bool keepK = P.keep2 || P.keep3;
      const bool ext_state = pf && p->model != FA_PHASE_FRACTURE && p->model != FA_MICRO_PHASE_FRACTURE && p->model != FA_GRADIENT_DAMAGE;
      // lane order of the pair phase: sorted by element count with (a,b)/(b,a) sharing a lane, except on triangle
      // meshes, where (node, neighbour) order with the staged row write-out measured faster (Triangle6: 1.9 / 2.45 ms)
      const bool tri = p->shape == FA_TRI3_G1 || p->shape == FA_TRI3_G3 || p->shape == FA_TRI6_G3;
      const bool row_order = dbg.tile_order >= 0 ? dbg.tile_order != 0 : tri;
      P.tile_stage = row_order ? 1 : 0;
      // geometry: two 512-thread workgroups per CU with 80 KB each; element slots of >= kTileBigSlot doubles (27-point
      // Hex20, general tangents in 3-D) get one 1024-thread workgroup with the whole LDS (Hex20 Gauss3^3 at 64^3: 8.8
      // against 10.9 ms)
      const int recw = fa::tile_rec_doubles(rank, c->fam, keepK, npe);
      // Hex8 kernels that need more than 128 registers whatever the geometry (off-diagonal blocks kept: 147-154; GradientDamage:
      // 202-206) run one 512-thread workgroup per CU anyway, so they take the whole LDS and larger boxes (round 3: cfg4 with
      // keepOffDiags at 128^3 9.97 -> 7.10 ms, GradientDamage 12.4 -> 7.49 ms; boxes of 4 x 4 x 3 / 5 x 4 x 4 nodes 7.5 / 7.8 ms)
      const bool heavy8 = p->shape == FA_HEX8_G2 && c->fam != fa::FAM_PF_MIEHE && (keepK || gd);
      const bool big = (rank == 3 || dbg.tile_big == 2) && (dbg.tile_big >= 0 ? dbg.tile_big != 0 : (heavy8 || fa::tile_slot_doubles(sh.nip, recw) >= fa::kTileBigSlot));
      // FA_TILE_BIG=2: the big LDS budget with a 512-thread workgroup -- one workgroup per CU at two waves per SIMD, so the
      // kernel may use 256 registers (1024 threads pin it at 128: the general-tangent kernels then spill 0.5 KB per lane)
      // Default for the general-tangent family (MiehePhase, the micromorphic models): cfg4 / MiehePhase at 96^3 6.60 -> 5.14 ms.
      // Not for Hex20 elastic (5.71 against 4.96 ms at 64^3: no spills to lose there).
      const bool big512 = big && (dbg.tile_big >= 0 ? dbg.tile_big == 2 : (c->fam == fa::FAM_PF_MIEHE || heavy8));
      P.tile_block = (big && !big512) ? fa::kTileBlockBig : fa::kTileBlock;
      // one lane per unordered node pair across tiles (TileOpts::sym_all) unless the lanes are in row order
      fa::TileOpts topt;
      const int lane_mode = dbg.tile_lanes >= 0 ? dbg.tile_lanes : 0;
      topt.lane_rows = (!row_order && lane_mode >= 1) ? 1 : 0;
      topt.no_pair_share = !row_order && lane_mode == 3;
      topt.sym_all = !row_order && lane_mode < 2 && (dbg.tile_sym >= 0 ? dbg.tile_sym != 0 : true);
      P.tile_sym = topt.sym_all ? 1 : 0;
      topt.split = row_order ? 0 : (dbg.tile_split >= 0 ? dbg.tile_split : 2);
      // grouped pair lanes: the kernel's record format is fixed at compile time, so the tiling must follow it
      topt.group = fa::tile_group(rank, c->fam);
      P.tile_fu = (topt.group > 1 && (dbg.tile_fu >= 0 ? dbg.tile_fu != 0 : true)) ? 1 : 0;   // capability; decided per call
      if (topt.group > 1 && !topt.sym_all) return fail(FA_ERR_UNSUPPORTED, "FA_TILE_SYM=0 / FA_TILE_LANES>=2 / row order are not available for the grouped pair lanes of the 3-D elastic kernels");
      P.tile_split = topt.split > 0 ? 1 : 0;
      {
        // Hex8: boxes of 3 x 3 x 3 nodes bring 4 x 4 x 4 = 64 elements = 512 (element, Gauss point) lanes, one full round of the
        // workgroup, at 2.4 evaluations per point; the Morton cut (4 x 4 x 2 does not fit the LDS budget) gave 21.7 nodes / 57
        // elements at 2.7 (cfg4 at 128^3: 5.45 -> 5.25 ms, GET_INT_VECTOR 2.30 -> 2.07 ms)
        const int cellc = dbg.tile_cell >= 0 ? dbg.tile_cell : (p->shape == FA_HEX8_G2 ? ((heavy8 && big512) ? 444 : 333) : 0);
        if (rank == 3 && cellc >= 111) { topt.cell[0] = cellc / 100; topt.cell[1] = (cellc / 10) % 10; topt.cell[2] = cellc % 10; }
      }
      int budget = big ? fa::kTileLdsBig : fa::kTileLds;
      if (dbg.tile_budget > 0) budget = dbg.tile_budget;
      c->tile_path = fa::build_tiles(nnode, nel, npe, sh.nip, rank, nd, p->coords, p->conn, n2e_ptr.data(), n2e_ent.data(),
                                     nb_ptr.data(), nb.data(), recw, fa::tile_node_doubles(rank, pf) + (ext_state ? 2 : 0),
                                     budget, P.tile_block, row_order, ts, topt);
      if (c->tile_path) { c->ntile3 = ts.ntile; P.ntile3 = ts.ntile; P.tile_lds = ts.lds_bytes; }
      // (not for Hex20 Gauss2^3, the one configuration that measured slower with it: 3.57 -> 3.62 ms at 64^3)
      if (c->tile_path && (dbg.pad_tnodes >= 0 ? dbg.pad_tnodes != 0 : p->shape != FA_HEX20_G2)) { c->tn_stride = pad_tile_nodes(ts); P.tn_stride = c->tn_stride; }
      // GET_INT_VECTOR passes keep no records, so a tile set sized for GET_MATRIX0 pays its Gauss-point redundancy for
      // nothing: measured faster than the row / cooperative kernels up to ~3 evaluations per point (Hex20 Gauss3: 5)
      c->tile_force_only = c->tile_path && (env ? true : (ts.point_redundancy < 3.0 && !big));   // big geometry: one workgroup per CU, 1.1 against 0.67 ms for Hex20
      if (dbg.verbose)
        fprintf(stderr, "[falcon_assemble] tiles: %s, %d tiles, %.1f owned nodes / %.1f elements per tile, LDS <= %d bytes, "
                "Gauss-point redundancy %.3f\n", c->tile_path ? "ok" : "not built", ts.ntile,
                ts.ntile ? (double)nnode / ts.ntile : 0.0, ts.ntile ? (double)ts.telems.size() / ts.ntile : 0.0,
                ts.lds_bytes, ts.point_redundancy);
      if (dbg.verbose)
        fprintf(stderr, "[falcon_assemble] K blocks by destination: %lld into the rows of the lane's node, %lld mirrored into rows of the same "
                "tile, %lld mirrored into rows of other tiles\n", (long long)ts.blocks_direct, (long long)ts.blocks_mirror_own, (long long)ts.blocks_mirror_halo);
      if (dbg.verbose && topt.group > 1) {
        int64_t lanes = (int64_t)ts.tpairs.size() / 4, nbsum = 0, work = 0, hist[5] = { 0, 0, 0, 0, 0 };
        for (int64_t i = 0; i < lanes; i++) {
          const uint32_t x = ts.tpairs[(size_t)i * 4];
          const int nbq = (int)((x >> 16) & 0xFF), cq = (int)(x >> 24);
          nbsum += nbq; work += (int64_t)nbq * cq; hist[std::min(nbq, 4)]++;
        }
        fprintf(stderr, "[falcon_assemble] grouped pair lanes: %lld lanes, %.2f neighbours per lane (1: %lld, 2: %lld, 3: %lld, 4: %lld), "
                "%.1f (neighbour, element) units per element\n", (long long)lanes, lanes ? (double)nbsum / lanes : 0.0,
                (long long)hist[1], (long long)hist[2], (long long)hist[3], (long long)hist[4], (double)work / (double)nel);
      }
    }
  }
  // ---- tile kernel, GET_INT_VECTOR passes: tiles of their own.  The passes run phases 0-1 only and keep no Gauss-point
  //      records, so the LDS holds the nodal data of far larger tiles: fewer evaluations per point than on the GET_MATRIX0
  //      tiles (Hex20 Gauss3^3: 2.9) and two 512-thread workgroups per CU whatever the geometry of the matrix pass ----
  fa::TileSet tsT;
  if (c->tile_path && (dbg.tile_fset >= 0 ? dbg.tile_fset != 0 : true)) {
    const bool ext_state = pf && p->model != FA_PHASE_FRACTURE && p->model != FA_MICRO_PHASE_FRACTURE && p->model != FA_GRADIENT_DAMAGE;
    fa::TileOpts o;
    o.diag_only = true; o.order = 0; o.group = fa::tile_group(rank, c->fam);
    const int fblock = (rank == 3 && dbg.tile_fblock == fa::kTileBlockBig) ? fa::kTileBlockBig : fa::kTileBlock;
    o.max_el = dbg.tile_fset > 1 ? dbg.tile_fset : 512;               // (element, Gauss point) lanes: a few passes of the workgroup
    c->tileF.ok = fa::build_tiles(nnode, nel, npe, sh.nip, rank, nd, p->coords, p->conn, n2e_ptr.data(), n2e_ent.data(),
                                  nb_ptr.data(), nb.data(), 0, fa::tile_node_doubles(rank, pf) + (ext_state ? 2 : 0),
                                  fa::kTileLds, fblock, false, tsT, o);
    if (c->tileF.ok) { c->tileF.ntile = tsT.ntile; c->tileF.lds = tsT.lds_bytes; c->tileF.redundancy = tsT.point_redundancy; c->tileF.block = fblock; }
    if (c->tileF.ok && (dbg.pad_tnodes >= 0 ? dbg.pad_tnodes != 0 : true)) c->tn_strideF = pad_tile_nodes(tsT);
    if (dbg.verbose)
      fprintf(stderr, "[falcon_assemble] force-only tiles of the tile kernel: %s, %d tiles, %.1f owned nodes / %.1f elements per tile, LDS <= %d bytes, "
              "Gauss-point redundancy %.3f\n", c->tileF.ok ? "ok" : "not built", tsT.ntile, tsT.ntile ? (double)nnode / tsT.ntile : 0.0,
              tsT.ntile ? (double)tsT.telems.size() / tsT.ntile : 0.0, tsT.lds_bytes, tsT.point_redundancy);
    if (c->tileF.ok && dbg.tiles < 0) c->tile_force_only = true;
  }
  // ---- element-matrix kernel (Quad4, Triangle3/Gauss1): one tile set sized for the element-matrix slots of
  //      GET_MATRIX0 passes, one with f-only slots (larger tiles) for GET_INT_VECTOR passes ----
  fa::TileSet tsK, tsF;
  {
    const bool shape_ok = p->shape == FA_QUAD4_G2 || p->shape == FA_TRI3_G1;
    const bool wanted = (dbg.emat >= 0 ? dbg.emat != 0 : false) && !gd && !c->d_first_comp_needed;   // (no GradientDamage / Micro + AmorPhase instantiations of the opt-in kernel)
    if (shape_ok && wanted && P.canon && nel > 0) {
      const bool keepK = P.keep2 || P.keep3;
      const bool ext_state = pf && p->model != FA_PHASE_FRACTURE && p->model != FA_MICRO_PHASE_FRACTURE && p->model != FA_GRADIENT_DAMAGE;
      const int nodw = fa::tile_node_doubles(rank, pf) + (ext_state ? 2 : 0);
      const int budget = dbg.emat_lds > 0 ? dbg.emat_lds : 80 * 1024;
      auto lpe_of = [&](int want) { return (want == 2 && sh.nip % 2 == 0) ? 2 : 1; };
      fa::TileOpts o;
      o.no_tab = true; o.no_f_image = true; o.order = 2;
      c->ematK.lpe = lpe_of(dbg.emat_lpe > 0 ? dbg.emat_lpe : 2);
      o.slot_doubles = fa::emat_slot_doubles(rank, npe, pf, true, keepK); o.max_el = 256 / c->ematK.lpe;
      c->ematK.ok = fa::build_tiles(nnode, nel, npe, sh.nip, rank, nd, p->coords, p->conn, n2e_ptr.data(), n2e_ent.data(),
                                    nb_ptr.data(), nb.data(), 0, nodw, budget, 256, true, tsK, o);
      c->ematF.lpe = lpe_of(dbg.emat_lpe_f > 0 ? dbg.emat_lpe_f : 1);
      o.slot_doubles = fa::emat_slot_doubles(rank, npe, pf, false, false); o.max_el = 256 / c->ematF.lpe; o.diag_only = true;
      c->ematF.ok = fa::build_tiles(nnode, nel, npe, sh.nip, rank, nd, p->coords, p->conn, n2e_ptr.data(), n2e_ent.data(),
                                    nb_ptr.data(), nb.data(), 0, nodw, budget, 256, true, tsF, o);
      if (c->ematK.ok) { c->ematK.ntile = tsK.ntile; c->ematK.lds = tsK.lds_bytes; c->ematK.redundancy = tsK.point_redundancy; }
      if (c->ematF.ok) { c->ematF.ntile = tsF.ntile; c->ematF.lds = tsF.lds_bytes; c->ematF.redundancy = tsF.point_redundancy; }
      if (dbg.verbose)
        fprintf(stderr, "[falcon_assemble] element-matrix tiles: K %s %d tiles (%.1f nodes, LDS <= %d, redundancy %.3f, %d lanes/element), "
                "f %s %d tiles (%.1f nodes, LDS <= %d, redundancy %.3f, %d lanes/element)\n",
                c->ematK.ok ? "ok" : "not built", tsK.ntile, tsK.ntile ? (double)nnode / tsK.ntile : 0.0, tsK.lds_bytes, tsK.point_redundancy, c->ematK.lpe,
                c->ematF.ok ? "ok" : "not built", tsF.ntile, tsF.ntile ? (double)nnode / tsF.ntile : 0.0, tsF.lds_bytes, tsF.point_redundancy, c->ematF.lpe);
    }
  }
  // ---- force-vector kernel (GET_INT_VECTOR passes of Quad4 / Triangle3 / Hex8): needs the canonical dof numbering only ----
  fa::TileSet tsV;
  {
    const bool shape_ok = p->shape == FA_QUAD4_G2 || p->shape == FA_TRI3_G1 || p->shape == FA_HEX8_G2;
    const bool wanted = dbg.fvec >= 0 ? dbg.fvec != 0 : true;
    if (shape_ok && wanted && canonical_dofs && c->ndof == nnode * nd && nel > 0 && maxval <= 8) {
      const bool ext_state = pf && p->model != FA_PHASE_FRACTURE && p->model != FA_MICRO_PHASE_FRACTURE && p->model != FA_GRADIENT_DAMAGE;
      const int nodw = fa::tile_node_doubles(rank, pf) + (ext_state ? 2 : 0);
      // Hex8: 128 threads = the 125 elements around a Morton cell of 4 x 4 x 4 nodes (cfg4 at 128^3: 0.945 ms; 192 threads with
      // boxes of 5 x 5 x 4 nodes 1.01 ms, with the Morton cut 1.37 ms; the tile kernel's Gauss-point lanes 2.06 ms)
      c->fvec.block = (dbg.fvec_blk == 128 || dbg.fvec_blk == 192 || dbg.fvec_blk == 256 || dbg.fvec_blk == 320) ? dbg.fvec_blk : ((rank == 3 || p->shape == FA_TRI3_G1) ? 128 : 192);   // Triangle3: 0.229 against 0.242 ms on 8 M elements (round 3)
      fa::TileOpts o;
      o.no_tab = true; o.no_f_image = true; o.fvec = maxval; o.own_doubles = (maxval * nd) | 1; o.own_doubles_alt = nd; o.max_el = c->fvec.block;
      o.conn_pad4 = npe <= 4; o.max_all = 2 * c->fvec.block;
      if (rank == 3) {                                  // Hex8: more nodes than elements in a tile; box cells as for the tile kernel
        o.max_all = 0;
        const int cellc = dbg.fvec_cell >= 0 ? dbg.fvec_cell : 0;
        if (cellc >= 111) { o.cell[0] = cellc / 100; o.cell[1] = (cellc / 10) % 10; o.cell[2] = cellc % 10; }
      }
      o.sort_local = dbg.tile_sort >= 0 ? dbg.tile_sort != 0 : true;
      c->fvec.ok = fa::build_tiles(nnode, nel, npe, sh.nip, rank, nd, p->coords, p->conn, n2e_ptr.data(), n2e_ent.data(),
                                   nb_ptr.data(), nb.data(), 0, nodw, 48 * 1024, c->fvec.block, false, tsV, o);
      if (c->fvec.ok) {
        c->fvec.ntile = tsV.ntile; c->fvec.lds = tsV.lds_bytes; c->fvec.lds_alt = tsV.lds_bytes_alt; c->fvec.redundancy = tsV.point_redundancy; c->fvec.lpe = maxval;
        c->fvec.conn_stride = tsV.conn_stride;
        if (dbg.pad_tnodes >= 0 ? dbg.pad_tnodes != 0 : true) {
          c->fvec.tn_stride = pad_tile_nodes(tsV);
          c->fvec.te_stride = c->fvec.tn_stride ? pad_tile_elems(tsV, (npe + 3) / 4, tsV.conn_stride) : 0;
          if (!c->fvec.te_stride) c->fvec.tn_stride = 0;    // the kernel takes both or neither
        }
        // persistent LDS-DMA form: PhaseFracture / MicroPhaseFracture / LinearElasticity, workgroups of 192 or 256
        const bool model_ok = p->model == FA_LINEAR_ELASTICITY || p->model == FA_PHASE_FRACTURE || p->model == FA_MICRO_PHASE_FRACTURE;
        if (model_ok && (dbg.persist > 0) && (c->fvec.block == 192 || c->fvec.block == 256)) {
          c->fvec.maxn = (tsV.max_all + 63) / 64 * 64; c->fvec.maxe = (tsV.max_el + 63) / 64 * 64;
          const bool hist = p->model != FA_LINEAR_ELASTICITY;
          c->fvec.plds = 2 * fa::stage_bytes_host(pf, sh.nip, c->fvec.maxn, c->fvec.maxe, hist) + tsV.max_own * ((maxval * nd) | 1) * 8;
          c->fvec.persistent = c->fvec.plds <= 160 * 1024 && c->fvec.maxe <= c->fvec.block;
        }
      }
      if (dbg.verbose)
        fprintf(stderr, "[falcon_assemble] force-vector tiles: %s, %d tiles of <= %d elements (%.1f nodes, %.1f elements per tile), LDS <= %d bytes, "
                "Gauss-point redundancy %.3f\n", c->fvec.ok ? "ok" : "not built", tsV.ntile, c->fvec.block,
                tsV.ntile ? (double)nnode / tsV.ntile : 0.0, tsV.ntile ? (double)tsV.telems.size() / tsV.ntile : 0.0, tsV.lds_bytes, tsV.point_redundancy);
    }
  }
  // ---- element tiles: GET_INT_VECTOR passes evaluate every Gauss point once (node tiles: Quad4 1.14, Hex8 1.95, Hex20 2.2
  //      times) and add their f rows with global atomics.  Not for FA_DETERMINISTIC / FA_PART_* passes.  Force-vector kernel:
  //      every shape (cfg3 1.23 -> 1.05 ms, Miehe 2.35 -> 1.83, cfg4 at 128^3 0.96 -> 0.56); tile kernel: the 3-D shapes
  //      (Hex20 Gauss3^3 at 64^3 1.24 -> 0.68 ms; the 2-D shapes gain 0-17 % at tile sizes that differ per shape: FA_ELEM_TILES=2,
  //      profiles/r03_batch30-32) ----
  fa::TileSet tsU, tsTU;
  {
    const int want = dbg.elem_tiles >= 0 ? dbg.elem_tiles : 1;
    const bool ext_state = pf && p->model != FA_PHASE_FRACTURE && p->model != FA_MICRO_PHASE_FRACTURE && p->model != FA_GRADIENT_DAMAGE;
    const int nodw = fa::tile_node_doubles(rank, pf) + (ext_state ? 2 : 0);
    if (want && nel > 0 && c->fvec.ok && !c->fvec.persistent) {
      c->fvecU.block = c->fvec.block;
      c->fvecU.ok = fa::build_element_tiles(nnode, nel, npe, sh.nip, rank, nd, p->coords, p->conn, nodw, 48 * 1024, c->fvecU.block, 0,
                                            false, npe <= 4, tsU);
      if (c->fvecU.ok) {
        c->fvecU.ntile = tsU.ntile; c->fvecU.lds = c->fvecU.lds_alt = tsU.lds_bytes; c->fvecU.redundancy = 1.0; c->fvecU.lpe = maxval;
        c->fvecU.conn_stride = tsU.conn_stride;
        if (dbg.pad_tnodes >= 0 ? dbg.pad_tnodes != 0 : true) {
          c->fvecU.tn_stride = pad_tile_nodes(tsU);
          c->fvecU.te_stride = c->fvecU.tn_stride ? pad_tile_elems(tsU, (npe + 3) / 4, tsU.conn_stride) : 0;
          if (!c->fvecU.te_stride) c->fvecU.tn_stride = 0;
        }
      }
      if (dbg.verbose)
        fprintf(stderr, "[falcon_assemble] force-vector ELEMENT tiles: %s, %d tiles of <= %d elements (%.1f nodes per tile), LDS <= %d bytes\n",
                c->fvecU.ok ? "ok" : "not built", tsU.ntile, c->fvecU.block, tsU.ntile ? (double)tsU.tnodes.size() / tsU.ntile : 0.0, tsU.lds_bytes);
    } else if (want && (rank == 3 || want > 1) && nel > 0 && c->tileF.ok && c->tile_force_only) {
      c->tileU.block = c->tileF.block;
      const int max_el = dbg.elem_tile_el > 0 ? dbg.elem_tile_el : 64;
      c->tileU.ok = fa::build_element_tiles(nnode, nel, npe, sh.nip, rank, nd, p->coords, p->conn, nodw, fa::kTileLds, max_el, 0,
                                            true, false, tsTU);
      if (c->tileU.ok) {
        c->tileU.ntile = tsTU.ntile; c->tileU.lds = tsTU.lds_bytes; c->tileU.redundancy = 1.0;
        if (dbg.pad_tnodes >= 0 ? dbg.pad_tnodes != 0 : true) c->tileU.tn_stride = pad_tile_nodes(tsTU);
      }
      if (dbg.verbose)
        fprintf(stderr, "[falcon_assemble] tile kernel, ELEMENT tiles of the force-only passes: %s, %d tiles of <= %d elements (%.1f nodes per tile), LDS <= %d bytes\n",
                c->tileU.ok ? "ok" : "not built", tsTU.ntile, max_el, tsTU.ntile ? (double)tsTU.tnodes.size() / tsTU.ntile : 0.0, tsTU.lds_bytes);
    }
  }
  c->nwg = rows_ok ? (int)wg_start.size() - 1 : 0;
  c->ngrid_generic = (int)((nel * npe + kBlock - 1) / kBlock);

  // ---- upload ----
  FA_ON_DEVICE(p->device);
  fa_status st;
  if ((st = resolve_scratch(p->device))) return st;
  { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, p->device) == hipSuccess && v > 0) c->ncu = v; }   // set-up of fa_assemble_noalloc: once per device, device current
  if ((st = upload(&c->d_conn, p->conn, (size_t)nel * npe))) return st;
  if ((st = upload(&c->d_coords, p->coords, (size_t)nnode * rank))) return st;
  if ((st = upload(&c->d_row_ptr, h_rp, (size_t)c->ndof + 1))) return st;
  if ((st = upload(&c->d_col_idx, h_ci, (size_t)c->nnz))) return st;
  if (!canonical_dofs) { if ((st = upload(&c->d_node_dofs, p->node_dofs, (size_t)nnode * nd))) return st; }
  if ((st = upload(&c->d_n2e_ptr, n2e_ptr.data(), n2e_ptr.size()))) return st;
  if ((st = upload(&c->d_n2e_ent, n2e_ent.data(), n2e_ent.size()))) return st;
  if ((st = upload(&c->d_pos, pos.data(), pos.size()))) return st;
  if (rows_ok) {
    if ((st = upload(&c->d_wg_start, wg_start.data(), wg_start.size()))) return st;
    if ((st = upload(&c->d_lrec, lrec.data(), lrec.size()))) return st;
    if ((st = upload(&c->d_wginfo, wginfo.data(), wginfo.size()))) return st;
  }
  if (c->pair_path) {
    if ((st = upload(&c->d_lrec2, lrec2.data(), lrec2.size()))) return st;
    if ((st = upload(&c->d_wginfo2, wginfo2.data(), wginfo2.size()))) return st;
  }
  if (c->patch_path) {
    if ((st = upload(&c->d_pn_ptr, ps.pn_ptr.data(), ps.pn_ptr.size()))) return st;
    if ((st = upload(&c->d_pe_ptr, ps.pe_ptr.data(), ps.pe_ptr.size()))) return st;
    if ((st = upload(&c->d_pr_ptr, ps.pr_ptr.data(), ps.pr_ptr.size()))) return st;
    if ((st = upload(&c->d_pnode, ps.pnode.data(), ps.pnode.size()))) return st;
    if ((st = upload(&c->d_poff, ps.poff.data(), ps.poff.size()))) return st;
    if ((st = upload(&c->d_prec, ps.prec.data(), ps.prec.size()))) return st;
    if ((st = upload(&c->d_prun, ps.prun.data(), ps.prun.size()))) return st;
  }
  {
    std::vector<double> tab;
    tab.insert(tab.end(), sh.N.begin(), sh.N.end());
    tab.insert(tab.end(), sh.dN.begin(), sh.dN.end());
    tab.insert(tab.end(), sh.w.begin(), sh.w.end());
    if ((st = upload(&c->d_tab, tab.data(), tab.size()))) return st;
  }
  if (c->tile_path) {
    if ((st = upload(&c->d_tinfo, ts.tinfo.data(), ts.tinfo.size()))) return st;
    if ((st = upload(&c->d_tnodes, ts.tnodes.data(), ts.tnodes.size()))) return st;
    if ((st = upload(&c->d_telems, ts.telems.data(), ts.telems.size()))) return st;
    if ((st = upload(&c->d_tconn, ts.tconn.data(), ts.tconn.size()))) return st;
    if ((st = upload(&c->d_tpairs, ts.tpairs.data(), ts.tpairs.size()))) return st;
    if ((st = upload(&c->d_tents, ts.tents.data(), std::max<size_t>(ts.tents.size(), 1)))) return st;
  }
  auto upload_set = [&](TileDev& d, const fa::TileSet& t) -> fa_status {
    fa_status s2;
    if ((s2 = upload(&d.tinfo, t.tinfo.data(), t.tinfo.size()))) return s2;
    if ((s2 = upload(&d.tnodes, t.tnodes.data(), t.tnodes.size()))) return s2;
    if ((s2 = upload(&d.telems, t.telems.data(), t.telems.size()))) return s2;
    if ((s2 = upload(&d.tconn, t.tconn.data(), t.tconn.size()))) return s2;
    if ((s2 = upload(&d.tpairs, t.tpairs.data(), t.tpairs.size()))) return s2;
    return upload(&d.tents, t.tents.data(), std::max<size_t>(t.tents.size(), 1));
  };
  if (c->tileF.ok && (st = upload_set(c->tileF, tsT))) return st;
  if (c->ematK.ok && (st = upload_set(c->ematK, tsK))) return st;
  if (c->ematF.ok && (st = upload_set(c->ematF, tsF))) return st;
  if (c->fvec.ok) {
    if ((st = upload_set(c->fvec, tsV))) return st;
    if ((st = upload(&c->fvec.eslot, tsV.eslot.data(), tsV.eslot.size()))) return st;
    if ((st = upload(&c->fvec.nval, tsV.nval.data(), tsV.nval.size()))) return st;
  }
  if (c->fvecU.ok) {
    if ((st = upload_set(c->fvecU, tsU))) return st;
    if ((st = upload(&c->fvecU.eslot, tsU.eslot.data(), tsU.eslot.size()))) return st;
    if ((st = upload(&c->fvecU.nval, tsU.nval.data(), tsU.nval.size()))) return st;
  }
  if (c->tileU.ok && (st = upload_set(c->tileU, tsTU))) return st;
  const int64_t nipt = nel * sh.nip;
  if (p->model == FA_PHASE_FRACTURE || p->model >= FA_MICRO_PHASE_FRACTURE) {
    for (int k = 0; k < 2; k++) HIP_TRY(hipMalloc((void**)&c->d_hist[k], std::max<int64_t>(nipt, 1) * sizeof(double)));
    // preHist_.H0_ = Psi0_; newHist_.H0_ = 0 (PhaseFractureModel.cpp:359-360)
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, c->d_hist[0], nipt, p->model == FA_PHASE_FRACTURE ? P.psi0 : 0.0);   // preHist_.phasef_ = 0 (MicroPhaseFractureModel.cpp:359)
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, c->d_hist[1], nipt, 0.0);
  }
  c->max_parts = std::max(std::max(std::max(std::max(std::max(c->nwg, c->ntile2), c->ngrid_generic), c->npatch), c->ngrid_coop), c->ntile3);
  c->max_parts = std::max(c->max_parts, std::max(std::max(c->ematK.ntile, c->ematF.ntile), std::max(c->fvec.ntile, c->tileF.ntile)));
  c->max_parts = std::max(c->max_parts, std::max(c->fvecU.ntile, c->tileU.ntile));
  if (p->model >= FA_MICRO_PHASE_FRACTURE && p->model <= FA_MICRO_PHASE_FRACTURE_EXTIT && p->material == FA_AMOR_PHASE) {
    HIP_TRY(hipMalloc((void**)&c->d_first_comp, sizeof(unsigned long long)));
    if (c->dbg.micro_mask) HIP_TRY(hipMalloc((void**)&c->d_comp_mask, sizeof(uint32_t) * std::max<int64_t>(nel, 1)));
    if (dbg.verbose)
      fprintf(stderr, "[falcon_assemble] MicroPhaseFracture* + AmorPhase: stiffN of a tension point is K PVol once the element loop has "
              "passed a compression point and ZERO before (the reference reads an uninitialised array there); K and f_int therefore "
              "depend on the element numbering and on the partition (include/falcon_assemble.h)\n");
  }
  if (p->model >= FA_MICRO_PHASE_FRACTURE && p->model <= FA_MICRO_PHASE_FRACTURE_EXTIT) {
    HIP_TRY(hipMalloc((void**)&c->d_stale, sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->d_stale, 0, sizeof(unsigned long long)));
  }
  // the host getters wait for this event instead of synchronising a caller's stream that may no longer exist
  HIP_TRY(hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming));
  if (p->model == FA_PHASE_FRACTURE_EXTIT || p->model == FA_MICRO_PHASE_FRACTURE_EXTIT)
    HIP_TRY(hipMalloc((void**)&c->d_err_part, sizeof(double) * std::max(c->max_parts, 1)));
  HIP_TRY(hipDeviceSynchronize());

  P.conn = c->d_conn; P.coords = c->d_coords; P.row_ptr = c->d_row_ptr; P.col_idx = c->d_col_idx;
  P.node_dofs = c->d_node_dofs; P.n2e_ptr = c->d_n2e_ptr; P.n2e_ent = c->d_n2e_ent; P.pos = c->d_pos;
  P.wg_start = c->d_wg_start; P.lrec = c->d_lrec; P.wginfo = c->d_wginfo; P.lrec2 = c->d_lrec2; P.wginfo2 = c->d_wginfo2; P.ntile2 = c->ntile2; P.tab = c->d_tab; P.max_valence = c->max_valence;
  P.tinfo = c->d_tinfo; P.tnodes = c->d_tnodes; P.telems = c->d_telems; P.tconn = c->d_tconn; P.tpairs = c->d_tpairs; P.tents = c->d_tents;
  P.pn_ptr = c->d_pn_ptr; P.pe_ptr = c->d_pe_ptr; P.pr_ptr = c->d_pr_ptr; P.pnode = c->d_pnode; P.poff = c->d_poff;
  P.prec = c->d_prec; P.prun = c->d_prun; P.plds = c->plds;
  return FA_OK;
}

extern "C" fa_status fa_create(const fa_problem* p, fa_ctx** out) {
  if (!p || !out) return fail(FA_ERR_ARG, "null argument");
  *out = nullptr;
  fa_ctx* c = new fa_ctx();
  fa_status st = create_impl(p, c);
  if (st != FA_OK) { std::string keep = g_err; fa_destroy(c); g_err = keep; return st; }
  *out = c;
  return FA_OK;
}

extern "C" fa_status fa_shape_info(int32_t shape, int32_t* rank, int32_t* npe, int32_t* nip) {
  ShapeTab sh;
  if (!make_shape(shape, sh)) return fail(FA_ERR_ARG, "unknown shape id %d", shape);
  if (rank) *rank = sh.rank;
  if (npe) *npe = sh.npe;
  if (nip) *nip = sh.nip;
  return FA_OK;
}

extern "C" fa_status fa_shape_tables(int32_t shape, double* N, double* dN, double* w) {
  ShapeTab sh;
  if (!make_shape(shape, sh)) return fail(FA_ERR_ARG, "unknown shape id %d", shape);
  if (N) std::copy(sh.N.begin(), sh.N.end(), N);
  if (dN) std::copy(sh.dN.begin(), sh.dN.end(), dN);
  if (w) std::copy(sh.w.begin(), sh.w.end(), w);
  return FA_OK;
}

extern "C" int64_t fa_ndof(const fa_ctx* c) { return c->ndof; }
extern "C" int64_t fa_nnz(const fa_ctx* c) { return c->nnz; }
extern "C" int64_t fa_nip_total(const fa_ctx* c) { return c->nel * c->nip; }
extern "C" int32_t fa_ndof_per_node(const fa_ctx* c) { return c->ndofpn; }
extern "C" int32_t fa_strain_rows(const fa_ctx* c) { return c->srows; }
extern "C" int32_t fa_uses_row_kernel(const fa_ctx* c) { return c->rows_path ? 1 : 0; }
extern "C" double  fa_psi0(const fa_ctx* c) { return c->P.psi0; }
extern "C" int32_t fa_patch_stats(const fa_ctx* c, double* red) { if (red) *red = c->lane_redundancy; return c->patch_path ? c->npatch : 0; }

extern "C" fa_status fa_pattern_host(const fa_ctx* c, int64_t* row_ptr, int32_t* col_idx) {
  FA_ON_DEVICE(c->prob.device);
  if (row_ptr) HIP_TRY(hipMemcpy(row_ptr, c->d_row_ptr, sizeof(int64_t) * (c->ndof + 1), hipMemcpyDeviceToHost));
  if (col_idx) HIP_TRY(hipMemcpy(col_idx, c->d_col_idx, sizeof(int32_t) * c->nnz, hipMemcpyDeviceToHost));
  return FA_OK;
}

extern "C" fa_status fa_pattern_device(const fa_ctx* c, const int64_t** row_ptr, const int32_t** col_idx) {
  if (row_ptr) *row_ptr = c->d_row_ptr;
  if (col_idx) *col_idx = c->d_col_idx;
  return FA_OK;
}

__global__ void narrow_offsets_kernel(int64_t n, const int64_t* __restrict__ in, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)in[i];
}

extern "C" fa_status fa_pattern_device32(fa_ctx* c, const int32_t** row_ptr, const int32_t** col_idx) {
  if (!c) return fail(FA_ERR_ARG, "null argument");
  if (c->nnz > 2147483647LL)
    return fail(FA_ERR_UNSUPPORTED, "fa_pattern_device32: %lld stored entries do not fit 32-bit offsets (cuDSSSolver.cpp:655 has the same limit on the matrix size)", (long long)c->nnz);
  FA_ON_DEVICE(c->prob.device);
  if (row_ptr) {
    if (!c->d_row_ptr32) {
      HIP_TRY(hipMalloc((void**)&c->d_row_ptr32, sizeof(int32_t) * (c->ndof + 1)));
      hipLaunchKernelGGL(narrow_offsets_kernel, dim3((unsigned)((c->ndof + 256) / 256)), dim3(256), 0, nullptr, c->ndof + 1, c->d_row_ptr, c->d_row_ptr32);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(nullptr));
    }
    *row_ptr = c->d_row_ptr32;
  }
  if (col_idx) *col_idx = c->d_col_idx;
  return FA_OK;
}

extern "C" int32_t fa_device(const fa_ctx* c) { return c ? c->prob.device : -1; }

// noalloc: the call must not allocate -- the strain cache has to exist already and the errExt_ partials go to the
// file-scope scratch array (through its resolved device address) instead of the context's buffer
static fa_status assemble_impl(fa_ctx* c, const fa_assemble_args* a, bool noalloc) {
  if (!c || !a) return fail(FA_ERR_ARG, "null argument");
  if (!a->state || !a->f_int) return fail(FA_ERR_ARG, "state and f_int are required");
  const int model = c->prob.model;
  if ((model == FA_PHASE_FRACTURE_EXT || model == FA_PHASE_FRACTURE_EXTIT) && (!a->state0 || !a->state00))
    return fail(FA_ERR_ARG, "PhaseFractureExt/ExtIt need state0 and state00 (stateExt_)");
  if ((model == FA_MICRO_PHASE_FRACTURE_EXT || model == FA_MICRO_PHASE_FRACTURE_EXTIT) && !a->state00)
    return fail(FA_ERR_ARG, "MicroPhaseFractureExt/ExtIt need state00 (stateExt_)");
  // LinearElasticityModel answers GET_MATRIX0 only (LinearElasticityModel.cpp:226-246); force-only is still
  // served here because the kernel supports it.
  FA_ON_DEVICE(c->prob.device);
  hipStream_t st = (hipStream_t)a->stream;
  fa::Params P = c->P;
  P.state = a->state; P.state0 = a->state0; P.stateX = a->state00; P.dt = a->dt; P.dt0 = a->dt0;
  P.K = a->K_values; P.f = a->f_int;
  P.hist_pre = c->d_hist[0]; P.hist_new = c->d_hist[1];
  P.overwrite = (a->flags & FA_OVERWRITE) ? 1 : 0;
  P.strain = nullptr;
  if (a->flags & FA_STORE_STRAIN) {
    if (!c->d_strain) {
      if (noalloc) return fail(FA_ERR_ARG, "fa_assemble_noalloc: FA_STORE_STRAIN needs the strain cache of an earlier fa_assemble call with FA_STORE_STRAIN (allocating it here is not allowed)");
      HIP_TRY(hipMalloc((void**)&c->d_strain, sizeof(double) * std::max<int64_t>(1, c->nel * c->nip * c->srows)));
    }
    P.strain = c->d_strain;
  }
  const bool want_err = (model == FA_PHASE_FRACTURE_EXTIT || model == FA_MICRO_PHASE_FRACTURE_EXTIT) && a->err_ext;
  double* err_part = c->d_err_part;
  if (want_err && noalloc) {
    if (c->max_parts > kNoAllocMaxParts)
      return fail(FA_ERR_ARG, "fa_assemble_noalloc: this mesh needs %d partial sums of errExt_, the static scratch holds %d", c->max_parts, kNoAllocMaxParts);
    fa_status rs = resolve_scratch(c->prob.device);
    if (rs != FA_OK) return rs;
    err_part = g_scratch_ptr[c->prob.device];
    // every call on this GPU shares the array and finds the last call's contents: clear what this call will use
    HIP_TRY(hipMemsetAsync(err_part, 0, sizeof(double) * (size_t)c->max_parts, st));
  }
  P.err_part = want_err ? err_part : nullptr;
  const bool want_K = a->K_values != nullptr;
  const bool keep = want_K && (P.keep2 || P.keep3);
  // bitwise reproducibility exists on the ordered row kernel only: the tile / cooperative / generic kernels add f_int
  // (and, the last two, K) with atomics whose order varies from run to run
  const bool elem_ok = !(a->flags & (FA_FORCE_GENERIC | FA_DETERMINISTIC | FA_PART_INTERFACE | FA_PART_INTERIOR));   // element tiles: atomic f rows, no interface lists
  const TileDev* emat = (a->flags & FA_FORCE_GENERIC) ? nullptr : (a->K_values ? ((c->ematK.ok && !c->P.k2sig) ? &c->ematK : nullptr) : ((c->fvecU.ok && elem_ok) ? &c->fvecU : (c->fvec.ok ? &c->fvec : (c->ematF.ok ? &c->ematF : nullptr))));
  if ((a->flags & FA_DETERMINISTIC) && !emat && (!c->rows_path || (a->flags & FA_FORCE_GENERIC)))
    return fail(FA_ERR_UNSUPPORTED, "FA_DETERMINISTIC needs the row kernel (Triangle3/Gauss1, Quad4, Hex8 on a canonical node-major "
                                    "pattern): this context runs an atomic-scatter kernel");
  int path = c->coop ? fa::PATH_COOP : fa::PATH_GENERIC;
  if (c->rows_path && !(a->flags & FA_FORCE_GENERIC))
    path = (a->flags & FA_DETERMINISTIC) ? fa::PATH_ROWS_ORDERED : (c->patch_path ? fa::PATH_PATCH : (c->pair_path ? fa::PATH_ROWS_PAIR : fa::PATH_ROWS));
  // (the tile kernel's pair lanes read K2 and K3 from one recorded vector: not for GradientDamage with a Mazars strain)
  if (c->tile_path && (want_K || c->tile_force_only) && !(want_K && P.k2sig) && !(a->flags & (FA_FORCE_GENERIC | FA_DETERMINISTIC))) path = fa::PATH_TILES;
  // 3-D elastic tiles: f_int from the pair lanes (atomic adds to f) when K is assembled and no strain cache is asked for
  P.tile_fu = (path == fa::PATH_TILES && c->P.tile_fu && want_K && !(a->flags & FA_STORE_STRAIN)) ? 1 : 0;
  // interface / interior part of a pass (fa_set_interface): the launch runs a list of tiles instead of all of them
  const int part = (a->flags & FA_PART_INTERFACE) ? 0 : ((a->flags & FA_PART_INTERIOR) ? 1 : -1);
  if ((a->flags & FA_PART_INTERFACE) && (a->flags & FA_PART_INTERIOR)) return fail(FA_ERR_ARG, "FA_PART_INTERFACE and FA_PART_INTERIOR exclude each other");
  if (part >= 0) {
    if (!c->parts_set) return fail(FA_ERR_ARG, "FA_PART_* needs fa_set_interface first");
    if (c->d_first_comp) return fail(FA_ERR_UNSUPPORTED, "FA_PART_*: MicroPhaseFracture* with AmorPhase needs the first compression point of the whole pass");
    P.tile_fu = 0;                                     // f_int from the Gauss-point lanes: the K u form zeroes f for the whole mesh first
  }
  const bool tile_fset = path == fa::PATH_TILES && !want_K && c->tileF.ok && !emat;
  P.f_atomic = 0;
  const bool tile_uset = tile_fset && c->tileU.ok && elem_ok;
  const int tf_ntile = tile_uset ? c->tileU.ntile : c->tileF.ntile;
  if (tile_fset) {                                     // GET_INT_VECTOR pass of the tile kernel: its own, larger tiles (or element tiles)
    const TileDev& tf = tile_uset ? c->tileU : c->tileF;
    P.tinfo = tf.tinfo; P.tnodes = tf.tnodes; P.telems = tf.telems; P.tconn = tf.tconn;
    P.tpairs = tf.tpairs; P.tents = tf.tents; P.ntile3 = tf.ntile; P.tile_lds = tf.lds;
    P.tile_block = tf.block; P.tile_sym = 0; P.tile_stage = 0; P.tile_split = 0; P.tn_stride = tile_uset ? tf.tn_stride : c->tn_strideF;
    P.f_atomic = tile_uset ? 1 : 0;
  }
  if (emat) {                                          // element-matrix kernel: no atomics, so it also serves FA_DETERMINISTIC
    P.tn_stride = emat->tn_stride; P.te_stride = emat->te_stride;
    path = (emat == &c->fvec || emat == &c->fvecU) ? fa::PATH_FVEC : fa::PATH_EMAT;
    P.eslot = emat->eslot; P.nval = emat->nval; P.fvec_val = emat->lpe;
    P.tinfo = emat->tinfo; P.tnodes = emat->tnodes; P.telems = emat->telems; P.tconn = emat->tconn;
    P.tpairs = emat->tpairs; P.tents = emat->tents; P.ntile3 = emat->ntile; P.tile_lds = emat->lds;
    P.tile_block = emat->block; P.emat_lpe = emat->lpe;
    P.conn_stride = emat->conn_stride; P.persistent = 0;
    // force-vector kernel: FA_FVEC_ATOMIC=1 adds into one LDS row per node with atomics (a quarter of the buffer, one more
    // workgroup per CU) instead of ordered per-element rows: measured 1.25 against 1.27 ms on cfg3, not worth giving up the
    // repeatable sum, so it stays opt-in
    P.fvec_atomic = (emat == &c->fvec && !(a->flags & FA_DETERMINISTIC) && (c->dbg.fvec_atomic > 0) && !emat->persistent) ? 1 : 0;
    if (emat == &c->fvecU) { P.fvec_atomic = 1; P.f_atomic = 1; }   // element tiles: one LDS row per tile node, added to f_int with global atomics
    if (P.fvec_atomic) P.tile_lds = emat->lds_alt;
    if (emat->persistent) {
      P.persistent = 1; P.st_maxn = emat->maxn; P.st_maxe = emat->maxe; P.tile_lds = emat->plds;
    }
  }
  if ((P.tile_fu || P.f_atomic) && P.overwrite && c->nel > 0) HIP_TRY(hipMemsetAsync(a->f_int, 0, sizeof(double) * c->ndof, st));
  if (c->nel == 0) {
    if (P.overwrite) {
      if (want_K) HIP_TRY(hipMemsetAsync(a->K_values, 0, sizeof(double) * c->nnz, st));
      HIP_TRY(hipMemsetAsync(a->f_int, 0, sizeof(double) * c->ndof, st));
    }
    if (want_err) HIP_TRY(hipMemsetAsync(a->err_ext, 0, sizeof(double), st));
    HIP_TRY(hipEventRecord(c->done_ev, st));
    return FA_OK;
  }
  int grid;
  P.tile_list = nullptr;
  if (part >= 0) {
    const PartLists* pl = nullptr;
    if (path == fa::PATH_ROWS || path == fa::PATH_ROWS_ORDERED) pl = &c->part_rows;
    else if (path == fa::PATH_ROWS_PAIR) pl = &c->part_pair;
    else if (path == fa::PATH_TILES) pl = tile_fset ? &c->part_tileF : &c->part_tiles;
    else if (path == fa::PATH_FVEC && !emat->persistent) pl = &c->part_fvec;
    if (!pl) return fail(FA_ERR_UNSUPPORTED, "FA_PART_*: this context runs a kernel that is not organised in node tiles (generic / cooperative / element-matrix / patch / persistent kernels)");
    if (pl->n[part] == 0) {                            // nothing to do in this part
      if (want_err && part == 1) HIP_TRY(hipMemsetAsync(a->err_ext, 0, sizeof(double), st));
      HIP_TRY(hipEventRecord(c->done_ev, st));
      return FA_OK;
    }
    P.tile_list = pl->d[part];
    P.ntile = P.ntile2 = P.ntile3 = pl->n[part];
    grid = (pl->n[part] + 7) / 8 * 8;
  }
  else if (path == fa::PATH_PATCH) grid = c->npatch;
  else if (path == fa::PATH_ROWS || path == fa::PATH_ROWS_ORDERED) { P.ntile = c->nwg; grid = (c->nwg + 7) / 8 * 8; }
  else if (path == fa::PATH_ROWS_PAIR) grid = (c->ntile2 + 7) / 8 * 8;
  else if (path == fa::PATH_TILES) grid = ((tile_fset ? tf_ntile : c->ntile3) + 7) / 8 * 8;
  else if (path == fa::PATH_EMAT || path == fa::PATH_FVEC) {
    grid = (emat->ntile + 7) / 8 * 8;
    if (emat->persistent) {                            // as many workgroups as fit the GPU at once (LDS-bound), a multiple of 8
      const int per_cu = std::max(1, std::min(4, (160 * 1024) / std::max(emat->plds, 1)));
      grid = std::min(grid, std::max(8, c->ncu * per_cu / 8 * 8));
    }
  }
  else {
    grid = c->coop ? c->ngrid_coop : c->ngrid_generic;
    if (a->flags & FA_FORCE_GENERIC) P.canon = 0;      // testing: per-entry column search
    if (P.overwrite) {
      if (want_K) HIP_TRY(hipMemsetAsync(a->K_values, 0, sizeof(double) * c->nnz, st));
      HIP_TRY(hipMemsetAsync(a->f_int, 0, sizeof(double) * c->ndof, st));
    }
  }
#ifdef FA_TRACE
  static void* dbgbuf = nullptr; static int ncall = 0;
  const size_t dbgbytes = (size_t)std::max(std::max(c->ntile2, c->ntile3), 1) * 4 * 8 * sizeof(long long);
  if (!dbgbuf) { HIP_TRY(hipMalloc(&dbgbuf, dbgbytes)); HIP_TRY(hipMemset(dbgbuf, 0, dbgbytes)); }
  P.dbg = dbgbuf;
#endif
  P.first_comp = nullptr; P.comp_mask = nullptr;
  if (c->d_first_comp) {                                // buffer of the context (fa_create): nothing is allocated here
    HIP_TRY(hipMemsetAsync(c->d_first_comp, 0xff, sizeof(unsigned long long), st));
    const int rc0 = (c->rank == 2) ? fa::launch_first_comp_2d(P, c->npe, c->nip, c->d_first_comp, c->d_comp_mask, (void*)st)
                                   : fa::launch_first_comp_3d(P, c->npe, c->nip, c->d_first_comp, c->d_comp_mask, (void*)st);
    if (rc0 != 0) return fail(FA_ERR_HIP, "first_comp_kernel launch failed: %s", hipGetErrorString((hipError_t)rc0));
    P.first_comp = c->d_first_comp; P.comp_mask = c->d_comp_mask;   // the mask: FA_MICRO_AMOR_MASK=1 only
  }
  P.stale_hits = c->d_stale;
  if (c->d_stale) HIP_TRY(hipMemsetAsync(c->d_stale, 0, sizeof(unsigned long long), st));
  int rc = c->launch(P, c->fam, want_K, keep, path, grid, (void*)st);
#ifdef FA_TRACE
  if (++ncall == 6 && getenv("FA_TRACE_FILE")) {
    std::vector<long long> h(dbgbytes / sizeof(long long));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(h.data(), dbgbuf, dbgbytes, hipMemcpyDeviceToHost));
    FILE* fp = fopen(getenv("FA_TRACE_FILE"), "wb");
    if (fp) { fwrite(h.data(), 1, dbgbytes, fp); fclose(fp); }
  }
#endif
  if (rc == -1000) return fail(FA_ERR_UNSUPPORTED, "kernel family not built for this shape");
  if (rc != 0) return fail(FA_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  if (want_err && part != 0) {                          // (a split pass sums its partials in the interior call: order it after the interface call)
    const int nparts = (path == fa::PATH_ROWS || path == fa::PATH_ROWS_ORDERED) ? c->nwg : (path == fa::PATH_ROWS_PAIR ? c->ntile2 : (path == fa::PATH_TILES ? (tile_fset ? tf_ntile : c->ntile3) : ((path == fa::PATH_EMAT || path == fa::PATH_FVEC) ? emat->ntile : grid)));
    hipLaunchKernelGGL(reduce_err_kernel, dim3(1), dim3(256), 0, st, err_part, nparts, a->err_ext);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(c->done_ev, st));
  return FA_OK;
}

extern "C" fa_status fa_assemble(fa_ctx* c, const fa_assemble_args* a) { return assemble_impl(c, a, false); }

extern "C" fa_status fa_assemble_noalloc(fa_ctx* c, const fa_assemble_args* a) { return assemble_impl(c, a, true); }

extern "C" fa_status fa_assemble_host(fa_ctx* c, const double* state, const double* state0, const double* state00,
                                      double dt, double dt0, double* K, double* f, double* err_ext, int32_t flags) {
  if (!c || !state || !f) return fail(FA_ERR_ARG, "null argument");
  FA_ON_DEVICE(c->prob.device);
  const size_t nb = sizeof(double) * (size_t)c->ndof;
  auto stage = [&](double** d, const double* h) -> fa_status {
    if (!h) return FA_OK;
    if (!*d) HIP_TRY(hipMalloc((void**)d, nb));
    HIP_TRY(hipMemcpy(*d, h, nb, hipMemcpyHostToDevice));
    return FA_OK;
  };
  fa_status st;
  if ((st = stage(&c->s_state, state)) || (st = stage(&c->s_state0, state0)) || (st = stage(&c->s_state00, state00))) return st;
  if (K && !c->s_K) HIP_TRY(hipMalloc((void**)&c->s_K, sizeof(double) * std::max<int64_t>(1, c->nnz)));
  if (!c->s_f) HIP_TRY(hipMalloc((void**)&c->s_f, nb));
  if (!c->s_err) HIP_TRY(hipMalloc((void**)&c->s_err, sizeof(double)));
  fa_assemble_args a{};
  a.state = c->s_state; a.state0 = state0 ? c->s_state0 : nullptr; a.state00 = state00 ? c->s_state00 : nullptr;
  a.dt = dt; a.dt0 = dt0; a.K_values = K ? c->s_K : nullptr; a.f_int = c->s_f;
  a.err_ext = err_ext ? c->s_err : nullptr;
  a.flags = (flags & ~FA_ACCUMULATE) | FA_OVERWRITE; a.stream = nullptr;
  if ((st = fa_assemble(c, &a))) return st;
  HIP_TRY(hipDeviceSynchronize());
  const bool over = flags & FA_OVERWRITE;
  auto fetch = [&](double* h, const double* d, size_t n) -> fa_status {
    if (over) { HIP_TRY(hipMemcpy(h, d, n * sizeof(double), hipMemcpyDeviceToHost)); return FA_OK; }
    c->h_tmp.resize(n);
    HIP_TRY(hipMemcpy(c->h_tmp.data(), d, n * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) h[i] += c->h_tmp[i];
    return FA_OK;
  };
  if (K && (st = fetch(K, c->s_K, (size_t)c->nnz))) return st;
  if ((st = fetch(f, c->s_f, (size_t)c->ndof))) return st;
  if (err_ext) {
    if (c->prob.model == FA_PHASE_FRACTURE_EXTIT || c->prob.model == FA_MICRO_PHASE_FRACTURE_EXTIT) HIP_TRY(hipMemcpy(err_ext, c->s_err, sizeof(double), hipMemcpyDeviceToHost));
    else *err_ext = 0.0;
  }
  return FA_OK;
}

extern "C" fa_status fa_commit(fa_ctx* c) {
  if (!c) return fail(FA_ERR_ARG, "null argument");
  if (c->prob.model == FA_PHASE_FRACTURE || c->prob.model >= FA_MICRO_PHASE_FRACTURE) std::swap(c->d_hist[0], c->d_hist[1]);
  return FA_OK;
}

extern "C" fa_status fa_get_history(const fa_ctx* c, int32_t which, double* out) {
  if (!c || !out || which < 0 || which > 1) return fail(FA_ERR_ARG, "bad argument");
  if (!c->d_hist[which]) return fail(FA_ERR_ARG, "this model keeps no history array");
  FA_ON_DEVICE(c->prob.device);
  HIP_TRY(hipEventSynchronize(c->done_ev));           // an assemble on a non-blocking stream may still be writing
  HIP_TRY(hipMemcpy(out, c->d_hist[which], sizeof(double) * c->nel * c->nip, hipMemcpyDeviceToHost));
  return FA_OK;
}

extern "C" fa_status fa_set_history(fa_ctx* c, int32_t which, const double* in) {
  if (!c || !in || which < 0 || which > 1) return fail(FA_ERR_ARG, "bad argument");
  if (!c->d_hist[which]) return fail(FA_ERR_ARG, "this model keeps no history array");
  FA_ON_DEVICE(c->prob.device);
  HIP_TRY(hipEventSynchronize(c->done_ev));
  HIP_TRY(hipMemcpy(c->d_hist[which], in, sizeof(double) * c->nel * c->nip, hipMemcpyHostToDevice));
  return FA_OK;
}

// getStrain_: one lane per node; per incident element the sum over its Gauss points first, then the add (the order in
// which the reference's element loop reaches the node)
__global__ void strain_table_kernel(const int32_t* __restrict__ n2e_ptr, const int32_t* __restrict__ n2e_ent,
                                    const double* __restrict__ tabN, const double* __restrict__ strain, int srows, int str,
                                    int npe, int nip, int64_t nnode, double* __restrict__ table, double* __restrict__ weights) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nnode) return;
  double acc[7] = { 0, 0, 0, 0, 0, 0, 0 }, w = 0.0;    // str <= 7: GradientDamage adds the e_eq column
  for (int32_t t = n2e_ptr[n]; t < n2e_ptr[n + 1]; t++) {
    const int32_t ent = n2e_ent[t];
    const int64_t e = ent / npe; const int la = ent - (int32_t)e * npe;
    double el[7] = { 0, 0, 0, 0, 0, 0, 0 }, ew = 0.0;
    for (int ip = 0; ip < nip; ip++) {
      const double N = tabN[ip * npe + la];
      const double* sp = strain + (e * nip + ip) * srows;
      for (int c = 0; c < 7; c++) if (c < str) el[c] += N * sp[c];
      ew += N;
    }
    for (int c = 0; c < 7; c++) acc[c] += el[c];
    w += ew;
  }
  for (int c = 0; c < 7; c++) if (c < str) table[n * str + c] += acc[c];
  weights[n] += w;
}

extern "C" fa_status fa_strain_table(fa_ctx* c, double* table, double* weights, void* stream) {
  if (!c || !table || !weights) return fail(FA_ERR_ARG, "null argument");
  if (!c->d_strain) return fail(FA_ERR_ARG, "no strain cache: assemble with FA_STORE_STRAIN first");
  // GradientDamageModel::getStrain_ (GradientDamageModel.cpp:1071-1177): strCount + 1 columns, the last one the nonlocal
  // equivalent strain "e_eq" (:1135, :1161); its column switch (:1097-1131) has no case 4, so in 2-D the reference throws
  int ncol = c->str;
  if (c->prob.model == FA_GRADIENT_DAMAGE) {
    if (c->str != 1 && c->str != 3 && c->str != 6)
      return fail(FA_ERR_UNSUPPORTED, "unexpected number of strain components: %d (GradientDamageModel::getStrain_ throws this in 2-D, GradientDamageModel.cpp:1124-1130)", c->str);
    ncol = c->str + 1;
  }
  FA_ON_DEVICE(c->prob.device);
  if (c->nnode == 0) return FA_OK;
  hipLaunchKernelGGL(strain_table_kernel, dim3((unsigned)((c->nnode + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     c->d_n2e_ptr, c->d_n2e_ent, c->d_tab, c->d_strain, c->srows, ncol, c->npe, c->nip, c->nnode, table, weights);
  HIP_TRY(hipGetLastError());
  return FA_OK;
}

// GradientDamageModel's getStress_ (0) / getDamage_ (1) / getOutputData_ point values (2): fa_tile_impl.h, gd_table_kernel
static fa_status gd_table(fa_ctx* c, fa::Params& P, int mode, double* out, double* weights, void* stream) {
  P.hist_pre = c->d_hist[0]; P.hist_new = c->d_hist[1];
  const int ps = (c->prob.state == FA_PLANE_STRESS) ? 1 : 0;
  const int rc = (c->rank == 2) ? fa::launch_gd_table_2d(P, c->npe, c->nip, ps, mode, out, weights, stream)
                                : fa::launch_gd_table_3d(P, c->npe, c->nip, ps, mode, out, weights, stream);
  if (rc != 0) return fail(FA_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  HIP_TRY(hipEventRecord(c->done_ev, (hipStream_t)stream));   // modes 0 and 2 rewrite newHist_: the history getters wait for it
  return FA_OK;
}

// getDamage_ (GradientDamageModel.cpp:1184-1241)
extern "C" fa_status fa_damage_table(fa_ctx* c, double* table, double* weights, void* stream) {
  if (!c || !table || !weights) return fail(FA_ERR_ARG, "null argument");
  if (c->prob.model != FA_GRADIENT_DAMAGE) return fail(FA_ERR_UNSUPPORTED, "fa_damage_table: getDamage_ belongs to GradientDamage");
  FA_ON_DEVICE(c->prob.device);
  if (c->nnode == 0) return FA_OK;
  fa::Params P = c->P;
  return gd_table(c, P, 1, table, weights, stream);
}

extern "C" fa_status fa_stress_table(fa_ctx* c, double* table, double* weights, void* stream) {
  if (!c || !table || !weights) return fail(FA_ERR_ARG, "null argument");
  if (!c->d_strain) return fail(FA_ERR_ARG, "no strain cache: assemble with FA_STORE_STRAIN first");
  FA_ON_DEVICE(c->prob.device);
  if (c->nnode == 0) return FA_OK;
  fa::Params P = c->P;
  P.strain = c->d_strain;
  const int ps = (c->prob.state == FA_PLANE_STRESS) ? 1 : 0;
  if (c->prob.model == FA_GRADIENT_DAMAGE) return gd_table(c, P, 0, table, weights, stream);
  const int rc = (c->rank == 2) ? fa::launch_stress_table_2d(P, c->npe, c->nip, ps, table, weights, stream)
                                : fa::launch_stress_table_3d(P, c->npe, c->nip, ps, table, weights, stream);
  if (rc != 0) return fail(FA_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  return FA_OK;
}

// table[n][j] += (strain | stress)[n][cols[j]]: point-value columns 0..str-1 are the strain, str..2 str-1 the stress
__global__ void output_split_filter_kernel(int64_t nnode, int str, int ncols, const int32_t* __restrict__ cols,
                                           const double* __restrict__ tstrain, const double* __restrict__ tstress,
                                           const double* __restrict__ wtmp, double* __restrict__ table,
                                           double* __restrict__ weights) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nnode) return;
  for (int j = 0; j < ncols; j++) {
    const int cj = cols[j];
    table[n * ncols + j] += (cj < str) ? tstrain[n * str + cj] : tstress[n * str + (cj - str)];
  }
  weights[n] += wtmp[n];
}

// table[n][j] += all[n][cols[j]] for point values stored interleaved
__global__ void output_filter_kernel(int64_t nnode, int ntmp, int ncols, const int32_t* __restrict__ cols,
                                     const double* __restrict__ all, const double* __restrict__ wtmp,
                                     double* __restrict__ table, double* __restrict__ weights) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nnode) return;
  for (int j = 0; j < ncols; j++) table[n * ncols + j] += all[n * ntmp + cols[j]];
  weights[n] += wtmp[n];
}

// getOutputData_ (PhaseFractureModel.cpp:1408-1508): strain.tensor and stress.tensor columns, filtered;
// GradientDamageModel.cpp:1254-1365: [strain.tensor | stress.tensor | damage]
extern "C" fa_status fa_output_data(fa_ctx* c, int32_t ncols, const int32_t* cols, double* table, double* weights, void* stream) {
  if (!c || !table || !weights || ncols < 0 || (ncols > 0 && !cols)) return fail(FA_ERR_ARG, "bad argument");
  if (!c->d_strain) return fail(FA_ERR_ARG, "no strain cache: assemble with FA_STORE_STRAIN first");
  const bool gd = c->prob.model == FA_GRADIENT_DAMAGE;
  const int str = c->str, ntmp = 2 * str + (gd ? 1 : 0);   // GradientDamage announces "damage" after the two tensors (:1270-1273)
  if (ncols > ntmp) return fail(FA_ERR_ARG, "fa_output_data: %d columns requested, the point values have %d", ncols, ntmp);
  for (int j = 0; j < ncols; j++)
    if (cols[j] < 0 || cols[j] >= ntmp) return fail(FA_ERR_ARG, "fa_output_data: column %d out of range [0, %d)", cols[j], ntmp);
  FA_ON_DEVICE(c->prob.device);
  if (c->nnode == 0) return FA_OK;
  hipStream_t st = (hipStream_t)stream;
  const size_t nn = (size_t)c->nnode;
  // layout of the scratch: [nnode][2 str] interleaved table, [nnode] weights, [nnode] second weights, 2 str column ids
  if (!c->d_out_tmp) HIP_TRY(hipMalloc((void**)&c->d_out_tmp, sizeof(double) * (nn * (ntmp + 2) + 16)));
  double* tstrain = c->d_out_tmp;                      // [nnode][str]
  double* tstress = tstrain + nn * str;                // [nnode][str]
  double* w1 = tstress + nn * str; double* w2 = w1 + nn;
  int32_t* dcols = (int32_t*)(w2 + nn);
  HIP_TRY(hipMemsetAsync(c->d_out_tmp, 0, sizeof(double) * nn * (ntmp + 2), st));
  if (ncols > 0) HIP_TRY(hipMemcpyAsync(dcols, cols, sizeof(int32_t) * ncols, hipMemcpyHostToDevice, st));
  if (gd) {                                            // one kernel writes all 2 str + 1 point values, interleaved
    fa::Params P = c->P;
    P.strain = c->d_strain;
    double* tall = c->d_out_tmp; double* wg = tall + nn * ntmp;   // dcols = wg + nn: the column ids sit behind both
    fa_status grc = gd_table(c, P, 2, tall, wg, stream);
    if (grc != FA_OK) return grc;
    hipLaunchKernelGGL(output_filter_kernel, dim3((unsigned)((c->nnode + 255) / 256)), dim3(256), 0, st, c->nnode, ntmp, ncols,
                       dcols, tall, wg, table, weights);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    return FA_OK;
  }
  fa_status rc = fa_strain_table(c, tstrain, w1, stream);
  if (rc != FA_OK) return rc;
  if ((rc = fa_stress_table(c, tstress, w2, stream)) != FA_OK) return rc;
  hipLaunchKernelGGL(output_split_filter_kernel, dim3((unsigned)((c->nnode + 255) / 256)), dim3(256), 0, st, c->nnode, str, ncols,
                     dcols, tstrain, tstress, w1, table, weights);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));                   // cols is a host array the caller may reuse
  return FA_OK;
}

// getArcFunc_ (PhaseFractureModel.cpp:844-1063)
extern "C" fa_status fa_arc_func(fa_ctx* c, const double* state, const double* state0, double* jac10, double* fe_dissipation,
                                 void* stream) {
  if (!c || !state || !state0 || !jac10 || !fe_dissipation) return fail(FA_ERR_ARG, "null argument");
  if (c->prob.model != FA_PHASE_FRACTURE && c->prob.model != FA_MICRO_PHASE_FRACTURE)
    return fail(FA_ERR_UNSUPPORTED, "fa_arc_func: getArcFunc_ is provided for PhaseFracture and MicroPhaseFracture only");
  FA_ON_DEVICE(c->prob.device);
  hipStream_t st = (hipStream_t)stream;
  const int nblk = (int)((c->nnode + 127) / 128);
  if (!c->d_arc_part) HIP_TRY(hipMalloc((void**)&c->d_arc_part, sizeof(double) * std::max(1, nblk)));
  HIP_TRY(hipMemsetAsync(jac10, 0, sizeof(double) * c->ndof, st));     // jac10 = 0.0 (:870): dofs of other models too
  fa::Params P = c->P;
  P.state = state;
  const int rc = (c->rank == 2) ? fa::launch_arc_func_2d(P, c->npe, c->nip, state0, jac10, c->d_arc_part, stream)
                                : fa::launch_arc_func_3d(P, c->npe, c->nip, state0, jac10, c->d_arc_part, stream);
  if (rc != 0) return fail(FA_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
  hipLaunchKernelGGL(reduce_err_kernel, dim3(1), dim3(256), 0, st, c->d_arc_part, nblk, fe_dissipation);
  HIP_TRY(hipGetLastError());
  return FA_OK;
}

// one workgroup, fixed order: strided partial sums per thread, then a tree over the 1024 threads
__global__ __launch_bounds__(1024) void gather_sum_kernel(const double* __restrict__ vec, const int64_t* __restrict__ idx,
                                                          int64_t n, double* __restrict__ out) {
  __shared__ double s[1024];
  double acc = 0.0;
  for (int64_t k = threadIdx.x; k < n; k += 1024) { const int64_t i = idx[k]; if (i >= 0) acc += vec[i]; }
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = s[0];
}

extern "C" fa_status fa_gather_sum(fa_ctx* c, const double* vec, const int64_t* idx, int64_t n, double* out, void* stream) {
  if (!c || !vec || !out || n < 0 || (n > 0 && !idx)) return fail(FA_ERR_ARG, "bad argument");
  FA_ON_DEVICE(c->prob.device);
  hipLaunchKernelGGL(gather_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, vec, idx, n, out);
  HIP_TRY(hipGetLastError());
  return FA_OK;
}

extern "C" double* fa_history_device(fa_ctx* c, int32_t which) {
  return (c && which >= 0 && which <= 1) ? c->d_hist[which] : nullptr;
}

extern "C" fa_status fa_get_strain_cache(const fa_ctx* c, double* out) {
  if (!c || !out) return fail(FA_ERR_ARG, "bad argument");
  if (!c->d_strain) return fail(FA_ERR_ARG, "no call with FA_STORE_STRAIN has been made");
  FA_ON_DEVICE(c->prob.device);
  HIP_TRY(hipEventSynchronize(c->done_ev));
  HIP_TRY(hipMemcpy(out, c->d_strain, sizeof(double) * c->nel * c->nip * c->srows, hipMemcpyDeviceToHost));
  return FA_OK;
}

// MicroPhaseFracture*: Gauss points of the last fa_assemble call that took the `d > 1` branch of
// MicroPhaseFractureModel.cpp:772-776 (the reference keeps a stale dphidStrain there, the kernels use 0)
extern "C" fa_status fa_micro_stale_hits(const fa_ctx* c, int64_t* out) {
  if (!c || !out) return fail(FA_ERR_ARG, "bad argument");
  *out = 0;
  if (!c->d_stale) return FA_OK;                       // not a micromorphic model: the branch does not exist
  FA_ON_DEVICE(c->prob.device);
  HIP_TRY(hipEventSynchronize(c->done_ev));
  unsigned long long v = 0;
  HIP_TRY(hipMemcpy(&v, c->d_stale, sizeof v, hipMemcpyDeviceToHost));
  *out = (int64_t)v;
  return FA_OK;
}

// ---- interface-first assembly for the multi-GPU exchange ----
// The kernels are organised in node tiles, each of which writes the rows (K rows, f_int entries) of the nodes it owns, so
// "the interface rows first" is a launch over the tiles that own an interface node.  The lists are made here, once.
extern "C" fa_status fa_set_interface(fa_ctx* c, int32_t nranges, const int64_t* node_ranges) {
  if (!c || nranges < 0 || (nranges > 0 && !node_ranges)) return fail(FA_ERR_ARG, "bad argument");
  FA_ON_DEVICE(c->prob.device);
  std::vector<int32_t> pre((size_t)c->nnode + 1, 0);   // prefix sums of the interface marks
  {
    std::vector<uint8_t> mark((size_t)c->nnode, 0);
    for (int r = 0; r < nranges; r++) {
      const int64_t b = node_ranges[2 * r], e = node_ranges[2 * r + 1];
      if (b < 0 || e > c->nnode || b > e) return fail(FA_ERR_ARG, "fa_set_interface: node range %d = [%lld, %lld) is outside [0, %lld]", r, (long long)b, (long long)e, (long long)c->nnode);
      for (int64_t n = b; n < e; n++) mark[(size_t)n] = 1;
    }
    for (int64_t n = 0; n < c->nnode; n++) pre[(size_t)n + 1] = pre[(size_t)n] + mark[(size_t)n];
  }
  auto marked = [&](int64_t n) { return pre[(size_t)n + 1] != pre[(size_t)n]; };
  auto publish = [&](PartLists& pl, const std::vector<int32_t> (&lists)[2]) -> fa_status {
    for (int k = 0; k < 2; k++) {
      if (pl.d[k]) { (void)hipFree(pl.d[k]); pl.d[k] = nullptr; }
      pl.n[k] = (int)lists[k].size();
      fa_status st = upload(&pl.d[k], lists[k].data(), lists[k].size());
      if (st != FA_OK) return st;
    }
    return FA_OK;
  };
  // tiles that are ranges of consecutive nodes (row kernels): descriptor words 0, 1 = first node, one past the last
  auto ranges = [&](const int32_t* d_info, int ntile, PartLists& pl) -> fa_status {
    if (!d_info || ntile <= 0) return FA_OK;
    std::vector<int32_t> wi((size_t)ntile * fa::kWgInfo);
    HIP_TRY(hipMemcpy(wi.data(), d_info, wi.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<int32_t> lists[2];
    for (int t = 0; t < ntile; t++) {
      const int32_t n0 = wi[(size_t)t * fa::kWgInfo], n1 = wi[(size_t)t * fa::kWgInfo + 1];
      lists[(pre[(size_t)n1] != pre[(size_t)n0]) ? 0 : 1].push_back(t);
    }
    return publish(pl, lists);
  };
  // tiles with a node list, owned nodes first (tile kernel, force-vector kernel)
  auto nodelists = [&](const int32_t* d_tinfo, const int32_t* d_tnodes, int ntile, PartLists& pl) -> fa_status {
    if (!d_tinfo || ntile <= 0) return FA_OK;
    std::vector<int32_t> ti((size_t)ntile * fa::kTileInfo);
    HIP_TRY(hipMemcpy(ti.data(), d_tinfo, ti.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    const size_t nn = (size_t)ti[(size_t)(ntile - 1) * fa::kTileInfo] + (size_t)ti[(size_t)(ntile - 1) * fa::kTileInfo + 1];
    std::vector<int32_t> tn(nn);
    HIP_TRY(hipMemcpy(tn.data(), d_tnodes, nn * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<int32_t> lists[2];
    for (int t = 0; t < ntile; t++) {
      const int32_t* q = ti.data() + (size_t)t * fa::kTileInfo;
      bool any = false;
      for (int i = 0; i < q[2] && !any; i++) any = marked(tn[(size_t)q[0] + i]);
      lists[any ? 0 : 1].push_back(t);
    }
    return publish(pl, lists);
  };
  fa_status st;
  if (c->rows_path && (st = ranges(c->d_wginfo, c->nwg, c->part_rows))) return st;
  if (c->pair_path && (st = ranges(c->d_wginfo2, c->ntile2, c->part_pair))) return st;
  if (c->tile_path && (st = nodelists(c->d_tinfo, c->d_tnodes, c->ntile3, c->part_tiles))) return st;
  if (c->fvec.ok && (st = nodelists(c->fvec.tinfo, c->fvec.tnodes, c->fvec.ntile, c->part_fvec))) return st;
  if (c->tileF.ok && (st = nodelists(c->tileF.tinfo, c->tileF.tnodes, c->tileF.ntile, c->part_tileF))) return st;
  c->parts_set = true;
  return FA_OK;
}
