// fa_constraints.hip -- constraint elimination and residual evaluation on the device-resident K / f_int
// (SURVEY.md 8f rows N2 / N3): what sits between the assembly and a device linear solver.
//
// Reference call sites: src/solvers/linear/cuDSSSolver.cpp:108-121 (StdConstrainer wraps the matrix), :293-294
// (getRhs / initLhs), :416 (getLhs); src/solvers/implicit/BFGSModule.cpp:438-489 (updateResidual).  The Jive classes
// behind them (jive::util::Constraints, evalMasterDofs / evalSlaveDofs, StdConstrainer) are not part of the reference
// checkout; their documented effect is restated in oracle/falcon_oracle_cons.c, which the GPU tests compare with.
//
// A constraint is u[slave] = rval + sum coef * u[master] (masters are never slaves): u = T u + c.  fa_cons_eliminate
// turns K du = r into (T^t K T + I_S) du = T^t (r - K c) in place.  All lists the kernels walk are built once, on the
// host, in fa_cons_create: only the entries in slave rows / slave columns and their fold destinations are touched, so
// a call costs O(boundary), not O(nnz).  Every output entry is summed by one lane in a fixed order: no atomics,
// bitwise repeatable.
//
// Set-up never copies the pattern to the host: a kernel scans the nnz column indices for slave columns at HBM speed, the
// rows the elimination touches (hit rows, slave rows, master rows) are gathered and copied back, and everything else is
// done on those: O(nnz) device reads, O(boundary) host work and memory.  Entries the constrained matrix needs and the
// context's pattern lacks (StdConstrainer builds a new structure for them, cuDSSSolver.cpp:108-121) go into an OUTPUT
// pattern the constraint set owns: fa_cons_eliminate_out copies K into it and eliminates there.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <unordered_map>
#include <vector>

#include "../../include/falcon_assemble.h"
#include "fa_internal.h"

struct fa_cons {
  fa_ctx* ctx = nullptr;
  int device = 0;
  int64_t ndof = 0, nslave = 0, nmaster_rows = 0, naff = 0, ncol_ent = 0, nd1 = 0, nd2 = 0;
  bool has_masters = false;
  // slaves
  int64_t* d_slave = nullptr; double* d_rval = nullptr; int64_t* d_sm_ptr = nullptr; int64_t* d_sm_dof = nullptr;
  double* d_sm_coef = nullptr; int64_t* d_sdiag = nullptr;
  // rows holding entries in slave columns: rhs[row] -= sum K[pos] * rval[slave]
  int64_t* d_a_row = nullptr; int64_t* d_a_ptr = nullptr; int64_t* d_a_pos = nullptr; int32_t* d_a_sl = nullptr;
  // column fold K <- K T: destination entries (row, master column) and their sources (row, slave column)
  int64_t* d_d1_pos = nullptr; int64_t* d_d1_ptr = nullptr; int64_t* d_s1_pos = nullptr; double* d_s1_coef = nullptr;
  // row fold K <- T^t K: destination entries (master row, column) and their sources (slave row, column)
  int64_t* d_d2_pos = nullptr; int64_t* d_d2_ptr = nullptr; int64_t* d_s2_pos = nullptr; double* d_s2_coef = nullptr;
  // vector fold: master dof -> (slave dof, coef)
  int64_t* d_m_dof = nullptr; int64_t* d_m_ptr = nullptr; int64_t* d_ms_dof = nullptr; double* d_ms_coef = nullptr;
  // CSR the kernels index K with: the context's (device pointers it owns) or, with fill-in, the output pattern below
  const int64_t* d_row_ptr = nullptr;
  double* d_part = nullptr;                            // partial sums of fa_residual
  // output pattern (nfill > 0 only): the context's pattern plus the entries the elimination writes outside it
  int64_t nnz_in = 0, nfill = 0, next = 0, next_len = 0;
  int64_t* d_rpo = nullptr; int32_t* d_cio = nullptr; int32_t* d_rpo32 = nullptr;
  // extended rows: input range [eb, ee), fill-in entries before the row (cum, next + 1 entries), offsets into the
  // per-entry lists ext_src (input position or -1) / ext_col (next + 1 entries)
  int64_t* d_eb = nullptr; int64_t* d_ee = nullptr; int64_t* d_cum = nullptr; int64_t* d_ext_off = nullptr;
  int64_t* d_ext_src = nullptr; int32_t* d_ext_col = nullptr; int64_t* d_ext_row = nullptr;
};

namespace {

constexpr int kRedBlocks = 1024;

#define CONS_TRY(expr)                                                                             \
  do { hipError_t _e = (expr);                                                                     \
       if (_e != hipSuccess) return fa::set_error(FA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); } while (0)

struct Guard {
  int prev = -1; bool ok = false;
  explicit Guard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (prev == dev) || hipSetDevice(dev) == hipSuccess;
    if (prev == dev) prev = -1;
  }
  ~Guard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

template <class T>
fa_status to_device(T** d, const std::vector<T>& h) {
  *d = nullptr;
  CONS_TRY(hipMalloc((void**)d, sizeof(T) * std::max<size_t>(1, h.size())));
  if (!h.empty()) CONS_TRY(hipMemcpy(*d, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  return FA_OK;
}

inline unsigned blocks_for(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + 255) / 256); }

// rhs[row] -= sum over the row's slave-column entries of K * rval
__global__ void cons_rhs_kernel(int64_t n, const int64_t* __restrict__ a_row, const int64_t* __restrict__ a_ptr,
                                const int64_t* __restrict__ a_pos, const int32_t* __restrict__ a_sl,
                                const double* __restrict__ rval, const double* __restrict__ K, double* __restrict__ rhs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int64_t e = a_ptr[i]; e < a_ptr[i + 1]; e++) acc += K[a_pos[e]] * rval[a_sl[e]];
  rhs[a_row[i]] -= acc;
}

// K[dst] += sum coef * K[src]: sources and destinations are disjoint sets of entries
__global__ void cons_fold_kernel(int64_t n, const int64_t* __restrict__ d_pos, const int64_t* __restrict__ d_ptr,
                                 const int64_t* __restrict__ s_pos, const double* __restrict__ s_coef, double* __restrict__ K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int64_t e = d_ptr[i]; e < d_ptr[i + 1]; e++) acc += s_coef[e] * K[s_pos[e]];
  K[d_pos[i]] += acc;
}

__global__ void cons_zero_entries_kernel(int64_t n, const int64_t* __restrict__ pos, double* __restrict__ K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) K[pos[i]] = 0.0;
}

// vec[master] += sum coef * vec[slave]
__global__ void cons_vec_fold_kernel(int64_t n, const int64_t* __restrict__ m_dof, const int64_t* __restrict__ m_ptr,
                                     const int64_t* __restrict__ ms_dof, const double* __restrict__ ms_coef,
                                     double* __restrict__ vec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int64_t e = m_ptr[i]; e < m_ptr[i + 1]; e++) acc += ms_coef[e] * vec[ms_dof[e]];
  vec[m_dof[i]] += acc;
}

// slave rows: zero, unit diagonal; rhs[slave] = 0.  K or vec may be null.
__global__ void cons_slave_rows_kernel(int64_t n, const int64_t* __restrict__ slave, const int64_t* __restrict__ sdiag,
                                       const int64_t* __restrict__ row_ptr, double* __restrict__ K, double* __restrict__ vec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t s = slave[i];
  if (K) {
    for (int64_t p = row_ptr[s]; p < row_ptr[s + 1]; p++) K[p] = 0.0;
    K[sdiag[i]] = 1.0;
  }
  if (vec) vec[s] = 0.0;
}

__global__ void cons_eval_slaves_kernel(int64_t n, const int64_t* __restrict__ slave, const double* __restrict__ rval,
                                        const int64_t* __restrict__ sm_ptr, const int64_t* __restrict__ sm_dof,
                                        const double* __restrict__ sm_coef, double* __restrict__ vec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double t = rval[i];
  for (int64_t e = sm_ptr[i]; e < sm_ptr[i + 1]; e++) t += sm_coef[e] * vec[sm_dof[e]];
  vec[slave[i]] = t;
}

__global__ void residual_axpy_kernel(int64_t n, const double* __restrict__ fext, const double* __restrict__ fint,
                                     double* __restrict__ r) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) r[i] = fext[i] - 1.0 * fint[i];
}

// bound projection (BFGSModule.cpp:452-479) and the two sums; fixed grid, fixed order
__global__ __launch_bounds__(256) void residual_norm_kernel(int64_t n, const double* __restrict__ r, const double* __restrict__ du,
                                                            const double* __restrict__ u, const double* __restrict__ lower,
                                                            const double* __restrict__ upper, double* __restrict__ r2,
                                                            double* __restrict__ part) {
  __shared__ double s0[256], s1[256];
  const double epsilon = 1.4901161193847656e-08;       // sqrt(Float::EPSILON)
  double a0 = 0.0, a1 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double f = r[i];
    if (lower) {
      double lb = lower[i], ub = upper[i];
      const double x = u[i];
      f = -f;
      lb += epsilon * fabs(lb);
      ub -= epsilon * fabs(ub);
      if (x <= lb) f = fmin(0.0, f);
      if (x >= ub) f = fmax(0.0, f);
      r2[i] = f;
    }
    a0 += f * f; a1 += f * du[i];
  }
  s0[threadIdx.x] = a0; s1[threadIdx.x] = a1;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { s0[threadIdx.x] += s0[threadIdx.x + w]; s1[threadIdx.x] += s1[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[blockIdx.x] = s0[0]; part[gridDim.x + blockIdx.x] = s1[0]; }
}

__global__ __launch_bounds__(256) void residual_final_kernel(int nb, const double* __restrict__ part, double* __restrict__ out) {
  __shared__ double s0[256], s1[256];
  double a0 = 0.0, a1 = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) { a0 += part[i]; a1 += part[nb + i]; }
  s0[threadIdx.x] = a0; s1[threadIdx.x] = a1;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { s0[threadIdx.x] += s0[threadIdx.x + w]; s1[threadIdx.x] += s1[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = sqrt(s0[0]); out[1] = s1[0]; out[2] = s0[0]; }
}

// ---- set-up kernels ----
__global__ void cons_mark_kernel(int64_t n, const int64_t* __restrict__ slave, int32_t* __restrict__ slmap) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) slmap[slave[i]] = (int32_t)i;
}

// entries in slave columns: counted (hits == nullptr) or listed, in no particular order (the host sorts them)
__global__ void cons_scan_kernel(int64_t nnz, const int32_t* __restrict__ ci, const int32_t* __restrict__ slmap,
                                 unsigned long long* __restrict__ counter, int64_t cap, int64_t* __restrict__ hits) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride)
    if (slmap[ci[p]] >= 0) {
      const unsigned long long k = atomicAdd(counter, 1ull);
      if (hits && (int64_t)k < cap) hits[k] = p;
    }
}

// row and column of the listed entries (row: the last r with row_ptr[r] <= p)
__global__ void cons_locate_kernel(int64_t n, const int64_t* __restrict__ hits, int64_t ndof, const int64_t* __restrict__ rp,
                                   const int32_t* __restrict__ ci, int64_t* __restrict__ row, int32_t* __restrict__ col) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = hits[i];
  int64_t lo = 0, hi = ndof;                            // invariant: rp[lo] <= p < rp[hi]
  while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (rp[mid] <= p) lo = mid; else hi = mid; }
  row[i] = lo; col[i] = ci[p];
}

__global__ void cons_row_bounds_kernel(int64_t n, const int64_t* __restrict__ rows, const int64_t* __restrict__ rp,
                                       int64_t* __restrict__ rb, int64_t* __restrict__ re) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { rb[i] = rp[rows[i]]; re[i] = rp[rows[i] + 1]; }
}

// the column lists of the gathered rows, packed: one wave per row
__global__ void cons_gather_rows_kernel(int64_t n, const int64_t* __restrict__ rb, const int64_t* __restrict__ re,
                                        const int64_t* __restrict__ off, const int32_t* __restrict__ ci, int32_t* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (i >= n) return;
  const int64_t len = re[i] - rb[i];
  for (int64_t j = threadIdx.x & 63; j < len; j += 64) out[off[i] + j] = ci[rb[i] + j];
}

// ---- output pattern ----
// number of extended rows whose input range starts at or before p / whose row index is below r
__device__ __forceinline__ int64_t upper_bound_dev(const int64_t* __restrict__ a, int64_t n, int64_t v) {
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
  return lo;
}

__global__ void cons_rpo_kernel(int64_t ndof, const int64_t* __restrict__ rp, int64_t next, const int64_t* __restrict__ ext_row,
                                const int64_t* __restrict__ cum, int64_t* __restrict__ rpo) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > ndof) return;
  rpo[r] = rp[r] + cum[upper_bound_dev(ext_row, next, r - 1)];      // fill-in entries in extended rows < r
}

// entries outside the extended rows move up by the fill-in before them: out[p + cum] = in[p].  A workgroup takes
// chunks of kCopyChunk consecutive entries and looks the extended rows up once per chunk; only chunks that hold one are
// searched entry by entry (and only among the rows of the chunk).
constexpr int kCopyChunk = 4096;
template <class T>
__global__ __launch_bounds__(256) void cons_shift_copy_kernel(int64_t nnz, int64_t next, const int64_t* __restrict__ eb,
                                                              const int64_t* __restrict__ ee, const int64_t* __restrict__ cum,
                                                              const T* __restrict__ in, T* __restrict__ out) {
  const int64_t nchunk = (nnz + kCopyChunk - 1) / kCopyChunk;
  for (int64_t ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
    const int64_t p0 = ch * kCopyChunk, p1 = (p0 + kCopyChunk < nnz) ? p0 + kCopyChunk : nnz;
    const int64_t i0 = upper_bound_dev(eb, next, p0), i1 = upper_bound_dev(eb, next, p1 - 1);   // uniform over the workgroup
    if (i0 == i1 && (i0 == 0 || p0 >= ee[i0 - 1])) {   // no extended row starts in or reaches into the chunk
      const int64_t sh = cum[i0];
      for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) out[p + sh] = in[p];
      continue;
    }
    for (int64_t p = p0 + threadIdx.x; p < p1; p += 256) {
      int64_t i = i0;
      while (i < i1 && eb[i] <= p) i++;                // the chunk holds few extended rows
      if (i > 0 && p < ee[i - 1]) continue;            // inside an extended row: cons_ext_copy_kernel
      out[p + cum[i]] = in[p];
    }
  }
}

// entries of the extended rows: from their input position, or the fill value (cols: the new column; values: 0)
template <class T>
__global__ void cons_ext_copy_kernel(int64_t ntot, int64_t next, const int64_t* __restrict__ eb, const int64_t* __restrict__ cum,
                                     const int64_t* __restrict__ ext_off, const int64_t* __restrict__ ext_src,
                                     const T* __restrict__ fill, const T* __restrict__ in, T* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ntot) return;
  const int64_t i = upper_bound_dev(ext_off, next + 1, j) - 1;
  const int64_t q = eb[i] + cum[i] + (j - ext_off[i]);
  const int64_t src = ext_src[j];
  out[q] = (src >= 0) ? in[src] : (fill ? fill[j] : T(0));
}

__global__ void cons_narrow_kernel(int64_t n, const int64_t* __restrict__ in, int32_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int32_t)in[i];
}

// one gathered row on the host
struct RowView {
  int64_t row = 0, rb = 0; int32_t len = 0; const int32_t* cols = nullptr; bool sorted = true;
  std::vector<int32_t> added;                          // fill-in columns, ascending
  int32_t find_old(int64_t c) const {                  // index among the input columns or -1
    if (sorted) { const int32_t* e = cols + len; const int32_t* q = std::lower_bound(cols, e, (int32_t)c); return (q != e && *q == c) ? (int32_t)(q - cols) : -1; }
    for (int32_t j = 0; j < len; j++) if (cols[j] == c) return j;
    return -1;
  }
  bool has_added(int64_t c) const { return std::binary_search(added.begin(), added.end(), (int32_t)c); }
};

struct Triple { int64_t dst, src; double coef; };
struct Fold { int64_t drow, dcol, srow, scol; double coef; };   // K(drow, dcol) += coef * K(srow, scol)

// group (dst, src, coef) triples by destination, sources in ascending order of their position
void group_triples(std::vector<Triple>& t, std::vector<int64_t>& d_pos, std::vector<int64_t>& d_ptr,
                   std::vector<int64_t>& s_pos, std::vector<double>& s_coef) {
  std::sort(t.begin(), t.end(), [](const Triple& a, const Triple& b) { return a.dst != b.dst ? a.dst < b.dst : a.src < b.src; });
  d_pos.clear(); d_ptr.assign(1, 0); s_pos.clear(); s_coef.clear();
  for (size_t i = 0; i < t.size(); i++) {
    if (i == 0 || t[i].dst != t[i - 1].dst) { if (i) d_ptr.push_back((int64_t)i); d_pos.push_back(t[i].dst); }
    s_pos.push_back(t[i].src); s_coef.push_back(t[i].coef);
  }
  if (!t.empty()) d_ptr.push_back((int64_t)t.size());
}

}  // namespace

extern "C" void fa_cons_destroy(fa_cons* c) {
  if (!c) return;
  Guard g(c->device);
  void* ptrs[] = { c->d_slave, c->d_rval, c->d_sm_ptr, c->d_sm_dof, c->d_sm_coef, c->d_sdiag, c->d_a_row, c->d_a_ptr,
                   c->d_a_pos, c->d_a_sl, c->d_d1_pos, c->d_d1_ptr, c->d_s1_pos, c->d_s1_coef, c->d_d2_pos, c->d_d2_ptr,
                   c->d_s2_pos, c->d_s2_coef, c->d_m_dof, c->d_m_ptr, c->d_ms_dof, c->d_ms_coef, c->d_part, c->d_rpo, c->d_cio,
                   c->d_rpo32, c->d_eb, c->d_ee, c->d_cum, c->d_ext_off, c->d_ext_src, c->d_ext_col, c->d_ext_row };
  for (void* p : ptrs) if (p) (void)hipFree(p);
  delete c;
}

extern "C" fa_status fa_cons_create(fa_ctx* ctx, int64_t nslave, const int64_t* slave_dofs, const double* rvalues,
                                    const int64_t* master_ptr, const int64_t* master_dofs, const double* master_coefs,
                                    fa_cons** out) {
  if (!ctx || !out || nslave < 0 || (nslave > 0 && !slave_dofs)) return fa::set_error(FA_ERR_ARG, "fa_cons_create: null or negative argument");
  *out = nullptr;
  const int64_t ndof = fa_ndof(ctx), nnz = fa_nnz(ctx);
  const int64_t nm = master_ptr ? master_ptr[nslave] : 0;
  if (master_ptr && nm > 0 && (!master_dofs || !master_coefs)) return fa::set_error(FA_ERR_ARG, "fa_cons_create: master_ptr without master_dofs / master_coefs");
  if (nslave > (int64_t)0x7fffffff) return fa::set_error(FA_ERR_UNSUPPORTED, "fa_cons_create: more than 2^31 - 1 slave dofs");
  std::unordered_map<int64_t, int64_t> sl;              // slave dof -> its index
  sl.reserve((size_t)nslave * 2 + 16);
  for (int64_t k = 0; k < nslave; k++) {
    const int64_t s = slave_dofs[k];
    if (s < 0 || s >= ndof) return fa::set_error(FA_ERR_ARG, "fa_cons_create: slave dof %lld out of range [0, %lld)", (long long)s, (long long)ndof);
    if (!sl.emplace(s, k).second) return fa::set_error(FA_ERR_ARG, "fa_cons_create: dof %lld is constrained twice", (long long)s);
  }
  auto slave_of = [&](int64_t dof) -> int64_t { auto it = sl.find(dof); return it == sl.end() ? -1 : it->second; };
  if (master_ptr) {
    if (master_ptr[0] != 0) return fa::set_error(FA_ERR_ARG, "fa_cons_create: master_ptr[0] must be 0");
    for (int64_t k = 0; k < nslave; k++)
      if (master_ptr[k + 1] < master_ptr[k]) return fa::set_error(FA_ERR_ARG, "fa_cons_create: master_ptr decreases at %lld", (long long)k);
    for (int64_t q = 0; q < nm; q++) {
      const int64_t m = master_dofs[q];
      if (m < 0 || m >= ndof) return fa::set_error(FA_ERR_ARG, "fa_cons_create: master dof %lld out of range", (long long)m);
      // Jive resolves such chains when the constraints are compressed; here the caller passes the compressed form
      if (slave_of(m) >= 0) return fa::set_error(FA_ERR_UNSUPPORTED, "fa_cons_create: master dof %lld is itself a slave (pass the constraints in compressed form)", (long long)m);
    }
  }
  fa_cons* c = new fa_cons;
  c->ctx = ctx; c->device = fa_device(ctx); c->ndof = ndof; c->nslave = nslave; c->has_masters = nm > 0; c->nnz_in = nnz;
  Guard guard(c->device);
  if (!guard.ok) { delete c; return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device); }
  const int32_t* d_ci = nullptr;
  fa_pattern_device(ctx, &c->d_row_ptr, &d_ci);
  const int64_t* d_rp = c->d_row_ptr;
  fa_status st = FA_OK;
  std::vector<void*> tmp;                               // set-up scratch on the device, freed on every way out
  auto cleanup = [&] { for (void* q : tmp) if (q) (void)hipFree(q); tmp.clear(); };
  auto fail_out = [&](fa_status rc) { cleanup(); fa_cons_destroy(c); return rc; };
#define SETUP_TRY(expr)                                                                                                  \
  do { hipError_t _e = (expr);                                                                                          \
       if (_e != hipSuccess) return fail_out(fa::set_error(FA_ERR_HIP, "fa_cons_create: %s failed: %s", #expr, hipGetErrorString(_e))); } while (0)
  auto dalloc = [&](size_t bytes) -> void* { void* q = nullptr; if (hipMalloc(&q, std::max<size_t>(8, bytes)) != hipSuccess) return nullptr; tmp.push_back(q); return q; };
#define SETUP_ALLOC(var, type, count)                                                                                    \
  type* var = (type*)dalloc(sizeof(type) * (size_t)(count));                                                            \
  if (!var) return fail_out(fa::set_error(FA_ERR_NOMEM, "fa_cons_create: hipMalloc of %lld bytes failed", (long long)(sizeof(type) * (size_t)(count))))

  // ---- 1. entries in slave columns: a device scan of the column indices (the pattern need not be structurally
  //         symmetric); their rows and columns located on the device as well
  std::vector<int64_t> hit_pos, hit_row; std::vector<int32_t> hit_col;
  if (nslave > 0 && nnz > 0) {
    SETUP_ALLOC(d_slmap, int32_t, ndof);
    SETUP_ALLOC(d_sl, int64_t, nslave);
    SETUP_ALLOC(d_cnt, unsigned long long, 1);
    SETUP_TRY(hipMemset(d_slmap, 0xFF, sizeof(int32_t) * (size_t)ndof));
    SETUP_TRY(hipMemcpy(d_sl, slave_dofs, sizeof(int64_t) * (size_t)nslave, hipMemcpyHostToDevice));
    SETUP_TRY(hipMemset(d_cnt, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(cons_mark_kernel, dim3(blocks_for(nslave)), dim3(256), 0, 0, nslave, d_sl, d_slmap);
    const unsigned scan_blocks = (unsigned)std::min<int64_t>(8192, std::max<int64_t>(1, (nnz + 255) / 256));
    hipLaunchKernelGGL(cons_scan_kernel, dim3(scan_blocks), dim3(256), 0, 0, nnz, d_ci, d_slmap, d_cnt, (int64_t)0, (int64_t*)nullptr);
    SETUP_TRY(hipGetLastError());
    unsigned long long nh = 0;
    SETUP_TRY(hipMemcpy(&nh, d_cnt, sizeof nh, hipMemcpyDeviceToHost));
    if (nh > 0) {
      SETUP_ALLOC(d_hits, int64_t, nh);
      SETUP_ALLOC(d_hrow, int64_t, nh);
      SETUP_ALLOC(d_hcol, int32_t, nh);
      SETUP_TRY(hipMemset(d_cnt, 0, sizeof(unsigned long long)));
      hipLaunchKernelGGL(cons_scan_kernel, dim3(scan_blocks), dim3(256), 0, 0, nnz, d_ci, d_slmap, d_cnt, (int64_t)nh, d_hits);
      hipLaunchKernelGGL(cons_locate_kernel, dim3(blocks_for((int64_t)nh)), dim3(256), 0, 0, (int64_t)nh, d_hits, ndof, d_rp, d_ci, d_hrow, d_hcol);
      SETUP_TRY(hipGetLastError());
      std::vector<int64_t> hp(nh), hr(nh); std::vector<int32_t> hc(nh);
      SETUP_TRY(hipMemcpy(hp.data(), d_hits, sizeof(int64_t) * nh, hipMemcpyDeviceToHost));
      SETUP_TRY(hipMemcpy(hr.data(), d_hrow, sizeof(int64_t) * nh, hipMemcpyDeviceToHost));
      SETUP_TRY(hipMemcpy(hc.data(), d_hcol, sizeof(int32_t) * nh, hipMemcpyDeviceToHost));
      std::vector<int64_t> order(nh);
      for (size_t i = 0; i < nh; i++) order[i] = (int64_t)i;
      std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return hp[a] < hp[b]; });   // the atomics' order is arbitrary
      hit_pos.resize(nh); hit_row.resize(nh); hit_col.resize(nh);
      for (size_t i = 0; i < nh; i++) { hit_pos[i] = hp[order[i]]; hit_row[i] = hr[order[i]]; hit_col[i] = hc[order[i]]; }
    }
    cleanup();
  }

  // ---- 2. the rows the elimination reads or writes: hit rows, slave rows, master rows -- gathered, copied back
  std::vector<int64_t> R(hit_row);
  R.insert(R.end(), slave_dofs, slave_dofs + nslave);
  if (master_ptr) R.insert(R.end(), master_dofs, master_dofs + nm);
  std::sort(R.begin(), R.end());
  R.erase(std::unique(R.begin(), R.end()), R.end());
  const int64_t nR = (int64_t)R.size();
  std::vector<int64_t> rb(nR), re(nR), roff(nR + 1, 0);
  std::vector<int32_t> rcols;
  if (nR > 0) {
    SETUP_ALLOC(d_R, int64_t, nR);
    SETUP_ALLOC(d_rb, int64_t, nR);
    SETUP_ALLOC(d_re, int64_t, nR);
    SETUP_ALLOC(d_roff, int64_t, nR + 1);
    SETUP_TRY(hipMemcpy(d_R, R.data(), sizeof(int64_t) * nR, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cons_row_bounds_kernel, dim3(blocks_for(nR)), dim3(256), 0, 0, nR, d_R, d_rp, d_rb, d_re);
    SETUP_TRY(hipGetLastError());
    SETUP_TRY(hipMemcpy(rb.data(), d_rb, sizeof(int64_t) * nR, hipMemcpyDeviceToHost));
    SETUP_TRY(hipMemcpy(re.data(), d_re, sizeof(int64_t) * nR, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < nR; i++) roff[i + 1] = roff[i] + (re[i] - rb[i]);
    rcols.resize((size_t)std::max<int64_t>(1, roff[nR]));
    if (roff[nR] > 0) {
      SETUP_ALLOC(d_rcols, int32_t, roff[nR]);
      SETUP_TRY(hipMemcpy(d_roff, roff.data(), sizeof(int64_t) * (nR + 1), hipMemcpyHostToDevice));
      hipLaunchKernelGGL(cons_gather_rows_kernel, dim3(blocks_for(nR * 64)), dim3(256), 0, 0, nR, d_rb, d_re, d_roff, d_ci, d_rcols);
      SETUP_TRY(hipGetLastError());
      SETUP_TRY(hipMemcpy(rcols.data(), d_rcols, sizeof(int32_t) * (size_t)roff[nR], hipMemcpyDeviceToHost));
    }
    cleanup();
  }
  std::vector<RowView> rows((size_t)nR);
  for (int64_t i = 0; i < nR; i++) {
    RowView& v = rows[i];
    v.row = R[i]; v.rb = rb[i]; v.len = (int32_t)(re[i] - rb[i]); v.cols = rcols.data() + roff[i];
    v.sorted = std::is_sorted(v.cols, v.cols + v.len);
  }
  auto row_of = [&](int64_t r) -> RowView& { return rows[(size_t)(std::lower_bound(R.begin(), R.end(), r) - R.begin())]; };

  // ---- 3. the folds, entry by entry as (row, column) pairs; destinations the pattern lacks become fill-in
  std::vector<int64_t> a_row, a_ptr(1, 0), a_pos; std::vector<int32_t> a_sl;
  std::vector<Fold> f1, f2;
  std::vector<std::pair<int64_t, int32_t>> add;         // (index into rows, column)
  auto commit_add = [&] {                                // sort / unique the new entries into their rows
    std::sort(add.begin(), add.end());
    add.erase(std::unique(add.begin(), add.end()), add.end());
    for (auto& e : add) {
      RowView& v = rows[(size_t)e.first];
      if (!v.has_added(e.second)) v.added.insert(std::upper_bound(v.added.begin(), v.added.end(), e.second), e.second);
    }
    add.clear();
  };
  // column fold K <- K T: (row, slave column) -> (row, master columns)
  for (size_t h = 0; h < hit_pos.size(); h++) {
    const int64_t row = hit_row[h], k = slave_of(hit_col[h]);
    if (a_row.empty() || a_row.back() != row) { if (!a_row.empty()) a_ptr.push_back((int64_t)a_pos.size()); a_row.push_back(row); }
    a_pos.push_back(hit_pos[h]); a_sl.push_back((int32_t)k);
    if (master_ptr) {
      const int64_t ri = std::lower_bound(R.begin(), R.end(), row) - R.begin();
      for (int64_t q = master_ptr[k]; q < master_ptr[k + 1]; q++) {
        if (rows[(size_t)ri].find_old(master_dofs[q]) < 0) add.emplace_back(ri, (int32_t)master_dofs[q]);
        f1.push_back({ row, master_dofs[q], row, hit_col[h], master_coefs[q] });
      }
    }
  }
  if (!a_row.empty()) a_ptr.push_back((int64_t)a_pos.size());
  commit_add();
  // row fold K <- T^t K: (slave row, non-slave column, the fill-in of the column fold included) -> (master rows, column);
  // the slaves' diagonal entries must exist as well
  for (int64_t k = 0; k < nslave; k++) {
    const int64_t s = slave_dofs[k];
    const int64_t si = std::lower_bound(R.begin(), R.end(), s) - R.begin();
    const RowView& sv = rows[(size_t)si];
    if (sv.find_old(s) < 0) add.emplace_back(si, (int32_t)s);
    if (!master_ptr) continue;
    auto fold_col = [&](int64_t col) {
      if (slave_of(col) >= 0) return;                    // a slave column: zero after the column fold
      for (int64_t q = master_ptr[k]; q < master_ptr[k + 1]; q++) {
        const int64_t mi = std::lower_bound(R.begin(), R.end(), master_dofs[q]) - R.begin();
        if (rows[(size_t)mi].find_old(col) < 0 && !rows[(size_t)mi].has_added(col)) add.emplace_back(mi, (int32_t)col);
        f2.push_back({ master_dofs[q], col, s, col, master_coefs[q] });
      }
    };
    for (int32_t j = 0; j < sv.len; j++) fold_col(sv.cols[j]);
    for (int32_t col : sv.added) fold_col(col);
  }
  commit_add();

  // ---- 4. layout of the output pattern: extended rows keep their order when the input row is sorted by column
  //         (fill-in merged in), otherwise the new entries go to the end of the row
  std::vector<int64_t> ext_row, eb, ee, cum(1, 0), ext_off(1, 0), ext_src; std::vector<int32_t> ext_col;
  std::vector<int64_t> row_ext((size_t)nR, -1), row_shift((size_t)nR, 0);
  {
    int64_t total = 0;
    for (int64_t i = 0; i < nR; i++) {
      RowView& v = rows[(size_t)i];
      row_shift[i] = total;
      if (v.added.empty()) continue;
      row_ext[i] = (int64_t)ext_row.size();
      ext_row.push_back(v.row); eb.push_back(v.rb); ee.push_back(v.rb + v.len);
      if (v.sorted) {
        int32_t a = 0; size_t b = 0;
        while (a < v.len || b < v.added.size()) {
          if (b >= v.added.size() || (a < v.len && v.cols[a] < v.added[b])) { ext_col.push_back(v.cols[a]); ext_src.push_back(v.rb + a); a++; }
          else { ext_col.push_back(v.added[b]); ext_src.push_back(-1); b++; }
        }
      } else {
        for (int32_t a = 0; a < v.len; a++) { ext_col.push_back(v.cols[a]); ext_src.push_back(v.rb + a); }
        for (int32_t col : v.added) { ext_col.push_back(col); ext_src.push_back(-1); }
      }
      total += (int64_t)v.added.size();
      cum.push_back(total); ext_off.push_back((int64_t)ext_col.size());
    }
    c->nfill = total; c->next = (int64_t)ext_row.size(); c->next_len = (int64_t)ext_col.size();
  }
  // position of (row, column) in the pattern the kernels will index: the output pattern when there is fill-in
  auto out_pos = [&](int64_t r, int64_t col) -> int64_t {
    const int64_t i = std::lower_bound(R.begin(), R.end(), r) - R.begin();
    const RowView& v = rows[(size_t)i];
    if (row_ext[i] < 0) { const int32_t j = v.find_old(col); return j < 0 ? -1 : v.rb + j + row_shift[i]; }
    const int64_t e = row_ext[i], b = ext_off[e], n = ext_off[e + 1] - b;
    for (int64_t j = 0; j < n; j++) if (ext_col[(size_t)(b + j)] == col) return v.rb + row_shift[i] + j;
    return -1;
  };
  std::vector<Triple> t1, t2;
  t1.reserve(f1.size()); t2.reserve(f2.size());
  for (const Fold& f : f1) t1.push_back({ out_pos(f.drow, f.dcol), out_pos(f.srow, f.scol), f.coef });
  for (const Fold& f : f2) t2.push_back({ out_pos(f.drow, f.dcol), out_pos(f.srow, f.scol), f.coef });
  for (const std::vector<Triple>* tv : { &t1, &t2 })
    for (const Triple& t : *tv)
      if (t.dst < 0 || t.src < 0) return fail_out(fa::set_error(FA_ERR_ARG, "fa_cons_create: an entry of the constrained matrix was not laid out"));
  for (size_t h = 0; h < a_pos.size(); h++) a_pos[h] = out_pos(hit_row[h], hit_col[h]);
  std::vector<int64_t> sdiag(nslave);
  for (int64_t k = 0; k < nslave; k++) sdiag[k] = out_pos(slave_dofs[k], slave_dofs[k]);

  std::vector<int64_t> d1_pos, d1_ptr, s1_pos, d2_pos, d2_ptr, s2_pos; std::vector<double> s1_coef, s2_coef;
  group_triples(t1, d1_pos, d1_ptr, s1_pos, s1_coef);
  group_triples(t2, d2_pos, d2_ptr, s2_pos, s2_coef);
  // vector fold: master dof -> its slaves in ascending slave order
  std::vector<Triple> tv;
  if (master_ptr)
    for (int64_t k = 0; k < nslave; k++)
      for (int64_t q = master_ptr[k]; q < master_ptr[k + 1]; q++) tv.push_back({ master_dofs[q], slave_dofs[k], master_coefs[q] });
  std::vector<int64_t> m_dof, m_ptr, ms_dof; std::vector<double> ms_coef;
  group_triples(tv, m_dof, m_ptr, ms_dof, ms_coef);

  std::vector<int64_t> h_slave(slave_dofs, slave_dofs + nslave), sm_ptr(nslave + 1, 0), sm_dof;
  std::vector<double> h_rval(nslave, 0.0), sm_coef;
  if (rvalues) std::copy(rvalues, rvalues + nslave, h_rval.begin());
  if (master_ptr) {
    std::copy(master_ptr, master_ptr + nslave + 1, sm_ptr.begin());
    sm_dof.assign(master_dofs, master_dofs + nm); sm_coef.assign(master_coefs, master_coefs + nm);
  }
  c->naff = (int64_t)a_row.size(); c->ncol_ent = (int64_t)a_pos.size();
  c->nd1 = (int64_t)d1_pos.size(); c->nd2 = (int64_t)d2_pos.size(); c->nmaster_rows = (int64_t)m_dof.size();
  if (d1_ptr.size() < 1) d1_ptr.assign(1, 0);
  if ((st = to_device(&c->d_slave, h_slave)) || (st = to_device(&c->d_rval, h_rval)) || (st = to_device(&c->d_sm_ptr, sm_ptr)) ||
      (st = to_device(&c->d_sm_dof, sm_dof)) || (st = to_device(&c->d_sm_coef, sm_coef)) || (st = to_device(&c->d_sdiag, sdiag)) ||
      (st = to_device(&c->d_a_row, a_row)) || (st = to_device(&c->d_a_ptr, a_ptr)) || (st = to_device(&c->d_a_pos, a_pos)) ||
      (st = to_device(&c->d_a_sl, a_sl)) || (st = to_device(&c->d_d1_pos, d1_pos)) || (st = to_device(&c->d_d1_ptr, d1_ptr)) ||
      (st = to_device(&c->d_s1_pos, s1_pos)) || (st = to_device(&c->d_s1_coef, s1_coef)) || (st = to_device(&c->d_d2_pos, d2_pos)) ||
      (st = to_device(&c->d_d2_ptr, d2_ptr)) || (st = to_device(&c->d_s2_pos, s2_pos)) || (st = to_device(&c->d_s2_coef, s2_coef)) ||
      (st = to_device(&c->d_m_dof, m_dof)) || (st = to_device(&c->d_m_ptr, m_ptr)) || (st = to_device(&c->d_ms_dof, ms_dof)) ||
      (st = to_device(&c->d_ms_coef, ms_coef))) return fail_out(st);
  if (hipMalloc((void**)&c->d_part, sizeof(double) * 2 * kRedBlocks) != hipSuccess) return fail_out(fa::set_error(FA_ERR_NOMEM, "fa_cons_create: hipMalloc failed"));

  // ---- 5. the output pattern on the device
  if (c->nfill > 0) {
    if ((st = to_device(&c->d_ext_row, ext_row)) || (st = to_device(&c->d_eb, eb)) || (st = to_device(&c->d_ee, ee)) ||
        (st = to_device(&c->d_cum, cum)) || (st = to_device(&c->d_ext_off, ext_off)) || (st = to_device(&c->d_ext_src, ext_src)) ||
        (st = to_device(&c->d_ext_col, ext_col))) return fail_out(st);
    const int64_t nnz_out = nnz + c->nfill;
    if (hipMalloc((void**)&c->d_rpo, sizeof(int64_t) * (size_t)(ndof + 1)) != hipSuccess ||
        hipMalloc((void**)&c->d_cio, sizeof(int32_t) * (size_t)nnz_out) != hipSuccess)
      return fail_out(fa::set_error(FA_ERR_NOMEM, "fa_cons_create: hipMalloc of the output pattern (%lld entries) failed", (long long)nnz_out));
    hipLaunchKernelGGL(cons_rpo_kernel, dim3(blocks_for(ndof + 1)), dim3(256), 0, 0, ndof, d_rp, c->next, c->d_ext_row, c->d_cum, c->d_rpo);
    const unsigned cb = (unsigned)std::min<int64_t>(1 << 20, std::max<int64_t>(1, (nnz + kCopyChunk - 1) / kCopyChunk));
    hipLaunchKernelGGL(cons_shift_copy_kernel<int32_t>, dim3(cb), dim3(256), 0, 0, nnz, c->next, c->d_eb, c->d_ee, c->d_cum, d_ci, c->d_cio);
    hipLaunchKernelGGL(cons_ext_copy_kernel<int32_t>, dim3(blocks_for(c->next_len)), dim3(256), 0, 0, c->next_len, c->next, c->d_eb, c->d_cum,
                       c->d_ext_off, c->d_ext_src, c->d_ext_col, d_ci, c->d_cio);
    SETUP_TRY(hipGetLastError());
    SETUP_TRY(hipDeviceSynchronize());
    c->d_row_ptr = c->d_rpo;                             // cons_slave_rows_kernel walks the slave rows of the output matrix
  }
#undef SETUP_TRY
#undef SETUP_ALLOC
  *out = c;
  return FA_OK;
}

// ---- the output pattern (fill-in) ----
extern "C" int64_t fa_cons_fill_in(const fa_cons* c) { return c ? c->nfill : -1; }
extern "C" int64_t fa_cons_nnz(const fa_cons* c) { return c ? c->nnz_in + c->nfill : -1; }

extern "C" fa_status fa_cons_pattern_device(const fa_cons* c, const int64_t** row_ptr, const int32_t** col_idx) {
  if (!c || !row_ptr || !col_idx) return fa::set_error(FA_ERR_ARG, "fa_cons_pattern_device: null argument");
  if (c->nfill == 0) return fa_pattern_device(c->ctx, row_ptr, col_idx);
  *row_ptr = c->d_rpo; *col_idx = c->d_cio;
  return FA_OK;
}

extern "C" fa_status fa_cons_pattern_device32(fa_cons* c, const int32_t** row_ptr, const int32_t** col_idx) {
  if (!c || !row_ptr || !col_idx) return fa::set_error(FA_ERR_ARG, "fa_cons_pattern_device32: null argument");
  if (c->nfill == 0) return fa_pattern_device32(c->ctx, row_ptr, col_idx);
  if (c->nnz_in + c->nfill > (int64_t)0x7fffffff) return fa::set_error(FA_ERR_UNSUPPORTED, "fa_cons_pattern_device32: %lld entries do not fit 32-bit offsets", (long long)(c->nnz_in + c->nfill));
  Guard g(c->device);
  if (!g.ok) return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device);
  if (!c->d_rpo32) {
    CONS_TRY(hipMalloc((void**)&c->d_rpo32, sizeof(int32_t) * (size_t)(c->ndof + 1)));
    hipLaunchKernelGGL(cons_narrow_kernel, dim3(blocks_for(c->ndof + 1)), dim3(256), 0, 0, c->ndof + 1, c->d_rpo, c->d_rpo32);
    CONS_TRY(hipGetLastError());
    CONS_TRY(hipDeviceSynchronize());
  }
  *row_ptr = c->d_rpo32; *col_idx = c->d_cio;
  return FA_OK;
}

extern "C" fa_status fa_cons_pattern_host(const fa_cons* c, int64_t* row_ptr, int32_t* col_idx) {
  if (!c || !row_ptr || !col_idx) return fa::set_error(FA_ERR_ARG, "fa_cons_pattern_host: null argument");
  if (c->nfill == 0) return fa_pattern_host(c->ctx, row_ptr, col_idx);
  Guard g(c->device);
  if (!g.ok) return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device);
  CONS_TRY(hipMemcpy(row_ptr, c->d_rpo, sizeof(int64_t) * (size_t)(c->ndof + 1), hipMemcpyDeviceToHost));
  CONS_TRY(hipMemcpy(col_idx, c->d_cio, sizeof(int32_t) * (size_t)(c->nnz_in + c->nfill), hipMemcpyDeviceToHost));
  return FA_OK;
}

extern "C" fa_status fa_cons_set_rvalues(fa_cons* c, const double* rvalues, void* stream) {
  if (!c || (c->nslave > 0 && !rvalues)) return fa::set_error(FA_ERR_ARG, "fa_cons_set_rvalues: null argument");
  Guard g(c->device);
  if (!g.ok) return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device);
  if (c->nslave > 0) {
    CONS_TRY(hipMemcpyAsync(c->d_rval, rvalues, sizeof(double) * c->nslave, hipMemcpyHostToDevice, (hipStream_t)stream));
    CONS_TRY(hipStreamSynchronize((hipStream_t)stream));   // the host buffer may be pageable and reused by the caller
  }
  return FA_OK;
}

static fa_status vec_fold(const fa_cons* c, double* vec, hipStream_t st) {
  if (c->nmaster_rows > 0) {
    hipLaunchKernelGGL(cons_vec_fold_kernel, dim3(blocks_for(c->nmaster_rows)), dim3(256), 0, st, c->nmaster_rows,
                       c->d_m_dof, c->d_m_ptr, c->d_ms_dof, c->d_ms_coef, vec);
    CONS_TRY(hipGetLastError());
  }
  return FA_OK;
}

extern "C" fa_status fa_cons_eval_masters(const fa_cons* c, double* vec, void* stream) {
  if (!c || !vec) return fa::set_error(FA_ERR_ARG, "fa_cons_eval_masters: null argument");
  Guard g(c->device);
  if (!g.ok) return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device);
  if (c->nslave == 0) return FA_OK;
  hipStream_t st = (hipStream_t)stream;
  fa_status rc = vec_fold(c, vec, st);
  if (rc != FA_OK) return rc;
  hipLaunchKernelGGL(cons_slave_rows_kernel, dim3(blocks_for(c->nslave)), dim3(256), 0, st, c->nslave, c->d_slave, c->d_sdiag,
                     c->d_row_ptr, (double*)nullptr, vec);
  CONS_TRY(hipGetLastError());
  return FA_OK;
}

extern "C" fa_status fa_cons_eval_slaves(const fa_cons* c, double* vec, void* stream) {
  if (!c || !vec) return fa::set_error(FA_ERR_ARG, "fa_cons_eval_slaves: null argument");
  Guard g(c->device);
  if (!g.ok) return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device);
  if (c->nslave == 0) return FA_OK;
  hipLaunchKernelGGL(cons_eval_slaves_kernel, dim3(blocks_for(c->nslave)), dim3(256), 0, (hipStream_t)stream, c->nslave,
                     c->d_slave, c->d_rval, c->d_sm_ptr, c->d_sm_dof, c->d_sm_coef, vec);
  CONS_TRY(hipGetLastError());
  return FA_OK;
}

static fa_status eliminate_on(const fa_cons* c, double* K, double* rhs, hipStream_t st);

extern "C" fa_status fa_cons_eliminate(const fa_cons* c, double* K, double* rhs, void* stream) {
  if (!c || (!K && !rhs)) return fa::set_error(FA_ERR_ARG, "fa_cons_eliminate: null argument");
  if (rhs && !K) return fa::set_error(FA_ERR_ARG, "fa_cons_eliminate: the right-hand side needs the unconstrained K (rhs -= K c)");
  if (c->nfill > 0)
    return fa::set_error(FA_ERR_ARG, "fa_cons_eliminate: the constrained matrix has %lld entries outside the context's pattern; "
                                     "use fa_cons_eliminate_out with an array of fa_cons_nnz entries", (long long)c->nfill);
  Guard g(c->device);
  if (!g.ok) return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device);
  if (c->nslave == 0) return FA_OK;
  return eliminate_on(c, K, rhs, (hipStream_t)stream);
}

extern "C" fa_status fa_cons_eliminate_out(const fa_cons* c, const double* K, double* K_out, double* rhs, void* stream) {
  if (!c || !K || !K_out) return fa::set_error(FA_ERR_ARG, "fa_cons_eliminate_out: null argument");
  if (K == K_out && c->nfill > 0) return fa::set_error(FA_ERR_ARG, "fa_cons_eliminate_out: with fill-in the output array must be a different one");
  Guard g(c->device);
  if (!g.ok) return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device);
  hipStream_t st = (hipStream_t)stream;
  if (c->nfill == 0) {
    if (K != K_out && c->nnz_in > 0) CONS_TRY(hipMemcpyAsync(K_out, K, sizeof(double) * (size_t)c->nnz_in, hipMemcpyDeviceToDevice, st));
  } else {
    const unsigned cb = (unsigned)std::min<int64_t>(1 << 20, std::max<int64_t>(1, (c->nnz_in + kCopyChunk - 1) / kCopyChunk));
    hipLaunchKernelGGL(cons_shift_copy_kernel<double>, dim3(cb), dim3(256), 0, st, c->nnz_in, c->next, c->d_eb, c->d_ee, c->d_cum, K, K_out);
    hipLaunchKernelGGL(cons_ext_copy_kernel<double>, dim3(blocks_for(c->next_len)), dim3(256), 0, st, c->next_len, c->next, c->d_eb, c->d_cum,
                       c->d_ext_off, c->d_ext_src, (const double*)nullptr, K, K_out);
    CONS_TRY(hipGetLastError());
  }
  if (c->nslave == 0) return FA_OK;
  return eliminate_on(c, K_out, rhs, st);
}

static fa_status eliminate_on(const fa_cons* c, double* K, double* rhs, hipStream_t st) {
  if (rhs && c->naff > 0) {                            // rhs -= K c, from the unconstrained K
    hipLaunchKernelGGL(cons_rhs_kernel, dim3(blocks_for(c->naff)), dim3(256), 0, st, c->naff, c->d_a_row, c->d_a_ptr,
                       c->d_a_pos, c->d_a_sl, c->d_rval, K, rhs);
    CONS_TRY(hipGetLastError());
  }
  if (c->nd1 > 0) {                                    // K <- K T
    hipLaunchKernelGGL(cons_fold_kernel, dim3(blocks_for(c->nd1)), dim3(256), 0, st, c->nd1, c->d_d1_pos, c->d_d1_ptr,
                       c->d_s1_pos, c->d_s1_coef, K);
    CONS_TRY(hipGetLastError());
  }
  if (c->ncol_ent > 0) {
    hipLaunchKernelGGL(cons_zero_entries_kernel, dim3(blocks_for(c->ncol_ent)), dim3(256), 0, st, c->ncol_ent, c->d_a_pos, K);
    CONS_TRY(hipGetLastError());
  }
  if (c->nd2 > 0) {                                    // K <- T^t K
    hipLaunchKernelGGL(cons_fold_kernel, dim3(blocks_for(c->nd2)), dim3(256), 0, st, c->nd2, c->d_d2_pos, c->d_d2_ptr,
                       c->d_s2_pos, c->d_s2_coef, K);
    CONS_TRY(hipGetLastError());
  }
  if (rhs) { fa_status rc = vec_fold(c, rhs, st); if (rc != FA_OK) return rc; }   // rhs <- T^t rhs
  hipLaunchKernelGGL(cons_slave_rows_kernel, dim3(blocks_for(c->nslave)), dim3(256), 0, st, c->nslave, c->d_slave, c->d_sdiag,
                     c->d_row_ptr, K, rhs);
  CONS_TRY(hipGetLastError());
  return FA_OK;
}

extern "C" fa_status fa_residual(const fa_cons* c, const double* fext, const double* fint, const double* du, const double* u,
                                 const double* lower, const double* upper, double* r, double* r2, double* out, void* stream) {
  if (!c || !fext || !fint || !du || !r || !out) return fa::set_error(FA_ERR_ARG, "fa_residual: null argument");
  const bool bounded = lower || upper;
  if (bounded && (!lower || !upper || !u || !r2)) return fa::set_error(FA_ERR_ARG, "fa_residual: bounds need lower, upper, u and r2");
  Guard g(c->device);
  if (!g.ok) return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device);
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = c->ndof;
  const int nb = (int)std::min<int64_t>(kRedBlocks, std::max<int64_t>(1, (n + 255) / 256));
  hipLaunchKernelGGL(residual_axpy_kernel, dim3(nb), dim3(256), 0, st, n, fext, fint, r);
  CONS_TRY(hipGetLastError());
  fa_status rc = fa_cons_eval_masters(c, r, stream);
  if (rc != FA_OK) return rc;
  hipLaunchKernelGGL(residual_norm_kernel, dim3(nb), dim3(256), 0, st, n, r, du, u, bounded ? lower : nullptr, upper, r2, c->d_part);
  CONS_TRY(hipGetLastError());
  hipLaunchKernelGGL(residual_final_kernel, dim3(1), dim3(256), 0, st, nb, c->d_part, out);
  CONS_TRY(hipGetLastError());
  return FA_OK;
}
