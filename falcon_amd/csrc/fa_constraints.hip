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
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <thread>
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
  // CSR (device pointers owned by the context)
  const int64_t* d_row_ptr = nullptr;
  double* d_part = nullptr;                            // partial sums of fa_residual
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

int64_t find_in_row(const std::vector<int64_t>& rp, const std::vector<int32_t>& ci, int64_t r, int64_t c) {
  for (int64_t p = rp[r]; p < rp[r + 1]; p++) if (ci[p] == c) return p;
  return -1;
}

struct Triple { int64_t dst, src; double coef; };

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
                   c->d_s2_pos, c->d_s2_coef, c->d_m_dof, c->d_m_ptr, c->d_ms_dof, c->d_ms_coef, c->d_part };
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
  std::vector<int64_t> sl(ndof, -1);
  for (int64_t k = 0; k < nslave; k++) {
    const int64_t s = slave_dofs[k];
    if (s < 0 || s >= ndof) return fa::set_error(FA_ERR_ARG, "fa_cons_create: slave dof %lld out of range [0, %lld)", (long long)s, (long long)ndof);
    if (sl[s] >= 0) return fa::set_error(FA_ERR_ARG, "fa_cons_create: dof %lld is constrained twice", (long long)s);
    sl[s] = k;
  }
  if (master_ptr) {
    if (master_ptr[0] != 0) return fa::set_error(FA_ERR_ARG, "fa_cons_create: master_ptr[0] must be 0");
    for (int64_t k = 0; k < nslave; k++)
      if (master_ptr[k + 1] < master_ptr[k]) return fa::set_error(FA_ERR_ARG, "fa_cons_create: master_ptr decreases at %lld", (long long)k);
    for (int64_t q = 0; q < nm; q++) {
      const int64_t m = master_dofs[q];
      if (m < 0 || m >= ndof) return fa::set_error(FA_ERR_ARG, "fa_cons_create: master dof %lld out of range", (long long)m);
      // Jive resolves such chains when the constraints are compressed; here the caller passes the compressed form
      if (sl[m] >= 0) return fa::set_error(FA_ERR_UNSUPPORTED, "fa_cons_create: master dof %lld is itself a slave (pass the constraints in compressed form)", (long long)m);
    }
  }
  fa_cons* c = new fa_cons;
  c->ctx = ctx; c->device = fa_device(ctx); c->ndof = ndof; c->nslave = nslave; c->has_masters = nm > 0;
  Guard guard(c->device);
  if (!guard.ok) { delete c; return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device); }
  const int32_t* d_ci = nullptr;
  fa_pattern_device(ctx, &c->d_row_ptr, &d_ci);

  std::vector<int64_t> rp(ndof + 1);
  std::vector<int32_t> ci(std::max<int64_t>(1, nnz));
  fa_status st = fa_pattern_host(ctx, rp.data(), ci.data());
  if (st != FA_OK) { delete c; return st; }

  // entries in slave columns, row by row (a full scan: the pattern need not be structurally symmetric)
  const int nthr = (int)std::min<int64_t>(std::max(1u, std::min(std::thread::hardware_concurrency(), 16u)), std::max<int64_t>(1, ndof / 65536));
  std::vector<std::vector<int64_t>> loc_pos(nthr);
  {
    std::vector<std::thread> th;
    const int64_t chunk = (ndof + nthr - 1) / nthr;
    for (int t = 0; t < nthr; t++)
      th.emplace_back([&, t] {
        const int64_t b = t * chunk, e = std::min(ndof, b + chunk);
        for (int64_t i = b; i < e; i++)
          for (int64_t p = rp[i]; p < rp[i + 1]; p++) if (sl[ci[p]] >= 0) loc_pos[t].push_back(p);
      });
    for (auto& x : th) x.join();
  }
  std::vector<int64_t> a_row, a_ptr(1, 0), a_pos; std::vector<int32_t> a_sl;
  std::vector<Triple> t1, t2;
  {
    int64_t row = 0;
    for (int t = 0; t < nthr; t++)
      for (int64_t p : loc_pos[t]) {
        while (rp[row + 1] <= p) row++;
        if (a_row.empty() || a_row.back() != row) { if (!a_row.empty()) a_ptr.push_back((int64_t)a_pos.size()); a_row.push_back(row); }
        const int64_t k = sl[ci[p]];
        a_pos.push_back(p); a_sl.push_back((int32_t)k);
        if (master_ptr)
          for (int64_t q = master_ptr[k]; q < master_ptr[k + 1]; q++) {
            const int64_t d = find_in_row(rp, ci, row, master_dofs[q]);
            if (d < 0) {
              delete c;
              return fa::set_error(FA_ERR_ARG, "fa_cons_create: the CSR pattern has no entry (%lld, %lld), which folding slave dof %lld into its master writes; "
                                               "pass fa_create a pattern that holds the constrained matrix's entries", (long long)row, (long long)master_dofs[q], (long long)ci[p]);
            }
            t1.push_back({ d, p, master_coefs[q] });
          }
      }
    if (!a_row.empty()) a_ptr.push_back((int64_t)a_pos.size());
  }
  // slave rows: the diagonal, and the fold of the non-slave columns into the master rows
  std::vector<int64_t> sdiag(nslave);
  for (int64_t k = 0; k < nslave; k++) {
    const int64_t s = slave_dofs[k];
    sdiag[k] = find_in_row(rp, ci, s, s);
    if (sdiag[k] < 0) { delete c; return fa::set_error(FA_ERR_ARG, "fa_cons_create: the CSR pattern has no diagonal entry for slave dof %lld", (long long)s); }
    if (master_ptr)
      for (int64_t p = rp[s]; p < rp[s + 1]; p++) {
        if (sl[ci[p]] >= 0) continue;
        for (int64_t q = master_ptr[k]; q < master_ptr[k + 1]; q++) {
          const int64_t d = find_in_row(rp, ci, master_dofs[q], ci[p]);
          if (d < 0) {
            delete c;
            return fa::set_error(FA_ERR_ARG, "fa_cons_create: the CSR pattern has no entry (%lld, %lld), which folding the row of slave dof %lld into its master writes; "
                                             "pass fa_create a pattern that holds the constrained matrix's entries", (long long)master_dofs[q], (long long)ci[p], (long long)s);
          }
          t2.push_back({ d, p, master_coefs[q] });
        }
      }
  }
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
      (st = to_device(&c->d_ms_coef, ms_coef))) { fa_cons_destroy(c); return st; }
  if (hipMalloc((void**)&c->d_part, sizeof(double) * 2 * kRedBlocks) != hipSuccess) { fa_cons_destroy(c); return fa::set_error(FA_ERR_NOMEM, "fa_cons_create: hipMalloc failed"); }
  *out = c;
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

extern "C" fa_status fa_cons_eliminate(const fa_cons* c, double* K, double* rhs, void* stream) {
  if (!c || (!K && !rhs)) return fa::set_error(FA_ERR_ARG, "fa_cons_eliminate: null argument");
  if (rhs && !K) return fa::set_error(FA_ERR_ARG, "fa_cons_eliminate: the right-hand side needs the unconstrained K (rhs -= K c)");
  Guard g(c->device);
  if (!g.ok) return fa::set_error(FA_ERR_HIP, "hipSetDevice(%d) failed", c->device);
  if (c->nslave == 0) return FA_OK;
  hipStream_t st = (hipStream_t)stream;
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
