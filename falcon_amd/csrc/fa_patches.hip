// fa_patches.hip -- host code only: recursive coordinate bisection of the mesh nodes into patches.
//
// A patch is the unit of work of assemble_patch: a compact set of nodes whose CSR rows fit the LDS
// image, plus every element touching one of them.  Compactness matters twice: the halo (elements
// shared with neighbouring patches, whose Gauss points are evaluated by both) shrinks with the
// surface/volume ratio, and the gathers of coordinates / state stay inside one L2.
#include "fa_patches.h"

#include "fa_internal.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>

namespace fa {
namespace {

struct Builder {
  int64_t nnode; int rank; const double* coords; const int64_t* row_ptr; int ndofpn; int cap, max_own;
  std::vector<int32_t> perm;
  std::vector<std::pair<int64_t, int64_t>> leaves;
  std::mutex mu;
  std::atomic<bool> bad{false};
  std::atomic<int> threads_left{0};

  int64_t blk(int32_t n) const { return row_ptr[(int64_t)(n + 1) * ndofpn] - row_ptr[(int64_t)n * ndofpn]; }

  void rec(int64_t lo, int64_t hi) {
    if (bad.load()) return;
    const int64_t cnt = hi - lo;
    if (cnt <= 0) return;
    int64_t tot = 0;
    double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
    for (int64_t i = lo; i < hi; i++) {
      const int32_t n = perm[i];
      tot += blk(n);
      for (int d = 0; d < rank; d++) { const double c = coords[(int64_t)n * rank + d]; mn[d] = std::min(mn[d], c); mx[d] = std::max(mx[d], c); }
    }
    int64_t m = std::max<int64_t>((tot + cap - 1) / cap, (cnt + max_own - 1) / max_own);
    if (m <= 1) { std::lock_guard<std::mutex> g(mu); leaves.emplace_back(lo, hi); return; }
    if (cnt == 1) { bad = true; return; }
    int dim = 0;
    for (int d = 1; d < rank; d++) if (mx[d] - mn[d] > mx[dim] - mn[dim]) dim = d;
    int64_t mid = lo + cnt * (m / 2) / m;
    mid = std::min(std::max(mid, lo + 1), hi - 1);
    const double* c = coords; const int r = rank;
    std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi, [c, r, dim](int32_t a, int32_t b) {
      const double ca = c[(int64_t)a * r + dim], cb = c[(int64_t)b * r + dim];
      return ca < cb || (ca == cb && a < b);
    });
    if (cnt > 200000 && threads_left.fetch_sub(1) > 0) {
      std::thread t([this, lo, mid] { rec(lo, mid); });
      rec(mid, hi);
      t.join();
    } else {
      rec(lo, mid);
      rec(mid, hi);
    }
  }
};

template <class F>
void par_for(int64_t n, F&& body) {
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, n / 256));
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

}  // namespace

bool build_patches(int64_t nnode, int64_t nel, int npe, int rank, const double* coords, const int32_t* conn,
                   const int32_t* n2e_ptr, const int32_t* n2e_ent, const uint8_t* pos, const int64_t* row_ptr, int ndofpn,
                   int cap_lds, int max_own, PatchSet& out) {
  Builder B{ nnode, rank, coords, row_ptr, ndofpn, cap_lds, max_own };
  B.perm.resize(nnode);
  for (int64_t i = 0; i < nnode; i++) B.perm[i] = (int32_t)i;
  B.threads_left = 15;
  B.rec(0, nnode);
  if (B.bad) return false;
  std::sort(B.leaves.begin(), B.leaves.end());
  const int np = (int)B.leaves.size();
  out.npatch = np;

  // owned nodes (ascending), LDS offsets, runs
  out.pn_ptr.assign(np + 1, 0);
  for (int p = 0; p < np; p++) out.pn_ptr[p + 1] = out.pn_ptr[p] + (int32_t)(B.leaves[p].second - B.leaves[p].first);
  out.pnode.resize(nnode);
  out.poff.resize(nnode + np);
  std::vector<int32_t> owner_patch(nnode), owner_idx(nnode);
  std::vector<int32_t> nruns(np), nelems(np), maxlds(np);
  par_for(np, [&](int64_t pb, int64_t pe) {
    for (int64_t p = pb; p < pe; p++) {
      int32_t* nodes = out.pnode.data() + out.pn_ptr[p];
      const int cnt = out.pn_ptr[p + 1] - out.pn_ptr[p];
      std::copy(B.perm.begin() + B.leaves[p].first, B.perm.begin() + B.leaves[p].second, nodes);
      std::sort(nodes, nodes + cnt);
      int32_t* off = out.poff.data() + out.pn_ptr[p] + p;
      int32_t o = 0; int runs = 0;
      for (int j = 0; j < cnt; j++) {
        off[j] = o; o += (int32_t)B.blk(nodes[j]);
        owner_patch[nodes[j]] = (int32_t)p; owner_idx[nodes[j]] = j;
        if (j == 0 || nodes[j] != nodes[j - 1] + 1) runs++;
      }
      off[cnt] = o; maxlds[p] = o; nruns[p] = runs;
    }
  });
  out.pr_ptr.assign(np + 1, 0);
  for (int p = 0; p < np; p++) out.pr_ptr[p + 1] = out.pr_ptr[p] + nruns[p];
  out.prun.resize(out.pr_ptr[np]);
  out.max_lds = 0;
  for (int p = 0; p < np; p++) out.max_lds = std::max(out.max_lds, maxlds[p]);

  // elements touching each patch (two passes: count, fill)
  auto gather = [&](int64_t p, std::vector<int32_t>& tmp) {
    tmp.clear();
    for (int32_t j = out.pn_ptr[p]; j < out.pn_ptr[p + 1]; j++) {
      const int32_t n = out.pnode[j];
      for (int32_t k = n2e_ptr[n]; k < n2e_ptr[n + 1]; k++) tmp.push_back(n2e_ent[k] / npe);
    }
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
  };
  par_for(np, [&](int64_t pb, int64_t pe) {
    std::vector<int32_t> tmp;
    for (int64_t p = pb; p < pe; p++) { gather(p, tmp); nelems[p] = (int32_t)tmp.size(); }
  });
  out.pe_ptr.assign(np + 1, 0);
  int64_t tot = 0;
  for (int p = 0; p < np; p++) { tot += nelems[p]; if (tot > 2000000000LL) return false; out.pe_ptr[p + 1] = (int32_t)tot; }
  const int RS = prec_stride(npe), OW = prec_own_words(npe), PW = prec_pos_words(npe);
  out.nrec = tot;
  out.prec.assign((size_t)tot * RS, 0);
  par_for(np, [&](int64_t pb, int64_t pe) {
    std::vector<int32_t> tmp;
    for (int64_t p = pb; p < pe; p++) {
      gather(p, tmp);
      for (size_t k = 0; k < tmp.size(); k++) {
        int32_t* rec = out.prec.data() + ((size_t)out.pe_ptr[p] + k) * RS;
        const int64_t e = tmp[k];
        rec[0] = (int32_t)e;
        uint16_t* own = (uint16_t*)(rec + 1 + npe);
        uint8_t* pp = (uint8_t*)(rec + 1 + npe + OW);
        for (int la = 0; la < npe; la++) {
          const int32_t n = conn[e * npe + la];
          rec[1 + la] = n;
          own[la] = (owner_patch[n] == p) ? (uint16_t)owner_idx[n] : (uint16_t)0xFFFF;
          for (int lb = 0; lb < npe; lb++) pp[la * PW * 4 + lb] = pos[(e * npe + la) * npe + lb];
        }
      }
      const int32_t* nodes = out.pnode.data() + out.pn_ptr[p];
      const int cnt = out.pn_ptr[p + 1] - out.pn_ptr[p];
      int32_t* run = out.prun.data() + out.pr_ptr[p];
      int j0 = 0;
      for (int j = 1; j <= cnt; j++)
        if (j == cnt || nodes[j] != nodes[j - 1] + 1) { *run++ = j0 | ((j - j0) << 16); j0 = j; }
    }
  });
  out.lane_redundancy = nel > 0 ? (double)tot / (double)nel : 1.0;
  return true;
}

}  // namespace fa
