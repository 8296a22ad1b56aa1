// fa_tiles.hip -- host-side construction of the node tiles (see fa_tiles.h, fa_tile_impl.h).  Host code only.
#include "fa_tiles.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

#include "fa_internal.h"

namespace fa {
namespace {

template <class F>
void par_for(int64_t n, F&& body) {
  unsigned hw = std::thread::hardware_concurrency();
  int nt = (int)std::min<int64_t>(std::max(1u, std::min(hw, 16u)), std::max<int64_t>(1, n / 64));
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

inline uint64_t spread3(uint64_t v) {                  // 21 bits -> every third bit
  v &= 0x1FFFFFULL;
  v = (v | (v << 32)) & 0x1F00000000FFFFULL;
  v = (v | (v << 16)) & 0x1F0000FF0000FFULL;
  v = (v | (v << 8)) & 0x100F00F00F00F00FULL;
  v = (v | (v << 4)) & 0x10C30C30C30C30C3ULL;
  v = (v | (v << 2)) & 0x1249249249249249ULL;
  return v;
}
inline uint64_t spread2(uint64_t v) {                  // 31 bits -> every second bit
  v &= 0x7FFFFFFFULL;
  v = (v | (v << 16)) & 0x0000FFFF0000FFFFULL;
  v = (v | (v << 8)) & 0x00FF00FF00FF00FFULL;
  v = (v | (v << 4)) & 0x0F0F0F0F0F0F0F0FULL;
  v = (v | (v << 2)) & 0x3333333333333333ULL;
  v = (v | (v << 1)) & 0x5555555555555555ULL;
  return v;
}

struct Cutter {                                        // greedy cut of the sorted node sequence
  int64_t nnode, nel; int npe, nip, ndofpn, rec_doubles, node_doubles, lds_budget, tabn, extra_bytes;
  TileOpts opts;
  const int32_t* conn; const int32_t* n2e_ptr; const int32_t* n2e_ent; const int64_t* nb_ptr;
  const std::vector<int32_t>& order; const std::vector<uint64_t>& key;
  std::vector<int32_t> estamp, nstamp;

  int64_t bytes(int64_t n_el, int64_t n_all, int64_t n_own) const {
    const int64_t slot = opts.fvec > 0 ? 0 : (opts.slot_doubles > 0 ? opts.slot_doubles : tile_slot_doubles(nip, rec_doubles));
    return 8 * (n_el * slot + n_all * node_doubles + n_own * (opts.own_doubles + (opts.no_f_image ? 0 : ndofpn) + 2) + (opts.sym_all ? 2 * (n_all - n_own) : 0) + (opts.group > 1 ? ndofpn * (n_all - n_own) : 0) + (opts.no_tab ? 0 : tabn)) + 2 * n_el * (opts.conn_pad4 ? 4 : npe) + 16 + extra_bytes;
  }
  // Tiles end where the Morton cell of 2^L codes ends (once half full) or where the LDS budget / counters say so.
  // Returns false if one node alone does not fit.  starts: first position of every tile in `order` (+ sentinel).
  bool cut(int L, int64_t upto, std::vector<int64_t>& starts, std::vector<int32_t>* counts, int64_t& by_limit) {
    estamp.assign((size_t)nel, -1); nstamp.assign((size_t)nnode, -1);
    starts.clear(); by_limit = 0;
    if (counts) counts->clear();
    int32_t cur = 0;
    int64_t n_own = 0, n_all = 0, n_el = 0, n_pairs = 0, n_ents = 0;
    uint64_t cell0 = 0;
    std::vector<int32_t> undo_e, undo_n;
    auto close = [&]() {
      if (counts) { counts->push_back((int32_t)n_own); counts->push_back((int32_t)n_all); counts->push_back((int32_t)n_el);
                    counts->push_back((int32_t)n_pairs); counts->push_back((int32_t)n_ents); }
      cur++; n_own = n_all = n_el = n_pairs = n_ents = 0;
    };
    const int64_t min_own = opts.cell[0] > 0 ? std::max<int64_t>(1, (int64_t)opts.cell[0] * std::max(opts.cell[1], 1) * std::max(opts.cell[2], 1) / 2)
                                             : ((L >= 1) ? (int64_t(1) << (L - 1)) : 1);
    for (int64_t i = 0; i < upto; i++) {
      const int32_t n = order[(size_t)i];
      const uint64_t cell = (L >= 63) ? 0 : (key[(size_t)i] >> L);
      if (n_own > 0 && cell != cell0 && n_own >= min_own) close();
      for (int attempt = 0; attempt < 2; attempt++) {
        if (n_own == 0) { starts.push_back(i); cell0 = cell; }
        undo_e.clear(); undo_n.clear();
        int64_t add_all = 0, add_el = 0;
        if (nstamp[n] != cur) { nstamp[n] = cur; undo_n.push_back(n); add_all++; }
        for (int32_t t = n2e_ptr[n]; t < n2e_ptr[n + 1]; t++) {
          const int32_t el = n2e_ent[t] / npe;
          if (estamp[el] == cur) continue;
          estamp[el] = cur; undo_e.push_back(el); add_el++;
          for (int a = 0; a < npe; a++) {
            const int32_t m = conn[(int64_t)el * npe + a];
            if (nstamp[m] != cur) { nstamp[m] = cur; undo_n.push_back(m); add_all++; }
          }
        }
        const int64_t deg = nb_ptr[n + 1] - nb_ptr[n], val = n2e_ptr[n + 1] - n2e_ptr[n];
        const bool fits = bytes(n_el + add_el, n_all + add_all, n_own + 1) <= lds_budget && n_all + add_all <= 65535 &&
                          n_el + add_el <= 65535 && n_pairs + deg <= 2000000 && val <= 255 &&
                          (opts.max_el <= 0 || n_el + add_el <= opts.max_el) && (opts.max_own <= 0 || n_own + 1 <= opts.max_own) &&
                          (opts.fvec <= 0 || val <= opts.fvec) && (opts.max_all <= 0 || n_all + add_all <= opts.max_all) &&
                          (opts.max_deg_sum <= 0 || n_pairs + deg <= opts.max_deg_sum);
        if (fits) {
          n_own++; n_all += add_all; n_el += add_el;
          if (opts.fvec <= 0) { n_pairs += opts.diag_only ? 1 : deg; n_ents += opts.diag_only ? val : val * npe; }
          break;
        }
        for (int32_t e : undo_e) estamp[e] = -1;
        for (int32_t m : undo_n) nstamp[m] = -1;
        if (n_own == 0) return false;                  // a single node does not fit
        close(); by_limit++;
      }
    }
    if (n_own > 0) close();
    starts.push_back(upto);
    return true;
  }
};

}  // namespace

bool build_tiles(int64_t nnode, int64_t nel, int npe, int nip, int rank, int ndofpn, const double* coords,
                 const int32_t* conn, const int32_t* n2e_ptr, const int32_t* n2e_ent, const int64_t* nb_ptr,
                 const int32_t* nb, int rec_doubles, int node_doubles, int lds_budget, int block, bool row_order, TileSet& T,
                 const TileOpts& opts) {
  if (nnode <= 0 || nel <= 0 || npe > 255) return false;
  // ---- spatial order: Morton code of the coordinates quantised to the mean node spacing ----
  double lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
  for (int d = 0; d < rank; d++) { lo[d] = hi[d] = coords[d]; }
  for (int64_t n = 0; n < nnode; n++)
    for (int d = 0; d < rank; d++) { lo[d] = std::min(lo[d], coords[n * rank + d]); hi[d] = std::max(hi[d], coords[n * rank + d]); }
  double vol = 1.0, ext = 0.0;
  for (int d = 0; d < rank; d++) ext = std::max(ext, hi[d] - lo[d]);
  if (!(ext > 0.0)) return false;
  int live = 0;
  for (int d = 0; d < rank; d++) if (hi[d] - lo[d] > 1e-12 * ext) { vol *= hi[d] - lo[d]; live++; }
  const double h = std::pow(vol / (double)nnode, 1.0 / std::max(live, 1));
  const uint64_t qmax = (rank == 3) ? 0x1FFFFFULL : 0x7FFFFFFFULL;
  std::vector<uint64_t> code((size_t)nnode);
  par_for(nnode, [&](int64_t b, int64_t e) {
    for (int64_t n = b; n < e; n++) {
      uint64_t q[3] = { 0, 0, 0 };
      for (int d = 0; d < rank; d++) {
        const double v = std::floor((coords[n * rank + d] - lo[d]) / h + 0.5);
        q[d] = (uint64_t)std::min<double>(std::max(v, 0.0), (double)qmax);
      }
      if (opts.cell[0] > 0) {                           // box cells: Morton code of the box, then the node's place in it
        uint64_t cq[3] = { 0, 0, 0 }, lq = 0, mul = 1;
        for (int d = 0; d < rank; d++) {
          const uint64_t cd = (uint64_t)std::max(opts.cell[d], 1);
          cq[d] = q[d] / cd; lq += (q[d] % cd) * mul; mul *= cd;
        }
        const uint64_t mc = (rank == 3) ? (spread3(cq[0]) | (spread3(cq[1]) << 1) | (spread3(cq[2]) << 2))
                                        : (spread2(cq[0]) | (spread2(cq[1]) << 1));
        code[(size_t)n] = (mc << 8) | (lq & 0xFFu);
        continue;
      }
      code[(size_t)n] = (rank == 3) ? (spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2))
                                    : (spread2(q[0]) | (spread2(q[1]) << 1));
    }
  });
  std::vector<int32_t> order((size_t)nnode);
  for (int64_t n = 0; n < nnode; n++) order[(size_t)n] = (int32_t)n;
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return code[a] != code[b] ? code[a] < code[b] : a < b; });
  std::vector<uint64_t> key((size_t)nnode);
  for (int64_t i = 0; i < nnode; i++) key[(size_t)i] = code[order[(size_t)i]];
  code.clear(); code.shrink_to_fit();

  const int tabn = nip * npe * (1 + rank) + nip;
  const int order_mode = opts.order >= 0 ? opts.order : (row_order ? 1 : 0);
  Cutter C{ nnode, nel, npe, nip, ndofpn, rec_doubles, node_doubles, lds_budget, tabn, (order_mode != 0 && !opts.diag_only) ? tile_stage_bytes(ndofpn, block) : 0, opts, conn, n2e_ptr, n2e_ent, nb_ptr, order, key, {}, {} };
  // ---- cell size: the Morton cell (or none: tiles filled to the budget) that evaluates the fewest elements per owned node ----
  std::vector<int64_t> starts; int64_t by_limit = 0;
  const int64_t sample = std::min<int64_t>(nnode, 200000);
  int L = 63;
  {
    std::vector<int32_t> cs;
    double best = 1e300;
    const int cand[] = { 63, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2 };
    if (opts.cell[0] > 0) L = 8;                        // box cells: the low 8 bits number the nodes of a box
    else
    for (int tryL : cand) {
      if (!C.cut(tryL, sample, starts, &cs, by_limit)) return false;
      double el = 0.0, own = 0.0;
      for (size_t t = 0; t + 4 < cs.size() + 1; t += 5) { own += cs[t]; el += cs[t + 2]; }
      const double score = el / std::max(own, 1.0);
      if (score < best * 0.999) { best = score; L = tryL; }
    }
  }
  std::vector<int32_t> counts;
  if (!C.cut(L, nnode, starts, &counts, by_limit)) return false;
  const int64_t ntile = (int64_t)starts.size() - 1;
  if (ntile <= 0 || ntile > 0x7FFFFFF0LL) return false;
  C.estamp.clear(); C.estamp.shrink_to_fit(); C.nstamp.clear(); C.nstamp.shrink_to_fit();

  // ---- offsets ----
  std::vector<int64_t> o_node((size_t)ntile + 1, 0), o_el((size_t)ntile + 1, 0), o_pair((size_t)ntile + 1, 0), o_ent((size_t)ntile + 1, 0);
  int64_t max_bytes = 0, max_bytes_alt = 0;
  for (int64_t t = 0; t < ntile; t++) {
    const int32_t* c = counts.data() + t * 5;
    if (opts.own_doubles_alt >= 0)
      max_bytes_alt = std::max(max_bytes_alt, C.bytes(c[2], c[1], c[0]) - 8 * (int64_t)c[0] * (opts.own_doubles - opts.own_doubles_alt));
    o_node[t + 1] = o_node[t] + c[1]; o_el[t + 1] = o_el[t] + c[2]; o_pair[t + 1] = o_pair[t] + c[3]; o_ent[t + 1] = o_ent[t] + c[4];
    max_bytes = std::max(max_bytes, C.bytes(c[2], c[1], c[0]));
  }
  const int64_t lim = 0x7FFFFFF0LL;
  if (o_node[ntile] > lim || o_el[ntile] * 4 > lim || o_pair[ntile] * 4 > lim || o_ent[ntile] > lim) return false;
  T.ntile = (int)ntile; T.lds_bytes = (int)max_bytes; T.lds_bytes_alt = (int)max_bytes_alt;
  T.tinfo.assign((size_t)ntile * kTileInfo, 0);
  T.tnodes.assign((size_t)o_node[ntile], 0); T.telems.assign((size_t)o_el[ntile], 0);
  const int cs = opts.conn_pad4 ? 4 : npe;
  if (opts.conn_pad4 && npe > 4) return false;
  T.conn_stride = cs;
  T.tconn.assign((size_t)o_el[ntile] * cs, 0);
  for (int64_t t = 0; t < ntile; t++) {
    const int32_t* c = counts.data() + t * 5;
    T.max_own = std::max(T.max_own, c[0]); T.max_all = std::max(T.max_all, c[1]); T.max_el = std::max(T.max_el, c[2]);
  }
  const bool group_mode = opts.group > 1 && order_mode == 0 && opts.sym_all;
  std::vector<std::vector<uint32_t>> group_ents;       // group mode: the entry words of a tile are counted as they are made
  if (group_mode) group_ents.resize((size_t)ntile);
  const bool split_mode = (opts.split > 0 && order_mode == 0) || group_mode;
  std::vector<std::vector<uint32_t>> split_pairs;      // split mode: the lane count of a tile is known only after its pairs are cut
  if (split_mode) split_pairs.resize((size_t)ntile); else T.tpairs.assign((size_t)o_pair[ntile] * 4, 0);
  if (!group_mode) T.tents.assign((size_t)o_ent[ntile], 0);
  T.point_redundancy = (double)o_el[ntile] / (double)nel;
  const int esw = (npe + 3) / 4;                        // fvec mode: rank words per element (one byte per element node)
  if (opts.fvec > 0) { if (npe > 8) return false; T.eslot.assign((size_t)o_el[ntile] * esw, 0); T.nval.assign((size_t)o_node[ntile], 0); }

  // ---- fill ----
  std::atomic<int64_t> st_direct{0}, st_mown{0}, st_mhalo{0};
  par_for(ntile, [&](int64_t tb, int64_t te) {
    int64_t n_direct = 0, n_mown = 0, n_mhalo = 0;
    std::vector<int32_t> els, halo, own_sorted;
    std::vector<std::pair<int32_t, int32_t>> lookup;   // (node, local index), sorted by node
    struct PairTmp { uint32_t w0, w2; uint32_t first, cnt; bool diag; };
    std::vector<PairTmp> pt; std::vector<uint32_t> ent;
    for (int64_t t = tb; t < te; t++) {
      const int32_t* c = counts.data() + t * 5;
      const int n_own = c[0], n_all = c[1], n_el = c[2];
      const int32_t* own = order.data() + starts[(size_t)t];
      if (opts.sort_local) {
        own_sorted.assign(own, own + n_own);
        std::sort(own_sorted.begin(), own_sorted.end());
        own = own_sorted.data();
      }
      int32_t* ti = T.tinfo.data() + t * kTileInfo;
      ti[0] = (int32_t)o_node[t]; ti[1] = n_all; ti[2] = n_own; ti[3] = (int32_t)o_el[t]; ti[4] = n_el;
      ti[5] = (int32_t)o_pair[t]; ti[6] = c[3]; ti[7] = (int32_t)o_ent[t];
      els.clear();
      for (int i = 0; i < n_own; i++)
        for (int32_t k = n2e_ptr[own[i]]; k < n2e_ptr[own[i] + 1]; k++) els.push_back(n2e_ent[k] / npe);
      std::sort(els.begin(), els.end()); els.erase(std::unique(els.begin(), els.end()), els.end());
      lookup.clear();
      for (int i = 0; i < n_own; i++) lookup.emplace_back(own[i], i);
      std::sort(lookup.begin(), lookup.end());
      auto find_local = [&](int32_t node) -> int {
        auto it = std::lower_bound(lookup.begin(), lookup.end(), std::make_pair(node, (int32_t)-1));
        return (it != lookup.end() && it->first == node) ? it->second : -1;
      };
      halo.clear();
      for (int32_t el : els)
        for (int a = 0; a < npe; a++) { const int32_t m = conn[(int64_t)el * npe + a]; if (find_local(m) < 0) halo.push_back(m); }
      std::sort(halo.begin(), halo.end()); halo.erase(std::unique(halo.begin(), halo.end()), halo.end());
      int32_t* tn = T.tnodes.data() + o_node[t];
      for (int i = 0; i < n_own; i++) tn[i] = own[i];
      for (size_t i = 0; i < halo.size(); i++) { tn[n_own + i] = halo[i]; lookup.emplace_back(halo[i], n_own + (int32_t)i); }
      std::sort(lookup.begin(), lookup.end());
      int32_t* tel = T.telems.data() + o_el[t];
      uint16_t* tc = T.tconn.data() + o_el[t] * cs;
      for (size_t s = 0; s < els.size(); s++) {
        const int32_t* cn = conn + (int64_t)els[s] * npe;
        const int first = find_local(cn[0]);
        tel[s] = els[s] | ((first >= 0 && first < n_own) ? (int32_t)0x80000000 : 0);
        for (int a = 0; a < npe; a++) tc[s * cs + a] = (uint16_t)find_local(cn[a]);
      }
      if (opts.fvec > 0) {
        // force-vector mode: rank of every tile element in the (element-ascending) incident lists of its nodes
        uint32_t* es = T.eslot.data() + o_el[t] * esw;
        uint32_t* nv = T.nval.data() + o_node[t];
        for (int ia = 0; ia < n_own; ia++) {
          const int32_t a = own[ia];
          nv[ia] = (uint32_t)(n2e_ptr[a + 1] - n2e_ptr[a]);
          for (int32_t u = n2e_ptr[a]; u < n2e_ptr[a + 1]; u++) {
            const int32_t el = n2e_ent[u] / npe, la = n2e_ent[u] - el * npe;
            const size_t slot = (size_t)(std::lower_bound(els.begin(), els.end(), el) - els.begin());
            es[slot * esw + (la >> 2)] |= (uint32_t)(u - n2e_ptr[a]) << (8 * (la & 3));
          }
        }
        ti[6] = 0;
        continue;
      }
      // pairs (owned node, neighbour) with the elements that hold both, element-ascending
      pt.clear(); ent.clear();
      for (int ia = 0; ia < n_own; ia++) {
        const int32_t a = own[ia];
        const int64_t deg = nb_ptr[a + 1] - nb_ptr[a];
        for (int64_t k = 0; k < deg; k++) {
          const int32_t b = nb[nb_ptr[a] + k];
          PairTmp q; q.first = (uint32_t)ent.size(); q.cnt = 0; q.w2 = 0;
          const int ib = find_local(b);
          if (opts.diag_only && ib != ia) continue;
          if (opts.no_pair_share) {
            // every directed pair its own lane
          } else if (order_mode == 0 && opts.sym_all && ib != ia) {            // one lane per unordered pair, in the tile of the smaller id
            if (b < a) continue;
            const int32_t* lo2 = nb + nb_ptr[b]; const int32_t* hi2 = nb + nb_ptr[b + 1];
            const uint32_t k2 = (uint32_t)(std::lower_bound(lo2, hi2, a) - lo2);
            q.w2 = (uint32_t)ib | (k2 << 16) | 0x80000000u;
          } else if (order_mode == 0 && ib >= 0 && ib < n_own && ib != ia) {   // both owned: one lane for (a, b) and (b, a)
            if (ib < ia) continue;
            const int32_t* lo2 = nb + nb_ptr[b]; const int32_t* hi2 = nb + nb_ptr[b + 1];
            const uint32_t k2 = (uint32_t)(std::lower_bound(lo2, hi2, a) - lo2);
            q.w2 = (uint32_t)ib | (k2 << 16) | 0x80000000u;
          }
          for (int32_t u = n2e_ptr[a]; u < n2e_ptr[a + 1]; u++) {
            const int32_t el = n2e_ent[u] / npe, la = n2e_ent[u] - el * npe;
            const int32_t* cn = conn + (int64_t)el * npe;
            for (int lb = 0; lb < npe; lb++)
              if (cn[lb] == b) {
                const uint32_t slot = (uint32_t)(std::lower_bound(els.begin(), els.end(), el) - els.begin());
                ent.push_back(slot | ((uint32_t)la << 16) | ((uint32_t)lb << 24));
                q.cnt++;
                break;
              }
          }
          q.w0 = (uint32_t)ia | ((uint32_t)k << 16) | (q.cnt << 24);
          q.diag = (ib == ia);
          n_direct++;
          if (q.w2 & 0x80000000u) { if (ib >= 0 && ib < n_own) n_mown++; else n_mhalo++; }
          if (q.diag) q.w2 |= 0x40000000u;             // bit 30: diagonal block (the element-matrix kernel sums f_int there)
          pt.push_back(q);
        }
      }
      if (order_mode == 2) {
        // plain (node, neighbour) order: consecutive lanes hold consecutive chunks of the same CSR rows
      } else if (order_mode == 1) {
        // the diagonal blocks first (they hold the most elements), then the others in (node, neighbour) order: the
        // lanes of a wave then write consecutive chunks of the same CSR rows
        std::stable_sort(pt.begin(), pt.end(), [&](const PairTmp& x, const PairTmp& y) {
          if (x.diag != y.diag) return x.diag;
          return x.diag && x.cnt > y.cnt;
        });
      } else {
        // lanes of a wave: equal element counts (uniform trip count), then equal first elements (their record reads
        // become LDS broadcasts instead of bank conflicts)
        std::stable_sort(pt.begin(), pt.end(), [&](const PairTmp& x, const PairTmp& y) {
          if (x.cnt != y.cnt) return x.cnt > y.cnt;
          if (x.cnt == 0) return false;
          return (ent[x.first] & 0xFFFFu) < (ent[y.first] & 0xFFFFu);
        });
      }
      uint32_t* tq = group_mode ? nullptr : T.tents.data() + o_ent[t];
      uint32_t run = 0;
      if (group_mode) {
        // pairs of one owned node with identical element lists -> groups of up to opts.group neighbours
        const uint32_t split = opts.split > 0 ? (uint32_t)opts.split : 0xFFFFu;
        auto lanes_of = [&](uint32_t cnt) { return cnt > 2u * split ? 4u : (cnt > split ? 2u : 1u); };
        struct Grp { uint32_t ia, cnt, nb; uint32_t pi[4]; bool diag; };
        std::vector<Grp> gs;
        {
          std::vector<size_t> idx(pt.size());
          for (size_t i = 0; i < pt.size(); i++) idx[i] = i;
          auto same_list = [&](const PairTmp& x, const PairTmp& y) {
            if (x.cnt != y.cnt) return false;
            for (uint32_t k = 0; k < x.cnt; k++) if ((ent[x.first + k] & 0xFFFFu) != (ent[y.first + k] & 0xFFFFu)) return false;
            return true;
          };
          auto list_less = [&](size_t xi, size_t yi) {
            const PairTmp &x = pt[xi], &y = pt[yi];
            const uint32_t ax = x.w0 & 0xFFFFu, ay = y.w0 & 0xFFFFu;
            if (ax != ay) return ax < ay;
            if (x.diag != y.diag) return x.diag;
            if (x.cnt != y.cnt) return x.cnt > y.cnt;
            for (uint32_t k = 0; k < x.cnt; k++) {
              const uint32_t sx = ent[x.first + k] & 0xFFFFu, sy = ent[y.first + k] & 0xFFFFu;
              if (sx != sy) return sx < sy;
            }
            return xi < yi;
          };
          std::sort(idx.begin(), idx.end(), list_less);
          for (size_t q = 0; q < idx.size(); q++) {
            const PairTmp& x = pt[idx[q]];
            const uint32_t ia = x.w0 & 0xFFFFu;
            if (!gs.empty() && !x.diag && !gs.back().diag && gs.back().ia == ia && gs.back().nb < (uint32_t)std::min(opts.group, 4) &&
                same_list(pt[gs.back().pi[0]], x)) { gs.back().pi[gs.back().nb++] = (uint32_t)idx[q]; continue; }
            Grp g; g.ia = ia; g.cnt = x.cnt; g.nb = 1; g.pi[0] = (uint32_t)idx[q]; g.pi[1] = g.pi[2] = g.pi[3] = 0; g.diag = x.diag;
            gs.push_back(g);
          }
        }
        // lane order: groups of 4 / 2 / 1 lanes, then by neighbours per lane, by elements per lane, by first element
        std::stable_sort(gs.begin(), gs.end(), [&](const Grp& x, const Grp& y) {
          const uint32_t gx = lanes_of(x.cnt), gy = lanes_of(y.cnt);
          if (gx != gy) return gx > gy;
          if (x.nb != y.nb) return x.nb > y.nb;
          const uint32_t px = (x.cnt + gx - 1) / gx, py = (y.cnt + gy - 1) / gy;
          if (px != py) return px > py;
          if (x.cnt == 0 || y.cnt == 0) return false;
          return (ent[pt[x.pi[0]].first] & 0xFFFFu) < (ent[pt[y.pi[0]].first] & 0xFFFFu);
        });
        std::vector<uint32_t>& tp = split_pairs[(size_t)t];
        std::vector<uint32_t>& te = group_ents[(size_t)t];
        tp.clear(); te.clear();
        for (const Grp& g : gs) {
          const uint32_t gl = lanes_of(g.cnt), lg = gl == 4 ? 2u : (gl == 2 ? 1u : 0u), per = (g.cnt + gl - 1) / gl;
          for (uint32_t j = 0; j < gl; j++) {
            const uint32_t b0 = std::min(g.cnt, j * per), b1 = std::min(g.cnt, (j + 1) * per), c = b1 - b0;
            const uint32_t off = (uint32_t)te.size();
            for (uint32_t m = 0; m < g.nb; m++) {        // ib | position of a in b's row << 16 | position of b in a's row << 24
              const PairTmp& x = pt[g.pi[m]];
              te.push_back((x.w2 & 0x00FFFFFFu) | (((x.w0 >> 16) & 0xFFu) << 24));
            }
            for (uint32_t k = b0; k < b1; k++) {
              const uint32_t e0 = ent[pt[g.pi[0]].first + k];
              uint32_t lbw = 0;
              for (uint32_t m = 0; m < 4; m++) lbw |= (ent[pt[g.pi[m < g.nb ? m : 0]].first + k] >> 24) << (8 * m);
              te.push_back(e0 & 0x00FFFFFFu);            // slot | la << 16
              te.push_back(lbw);
            }
            tp.push_back(g.ia | (g.nb << 16) | (c << 24));
            tp.push_back(off);
            tp.push_back((g.diag ? 0x40000000u : 0u) | (lg << 24) | (j << 26));
            tp.push_back(c ? (ent[pt[g.pi[0]].first + b0] & 0x00FFFFFFu) : 0u);
          }
        }
        ti[6] = (int32_t)(tp.size() / 4);
        continue;
      }
      if (split_mode) {
        // groups of 4 / 2 / 1 lanes, largest groups first (a group never straddles a wave), then by elements per lane
        auto lanes_of = [&](uint32_t cnt) { return cnt > 2u * opts.split ? 4u : (cnt > (uint32_t)opts.split ? 2u : 1u); };
        std::stable_sort(pt.begin(), pt.end(), [&](const PairTmp& x, const PairTmp& y) {
          if (opts.lane_rows) {                        // by owned node first: the stores of a wave stay inside a few row blocks
            const uint32_t ax = x.w0 & 0xFFFFu, ay = y.w0 & 0xFFFFu;
            if (ax != ay) return ax < ay;
          }
          const uint32_t gx = lanes_of(x.cnt), gy = lanes_of(y.cnt);
          if (gx != gy) return gx > gy;
          const uint32_t px = (x.cnt + gx - 1) / gx, py = (y.cnt + gy - 1) / gy;
          if (px != py) return px > py;
          if (x.cnt == 0 || y.cnt == 0) return false;
          return (ent[x.first] & 0xFFFFu) < (ent[y.first] & 0xFFFFu);
        });
        std::vector<uint32_t>& tp = split_pairs[(size_t)t];
        tp.clear();
        for (size_t i = 0; i < pt.size(); i++) {
          if (opts.lane_rows && i > 0 && (pt[i].w0 & 0xFFFFu) != (pt[i - 1].w0 & 0xFFFFu)) {
            // a node's lanes start at a multiple of 4 (its 4- and 2-lane groups then sit on aligned quads / pairs): idle lanes --
            // no elements, marked as the second lane of a group so that they write nothing -- fill the gap
            while ((tp.size() / 4) % 4 != 0) { tp.push_back(pt[i - 1].w0 & 0xFFFFu); tp.push_back(run); tp.push_back(1u << 26); tp.push_back(0u); }
          }
          const uint32_t g = lanes_of(pt[i].cnt), lg = g == 4 ? 2u : (g == 2 ? 1u : 0u), per = (pt[i].cnt + g - 1) / g;
          for (uint32_t j = 0; j < g; j++) {
            const uint32_t b0 = std::min(pt[i].cnt, j * per), b1 = std::min(pt[i].cnt, (j + 1) * per), c = b1 - b0;
            tp.push_back(((pt[i].w0) & 0x00FFFFFFu) | (c << 24));
            tp.push_back(run + b0);
            tp.push_back(pt[i].w2 | (lg << 24) | (j << 26));
            tp.push_back(c ? ent[pt[i].first + b0] : 0u);
          }
          for (uint32_t k = 0; k < pt[i].cnt; k++) tq[run + k] = ent[pt[i].first + k];
          run += pt[i].cnt;
        }
        ti[6] = (int32_t)(tp.size() / 4);
        continue;
      }
      ti[6] = (int32_t)pt.size();
      uint32_t* tp = T.tpairs.data() + o_pair[t] * 4;
      for (size_t i = 0; i < pt.size(); i++) {
        tp[4 * i] = pt[i].w0; tp[4 * i + 1] = run; tp[4 * i + 2] = pt[i].w2; tp[4 * i + 3] = pt[i].cnt ? ent[pt[i].first] : 0u;
        for (uint32_t k = 0; k < pt[i].cnt; k++) tq[run + k] = ent[pt[i].first + k];
        run += pt[i].cnt;
      }
    }
    st_direct += n_direct; st_mown += n_mown; st_mhalo += n_mhalo;
  });
  T.blocks_direct = st_direct; T.blocks_mirror_own = st_mown; T.blocks_mirror_halo = st_mhalo;
  T.max_pairs = 0;
  for (int64_t t = 0; t < ntile; t++) T.max_pairs = std::max(T.max_pairs, (int)T.tinfo[(size_t)t * kTileInfo + 6]);
  if (group_mode) {                                    // concatenate the tiles' entry words
    int64_t total = 0;
    for (int64_t t = 0; t < ntile; t++) {
      T.tinfo[(size_t)t * kTileInfo + 7] = (int32_t)total;
      total += (int64_t)group_ents[(size_t)t].size();
      if (total > lim) return false;
    }
    T.tents.assign((size_t)std::max<int64_t>(total, 1), 0);
    par_for(ntile, [&](int64_t tb, int64_t te) {
      for (int64_t t = tb; t < te; t++) {
        const std::vector<uint32_t>& v = group_ents[(size_t)t];
        if (!v.empty()) std::memcpy(T.tents.data() + (size_t)T.tinfo[(size_t)t * kTileInfo + 7], v.data(), v.size() * 4);
      }
    });
  }
  if (split_mode) {                                    // concatenate the tiles' lane records
    int64_t total = 0;
    for (int64_t t = 0; t < ntile; t++) {
      T.tinfo[(size_t)t * kTileInfo + 5] = (int32_t)total;
      total += (int64_t)split_pairs[(size_t)t].size() / 4;
      if (total * 4 > lim) return false;
    }
    T.tpairs.assign((size_t)total * 4, 0);
    par_for(ntile, [&](int64_t tb, int64_t te) {
      for (int64_t t = tb; t < te; t++) {
        const std::vector<uint32_t>& v = split_pairs[(size_t)t];
        if (!v.empty()) std::memcpy(T.tpairs.data() + (size_t)T.tinfo[(size_t)t * kTileInfo + 5] * 4, v.data(), v.size() * 4);
      }
    });
  }
  return true;
}

bool build_element_tiles(int64_t nnode, int64_t nel, int npe, int nip, int rank, int ndofpn, const double* coords,
                         const int32_t* conn, int node_doubles, int lds_budget, int max_el, int max_all, bool with_tab,
                         bool conn_pad4, TileSet& T) {
  if (nnode <= 0 || nel <= 0 || npe > 255 || max_el <= 0) return false;
  if (conn_pad4 && npe > 4) return false;
  // ---- Morton order of the element centroids, quantised to the mean element spacing ----
  double lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };
  for (int d = 0; d < rank; d++) { lo[d] = hi[d] = coords[d]; }
  for (int64_t n = 0; n < nnode; n++)
    for (int d = 0; d < rank; d++) { lo[d] = std::min(lo[d], coords[n * rank + d]); hi[d] = std::max(hi[d], coords[n * rank + d]); }
  double vol = 1.0, ext = 0.0;
  for (int d = 0; d < rank; d++) ext = std::max(ext, hi[d] - lo[d]);
  if (!(ext > 0.0)) return false;
  int live = 0;
  for (int d = 0; d < rank; d++) if (hi[d] - lo[d] > 1e-12 * ext) { vol *= hi[d] - lo[d]; live++; }
  const double h = std::pow(vol / (double)nel, 1.0 / std::max(live, 1));
  const uint64_t qmax = (rank == 3) ? 0x1FFFFFULL : 0x7FFFFFFFULL;
  std::vector<uint64_t> code((size_t)nel);
  par_for(nel, [&](int64_t b, int64_t e) {
    for (int64_t el = b; el < e; el++) {
      uint64_t q[3] = { 0, 0, 0 };
      for (int d = 0; d < rank; d++) {
        double c = 0.0;
        for (int a = 0; a < npe; a++) c += coords[(int64_t)conn[el * npe + a] * rank + d];
        const double v = std::floor((c / npe - lo[d]) / h);
        q[d] = (uint64_t)std::min<double>(std::max(v, 0.0), (double)qmax);
      }
      code[(size_t)el] = (rank == 3) ? (spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2))
                                     : (spread2(q[0]) | (spread2(q[1]) << 1));
    }
  });
  std::vector<int32_t> order((size_t)nel);
  for (int64_t e = 0; e < nel; e++) order[(size_t)e] = (int32_t)e;
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return code[a] != code[b] ? code[a] < code[b] : a < b; });
  code.clear(); code.shrink_to_fit();

  const int cs = conn_pad4 ? 4 : npe;
  const int64_t tabn = with_tab ? (int64_t)nip * npe * (1 + rank) + nip : 0;
  auto bytes = [&](int64_t n_el, int64_t n_all) { return 8 * (n_all * (node_doubles + ndofpn + 2) + tabn) + 2 * n_el * cs + 16; };
  // ---- greedy cut ----
  std::vector<int64_t> starts;                          // first position of every tile in `order` (+ sentinel)
  std::vector<int32_t> n_alls;
  {
    std::vector<int32_t> nstamp((size_t)nnode, -1);
    int32_t cur = 0; int64_t n_el = 0, n_all = 0;
    for (int64_t i = 0; i < nel; i++) {
      const int32_t* cn = conn + (int64_t)order[(size_t)i] * npe;
      for (int attempt = 0; attempt < 2; attempt++) {
        int64_t add = 0;
        for (int a = 0; a < npe; a++) {
          bool seen = nstamp[cn[a]] == cur;
          for (int b = 0; b < a && !seen; b++) seen = cn[b] == cn[a];
          if (!seen) add++;
        }
        const bool fits = n_el + 1 <= max_el && (max_all <= 0 || n_all + add <= max_all) && n_all + add <= 65535 &&
                          bytes(n_el + 1, n_all + add) <= lds_budget;
        if (fits) {
          if (n_el == 0) starts.push_back(i);
          for (int a = 0; a < npe; a++) nstamp[cn[a]] = cur;
          n_el++; n_all += add;
          break;
        }
        if (n_el == 0) return false;                    // a single element does not fit
        n_alls.push_back((int32_t)n_all); cur++; n_el = 0; n_all = 0;
      }
    }
    if (n_el > 0) n_alls.push_back((int32_t)n_all);
    starts.push_back(nel);
  }
  const int64_t ntile = (int64_t)starts.size() - 1;
  if (ntile <= 0 || ntile > 0x7FFFFFF0LL || (int64_t)n_alls.size() != ntile) return false;
  std::vector<int64_t> o_node((size_t)ntile + 1, 0);
  int64_t max_bytes = 0;
  for (int64_t t = 0; t < ntile; t++) {
    const int64_t n_el = starts[(size_t)t + 1] - starts[(size_t)t];
    o_node[t + 1] = o_node[t] + n_alls[(size_t)t];
    max_bytes = std::max(max_bytes, bytes(n_el, n_alls[(size_t)t]));
    T.max_all = std::max(T.max_all, n_alls[(size_t)t]); T.max_own = T.max_all; T.max_el = std::max(T.max_el, (int)n_el);
  }
  if (o_node[ntile] > 0x7FFFFFF0LL || nel * 4 > 0x7FFFFFF0LL) return false;
  T.ntile = (int)ntile; T.lds_bytes = (int)max_bytes; T.lds_bytes_alt = (int)max_bytes;
  T.conn_stride = cs; T.point_redundancy = 1.0;
  T.tinfo.assign((size_t)ntile * kTileInfo, 0);
  T.tnodes.assign((size_t)o_node[ntile], 0); T.telems.assign((size_t)nel, 0);
  T.tconn.assign((size_t)nel * cs, 0);
  T.tpairs.assign(4, 0); T.tents.assign(1, 0);
  if (!with_tab) { T.eslot.assign((size_t)nel * ((npe + 3) / 4), 0); T.nval.assign((size_t)o_node[ntile], 0); }
  par_for(ntile, [&](int64_t tb, int64_t te) {
    std::vector<int32_t> nodes;
    for (int64_t t = tb; t < te; t++) {
      const int64_t e0 = starts[(size_t)t], n_el = starts[(size_t)t + 1] - e0;
      nodes.clear();
      for (int64_t s = 0; s < n_el; s++)
        for (int a = 0; a < npe; a++) nodes.push_back(conn[(int64_t)order[(size_t)(e0 + s)] * npe + a]);
      std::sort(nodes.begin(), nodes.end()); nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
      const int n_all = (int)nodes.size();                 // == n_alls[t]
      int32_t* ti = T.tinfo.data() + t * kTileInfo;
      ti[0] = (int32_t)o_node[t]; ti[1] = n_all; ti[2] = n_all; ti[3] = (int32_t)e0; ti[4] = (int32_t)n_el;
      std::copy(nodes.begin(), nodes.end(), T.tnodes.begin() + o_node[t]);
      // the elements of a tile in ascending id (like the node tiles): on meshes numbered row by row neighbouring lanes then
      // read neighbouring history values
      std::sort(order.begin() + e0, order.begin() + e0 + n_el);
      for (int64_t s = 0; s < n_el; s++) {
        const int32_t el = order[(size_t)(e0 + s)];
        T.telems[(size_t)(e0 + s)] = el | (int32_t)0x80000000;     // the one tile of the element stores its per-point outputs
        for (int a = 0; a < npe; a++)
          T.tconn[(size_t)(e0 + s) * cs + a] = (uint16_t)(std::lower_bound(nodes.begin(), nodes.end(), conn[(int64_t)el * npe + a]) - nodes.begin());
      }
    }
  });
  return true;
}

}  // namespace fa
