// Host-side check of the tile builder (falcon_amd/csrc/fa_tiles.hip, compiled as plain C++ by tests/test_tile_builder.py).
// usage: tile_builder_check RANK N ROW_ORDER REC_DOUBLES BIG [MAX_EL: element tiles]   -> prints "key value" lines, exit code 0 when every invariant holds
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <set>
#include <vector>

#include "fa_internal.h"
#include "fa_tiles.h"

int main(int argc, char** argv) {
  const int rank = argc > 1 ? atoi(argv[1]) : 3, n = argc > 2 ? atoi(argv[2]) : 12;
  const bool row_order = argc > 3 && atoi(argv[3]) != 0;
  const int npe = rank == 3 ? 8 : 4, nip = npe, nd = rank + 1;
  const int rec = argc > 4 ? atoi(argv[4]) : fa::tile_rec_doubles(rank, fa::FAM_PF_ISO, false);
  const bool big = argc > 5 && atoi(argv[5]) != 0;           // the 1024-thread / whole-LDS geometry
  const int budget = big ? fa::kTileLdsBig : fa::kTileLds;
  const int m = n + 1;
  const int64_t nn = rank == 3 ? (int64_t)m * m * m : (int64_t)m * m, nel = rank == 3 ? (int64_t)n * n * n : (int64_t)n * n;
  std::vector<double> x(nn * rank);
  std::vector<int32_t> conn(nel * npe);
  auto id = [&](int i, int j, int k) { return (int32_t)((k * m + j) * m + i); };
  for (int k = 0; k < (rank == 3 ? m : 1); k++) for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) {
    const int q = id(i, j, k);
    x[q * rank] = i + 0.1 * ((i * 7 + j * 3) % 5) / 5.0; x[q * rank + 1] = j;       // mildly irregular in x
    if (rank == 3) x[q * rank + 2] = k;
  }
  int64_t e = 0;
  for (int k = 0; k < (rank == 3 ? n : 1); k++) for (int j = 0; j < n; j++) for (int i = 0; i < n; i++, e++) {
    int32_t* c = &conn[e * npe];
    c[0] = id(i, j, k); c[1] = id(i + 1, j, k); c[2] = id(i + 1, j + 1, k); c[3] = id(i, j + 1, k);
    if (rank == 3) { c[4] = id(i, j, k + 1); c[5] = id(i + 1, j, k + 1); c[6] = id(i + 1, j + 1, k + 1); c[7] = id(i, j + 1, k + 1); }
  }
  std::vector<int32_t> n2e_ptr(nn + 1, 0), n2e_ent(nel * npe);
  for (int64_t k = 0; k < nel * npe; k++) n2e_ptr[conn[k] + 1]++;
  for (int64_t q = 0; q < nn; q++) n2e_ptr[q + 1] += n2e_ptr[q];
  { std::vector<int32_t> fill(n2e_ptr.begin(), n2e_ptr.end() - 1); for (int64_t k = 0; k < nel * npe; k++) n2e_ent[fill[conn[k]]++] = (int32_t)k; }
  std::vector<int64_t> nb_ptr(nn + 1, 0); std::vector<int32_t> nb;
  for (int64_t q = 0; q < nn; q++) {
    std::vector<int32_t> t;
    for (int32_t k = n2e_ptr[q]; k < n2e_ptr[q + 1]; k++) { const int32_t* c = &conn[(int64_t)(n2e_ent[k] / npe) * npe]; t.insert(t.end(), c, c + npe); }
    std::sort(t.begin(), t.end()); t.erase(std::unique(t.begin(), t.end()), t.end());
    nb.insert(nb.end(), t.begin(), t.end()); nb_ptr[q + 1] = (int64_t)nb.size();
  }
  if (argc > 6 && atoi(argv[6]) > 0) {
    // ELEMENT tiles (build_element_tiles): every element in exactly one tile, all its nodes in that tile's list, ascending
    // element order inside a tile, limits respected
    const int max_el = atoi(argv[6]);
    fa::TileSet U;
    const bool oku = fa::build_element_tiles(nn, nel, npe, nip, rank, nd, x.data(), conn.data(), fa::tile_node_doubles(rank, true), budget,
                                             max_el, 0, !row_order, false, U);
    printf("built %d\n", oku ? 1 : 0);
    if (!oku) return 2;
    std::vector<int> eseen(nel, 0);
    int64_t bad = 0, tnodes = 0;
    for (int t = 0; t < U.ntile; t++) {
      const int32_t* ti = &U.tinfo[(size_t)t * fa::kTileInfo];
      if (ti[2] != ti[1] || ti[4] > max_el || ti[4] <= 0) bad++;
      tnodes += ti[1];
      for (int i = 1; i < ti[1]; i++) if (U.tnodes[ti[0] + i - 1] >= U.tnodes[ti[0] + i]) bad++;       // distinct, ascending
      for (int s = 0; s < ti[4]; s++) {
        const uint32_t w = (uint32_t)U.telems[ti[3] + s]; const int32_t el = (int32_t)(w & 0x7fffffff);
        if (!(w >> 31)) bad++;
        eseen[el]++;
        if (s > 0 && (int32_t)(U.telems[ti[3] + s - 1] & 0x7fffffff) >= el) bad++;
        for (int a = 0; a < npe; a++) {
          const int li = U.tconn[((size_t)ti[3] + s) * U.conn_stride + a];
          if (li >= ti[1] || U.tnodes[ti[0] + li] != conn[(int64_t)el * npe + a]) bad++;
        }
      }
    }
    int64_t once = 0;
    for (int v : eseen) once += (v == 1);
    printf("ntile %d\nlds_bytes %d\nlds_budget %d\nredundancy %.4f\nelements %lld\nelements_once %lld\nbad %lld\nnodes_per_tile %.2f\n",
           U.ntile, U.lds_bytes, budget, U.point_redundancy, (long long)nel, (long long)once, (long long)bad, (double)tnodes / U.ntile);
    return (once == nel && bad == 0 && U.lds_bytes <= budget) ? 0 : 1;
  }
  fa::TileSet T;
  const bool ok = fa::build_tiles(nn, nel, npe, nip, rank, nd, x.data(), conn.data(), n2e_ptr.data(), n2e_ent.data(), nb_ptr.data(),
                                  nb.data(), rec, fa::tile_node_doubles(rank, true), budget, big ? fa::kTileBlockBig : fa::kTileBlock, row_order, T);
  printf("built %d\n", ok ? 1 : 0);
  if (!ok) return 2;
  std::vector<int> owned(nn, 0), ewr(nel, 0), eseen(nel, 0), blocks(nb.size(), 0);
  int64_t bad = 0, nmirror = 0, nlanes = 0, tel = 0;
  for (int t = 0; t < T.ntile; t++) {
    const int32_t* ti = &T.tinfo[(size_t)t * fa::kTileInfo];
    tel += ti[4];
    for (int i = 0; i < ti[2]; i++) owned[T.tnodes[ti[0] + i]]++;
    for (int s = 0; s < ti[4]; s++) {
      const uint32_t w = (uint32_t)T.telems[ti[3] + s]; const int32_t el = (int32_t)(w & 0x7fffffff);
      if (w >> 31) ewr[el]++;
      eseen[el]++;
      if (s > 0 && (int32_t)(T.telems[ti[3] + s - 1] & 0x7fffffff) >= el) bad++;            // ascending
      bool touches = false;
      for (int a = 0; a < npe; a++) {
        const int li = T.tconn[((size_t)ti[3] + s) * npe + a];
        if (li >= ti[1] || T.tnodes[ti[0] + li] != conn[(int64_t)el * npe + a]) bad++;
        touches = touches || li < ti[2];
      }
      if (!touches) bad++;                                                                   // halo elements touch an owned node
    }
    std::set<std::pair<int, int>> seen;
    for (int p = 0; p < ti[6]; p++) {
      const uint32_t* pw = &T.tpairs[((size_t)ti[5] + p) * 4];
      const int ia = pw[0] & 0xffff, k = (pw[0] >> 16) & 0xff, c = pw[0] >> 24;
      nlanes++;
      if (ia >= ti[2]) { bad++; continue; }
      const int32_t a = T.tnodes[ti[0] + ia], b = nb[nb_ptr[a] + k];
      if (!seen.insert({ ia, k }).second) bad++;
      blocks[nb_ptr[a] + k]++;
      if (pw[2] >> 31) {
        const int ib = pw[2] & 0xffff, k2 = (pw[2] >> 16) & 0xff;
        if (row_order || ib >= ti[2] || T.tnodes[ti[0] + ib] != b || nb[nb_ptr[b] + k2] != a) bad++; else blocks[nb_ptr[b] + k2]++;
        nmirror++;
      }
      if (c == 0 || pw[3] != T.tents[(size_t)ti[7] + pw[1]]) bad++;                            // first entry travels in the record
      int32_t prev = -1;
      for (int q = 0; q < c; q++) {
        const uint32_t en = T.tents[(size_t)ti[7] + pw[1] + q];
        const int slot = en & 0xffff, la = (en >> 16) & 0xff, lb = en >> 24;
        if (slot >= ti[4]) { bad++; continue; }
        const int32_t el = (int32_t)(T.telems[ti[3] + slot] & 0x7fffffff);
        if (conn[(int64_t)el * npe + la] != a || conn[(int64_t)el * npe + lb] != b || el <= prev) bad++;
        prev = el;
      }
    }
  }
  int64_t own1 = 0, ewr1 = 0, blk1 = 0, ecov = 0;
  for (int v : owned) own1 += (v == 1);
  for (int v : ewr) ewr1 += (v == 1);
  for (int v : eseen) ecov += (v >= 1);
  for (int v : blocks) blk1 += (v == 1);
  printf("ntile %d\nlds_bytes %d\nlds_budget %d\nredundancy %.4f\nnodes %lld\nowned_once %lld\nelements %lld\nelements_written_once %lld\n"
         "elements_covered %lld\nblocks %zu\nblocks_written_once %lld\nlanes %lld\nmirrored %lld\nbad %lld\ntile_elements %lld\n",
         T.ntile, T.lds_bytes, budget, T.point_redundancy, (long long)nn, (long long)own1, (long long)nel, (long long)ewr1,
         (long long)ecov, blocks.size(), (long long)blk1, (long long)nlanes, (long long)nmirror, (long long)bad, (long long)tel);
  const bool good = own1 == nn && ewr1 == nel && ecov == nel && blk1 == (int64_t)blocks.size() && bad == 0 && T.lds_bytes <= budget;
  return good ? 0 : 1;
}
