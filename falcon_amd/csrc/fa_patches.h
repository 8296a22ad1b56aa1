// fa_patches.h -- host-side construction of the node patches the patch kernel works on.
#pragma once
#include <stdint.h>

#include <vector>

namespace fa {

struct PatchSet {
  int npatch = 0;
  std::vector<int32_t> pn_ptr, pe_ptr, pr_ptr;   // npatch+1 each
  std::vector<int32_t> pnode;                    // owned nodes, ascending inside a patch
  std::vector<int32_t> poff;                     // LDS offsets, (owned+1) entries per patch
  std::vector<int32_t> prec;                     // element records (prec_stride(npe) words each), elements ascending
  int64_t nrec = 0;
  std::vector<int32_t> prun;                     // runs of consecutive node ids: first | count << 16
  int max_lds = 0;                               // largest LDS image (doubles)
  double lane_redundancy = 0.0;                  // (patch elements * npe) / (nel * npe)
};

// Recursive coordinate bisection of the nodes into compact patches whose row blocks
// (block_len(node) doubles each) fit `cap_lds` doubles and that own at most `max_own` nodes.
// Returns false if a single node does not fit.
bool build_patches(int64_t nnode, int64_t nel, int npe, int rank, const double* coords, const int32_t* conn,
                   const int32_t* n2e_ptr, const int32_t* n2e_ent, const uint8_t* pos, const int64_t* row_ptr, int ndofpn,
                   int cap_lds, int max_own, PatchSet& out);

}  // namespace fa
