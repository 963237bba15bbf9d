// dg: PaGraph's streaming greedy train-vertex partitioner
// (PaGraph/partition/dg.py:14-103), host C++.  The algorithm is inherently
// sequential (every assignment reads the state all previous ones wrote), so it
// is a tight single-thread loop with O(1) scratch per visited neighbour rather
// than a GPU kernel; the L-hop closure that follows it is what runs on the GPU.
//
// Faithful details (each pinned by tests/golden/g4_*):
//   * neighbour set  = dg.py:18-27 in_neighbors_hop, including its hops>=3 quirk
//     (`neighs = nids[-1]` is evaluated once per depth, so from depth 2 on only
//     the LAST appended adjacency list is expanded);
//   * score          = dg.py:47-55: com[p] = 1 + |{u in N : belongs[u]==p}|,
//     score[p] = com[p] * (avg - p_vnum[p]) / (r_vnum[p] + 1) in float64 with
//     avg = V*0.65/P evaluated left to right;
//   * arg-max        = dg.py:30-35 on np.argsort(score)[-2:]; numpy's default
//     sort is an insertion sort (stable) for n <= 16, which fixes the tie order;
//   * bookkeeping    = dg.py:76-83.
#include <algorithm>
#include <cstring>
#include <vector>

#include "pg_common.h"

extern "C" int pg_dg_partition(int64_t V, const int64_t* indptr, const int32_t* indices,
                               const int64_t* train_nids, int64_t n_train, int32_t P, int32_t hops,
                               int8_t* belongs_out, uint8_t* r_mask_out, int64_t* p_vnum_out,
                               int64_t* r_vnum_out) {
  if (V <= 0 || !indptr || !indices || n_train < 0 || (n_train > 0 && !train_nids) || !belongs_out) return PG_ERR_INVALID;
  if (P < 2 || P > 127 || hops < 1) return PG_ERR_INVALID;  // belongs is int8 (dg.py:63); argsort[-2:] needs P>=2
  if (P > 16) return PG_ERR_UNSUPPORTED;                     // numpy's argsort stops being stable beyond 16
  std::vector<uint8_t> r_local;
  uint8_t* r = r_mask_out;
  if (!r) {
    r_local.assign((size_t)P * V, 0);
    r = r_local.data();
  } else {
    std::memset(r, 0, (size_t)P * V);
  }
  std::memset(belongs_out, 0xFF, (size_t)V);
  std::vector<int64_t> p_vnum(P, 0), r_vnum(P, 0), com(P);
  std::vector<double> score(P);
  std::vector<int32_t> order(P);
  std::vector<int64_t> stamp(V, -1);
  std::vector<int32_t> nb;  // the neighbour multiset/set of the current vertex
  nb.reserve(1 << 16);
  const double avg = (double)V * 0.65 / (double)P;

  for (int64_t step = 0; step < n_train; ++step) {
    const int64_t nid = train_nids[step];
    if (nid < 0 || nid >= V) return PG_ERR_INVALID;
    nb.clear();
    if (hops == 1) {
      // dg.py:19-20 — the raw CSC column (no dedup)
      nb.assign(indices + indptr[nid], indices + indptr[nid + 1]);
    } else {
      // dg.py:22-27 — union of the appended adjacency lists, np.unique'd
      auto add_list = [&](int64_t u) {
        for (int64_t e = indptr[u]; e < indptr[u + 1]; ++e) {
          const int32_t w = indices[e];
          if (stamp[w] != step) {
            stamp[w] = step;
            nb.push_back(w);
          }
        }
      };
      add_list(nid);                    // depth 0: nids = [in(nid)]
      int64_t last = nid;               // vertex whose list is nids[-1]
      for (int depth = 1; depth < hops; ++depth) {
        const int64_t lb = indptr[last], le = indptr[last + 1];
        for (int64_t e = lb; e < le; ++e) {   // for n in neighs: nids.append(in(n))
          add_list(indices[e]);
        }
        if (le > lb) last = indices[le - 1];  // nids[-1] is now in(last neighbour)
      }
    }
    // dg.py:47-55
    std::fill(com.begin(), com.end(), 1);
    for (int32_t u : nb) {
      const int8_t b = belongs_out[u];
      if (b >= 0) ++com[b];
    }
    for (int p = 0; p < P; ++p)
      score[p] = (double)com[p] * (-(double)p_vnum[p] + avg) / (double)(r_vnum[p] + 1);
    // dg.py:30-35 — stable ascending argsort, then the top two
    for (int p = 0; p < P; ++p) order[p] = p;
    for (int i = 1; i < P; ++i) {  // insertion sort, as numpy does for n <= 16
      const int32_t x = order[i];
      int j = i - 1;
      while (j >= 0 && score[x] < score[order[j]]) {
        order[j + 1] = order[j];
        --j;
      }
      order[j + 1] = x;
    }
    const int32_t i0 = order[P - 2], i1 = order[P - 1];
    int32_t ind;
    if (score[i0] != score[i1]) ind = i1;
    else ind = (p_vnum[i0] < p_vnum[i1]) ? i0 : i1;
    // dg.py:76-83
    if (belongs_out[nid] == -1) {
      belongs_out[nid] = (int8_t)ind;
      ++p_vnum[ind];
      uint8_t* rp = r + (size_t)ind * V;
      for (int32_t u : nb)
        if (!rp[u]) { rp[u] = 1; ++r_vnum[ind]; }
      if (!rp[nid]) { rp[nid] = 1; ++r_vnum[ind]; }
    }
  }
  if (p_vnum_out) std::copy(p_vnum.begin(), p_vnum.end(), p_vnum_out);
  if (r_vnum_out) std::copy(r_vnum.begin(), r_vnum.end(), r_vnum_out);
  return PG_OK;
}
