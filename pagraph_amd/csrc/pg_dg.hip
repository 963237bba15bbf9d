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
//   * arg-max        = dg.py:30-35 on np.argsort(score)[-2:] with numpy's default, unstable kind: its scalar
//     introsort restated below (np_argsort_f64) — an insertion sort (stable) up to 16 partitions, median-of-3
//     partitions above, which fixes the tie order for every P <= 127 (belongs is int8);
//   * bookkeeping    = dg.py:76-83.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "pg_common.h"

// ---- hops == 2 with builder threads + one committer ------------------------------------------------------------
// dg.py --num-hops 2 (README.md:117: the value for a 2-layer model without preprocessing) visits the two-hop
// in-neighbourhood of every train vertex: sum(deg^2) = 4.7e10 adjacency entries on the 10M/100M graph, 500 s in
// one thread. The dependency of the algorithm is between VERTICES (each assignment reads belongs / r_vnum as all
// earlier ones left them), but the expensive part — building the de-duplicated neighbour set N_i — depends on the
// graph only. So n_threads - 1 BUILDER threads run ahead, each taking whole vertices in train order, with a private
// V-bit bitmap (no shared writes: a team sharing one bitmap inside a vertex spent 80 cycles per visit on cache-line
// ping-pong and two barriers per vertex, and was no faster than one thread), and hand N_i over as a list of
// (64-bit word index, mask) pairs sorted by word; ONE committer thread consumes the lists strictly in train order:
// per word it counts the already-assigned members (one cache line of `belongs`), and after the arg-max merges the
// mask into the chosen partition's redundancy bitmap with a popcount of the new bits (dg.py:79-83: 64 vertices per
// operation). Same set arithmetic as the sequential code: bit-identical partitions (tests/golden/g4_*, and the
// sequential / threaded comparison in tests/test_host_logic.py).
namespace {

// np.argsort(v) (default kind) of n float64 on numpy's portable scalar path — numpy 2.2 npysort/quicksort.cpp aquicksort_,
// heapsort.cpp aheapsort_, restated from the algorithm: median-of-3 Hoare partitions while a range spans more than 16
// elements, insertion sort below, heapsort once the depth budget 2 * floor(log2 n) is spent. NaNs sort last. (On AVX2 /
// AVX-512 hosts numpy dispatches to x86-simd-sort, whose tie order differs: the reference's partition is machine-dependent;
// this is the order the fixtures pin — oracle/gen_golden.py runs the reference with the dispatch disabled.)
inline bool np_less(double a, double b) { return a < b || (b != b && a == a); }

void np_aheapsort(const double* v, int32_t* tosort, int n) {
  int32_t* a = tosort - 1;                       // 1-based
  for (int l = n >> 1; l > 0; --l) {
    const int32_t tmp = a[l];
    int i = l, j = l << 1;
    while (j <= n) {
      if (j < n && np_less(v[a[j]], v[a[j + 1]])) ++j;
      if (np_less(v[tmp], v[a[j]])) { a[i] = a[j]; i = j; j += j; }
      else break;
    }
    a[i] = tmp;
  }
  while (n > 1) {
    const int32_t tmp = a[n];
    a[n] = a[1];
    --n;
    int i = 1, j = 2;
    while (j <= n) {
      if (j < n && np_less(v[a[j]], v[a[j + 1]])) ++j;
      if (np_less(v[tmp], v[a[j]])) { a[i] = a[j]; i = j; j += j; }
      else break;
    }
    a[i] = tmp;
  }
}

void np_argsort_f64(const double* v, int n, int32_t* t) {
  for (int i = 0; i < n; ++i) t[i] = i;
  if (n < 2) return;
  int pl = 0, pr = n - 1;
  int stack[128], depth[64], sp = 0, dp = 0;
  int cdepth = 0;
  for (unsigned u = (unsigned)n; u >>= 1;) ++cdepth;
  cdepth *= 2;
  for (;;) {
    if (cdepth < 0) {
      np_aheapsort(v, t + pl, pr - pl + 1);
    } else {
      while (pr - pl > 15) {
        const int pm = pl + ((pr - pl) >> 1);
        if (np_less(v[t[pm]], v[t[pl]])) std::swap(t[pm], t[pl]);
        if (np_less(v[t[pr]], v[t[pm]])) std::swap(t[pr], t[pm]);
        if (np_less(v[t[pm]], v[t[pl]])) std::swap(t[pm], t[pl]);
        const double vp = v[t[pm]];
        int pi = pl, pj = pr - 1;
        std::swap(t[pm], t[pj]);
        for (;;) {
          do ++pi; while (np_less(v[t[pi]], vp));
          do --pj; while (np_less(vp, v[t[pj]]));
          if (pi >= pj) break;
          std::swap(t[pi], t[pj]);
        }
        std::swap(t[pi], t[pr - 1]);
        if (pi - pl < pr - pi) { stack[sp++] = pi + 1; stack[sp++] = pr; pr = pi - 1; }   // the larger part waits
        else { stack[sp++] = pl; stack[sp++] = pi - 1; pl = pi + 1; }
        depth[dp++] = --cdepth;
      }
      for (int pi = pl + 1; pi <= pr; ++pi) {
        const int32_t vi = t[pi];
        const double vp = v[vi];
        int pj = pi;
        while (pj > pl && np_less(vp, v[t[pj - 1]])) { t[pj] = t[pj - 1]; --pj; }
        t[pj] = vi;
      }
    }
    if (sp == 0) break;
    pr = stack[--sp];
    pl = stack[--sp];
    cdepth = depth[--dp];
  }
}

struct Dg2Slot {
  std::vector<int32_t> word;
  std::vector<uint64_t> mask;
  std::vector<int64_t> com;         // members assigned before the builder's snapshot, per partition
  std::vector<int32_t> pending;     // members assigned between the snapshot and this vertex's turn: the committer counts them
  std::atomic<int64_t> ready{-1};   // train index whose set is stored here
};

struct Dg2Shared {
  int64_t V;
  const int64_t* indptr;
  const int32_t* indices;
  const int64_t* train;
  int64_t n_train;
  int32_t P;
  const int8_t* belongs;                 // written by the committer only, before it publishes `committed`
  std::vector<int32_t> train_pos;        // vertex -> index in the train order (-1: not a train vertex)
  std::vector<Dg2Slot> slots;
  std::atomic<int64_t> next_build{0};
  std::atomic<int64_t> committed{0};     // vertices the committer is done with (their slots may be reused)
  std::atomic<bool> stop{false};
};

void dg2_builder(Dg2Shared* sh) {
  const int64_t W = (int64_t)sh->slots.size();
  std::vector<uint64_t> bits((size_t)(sh->V + 63) / 64, 0);
  std::vector<int32_t> touched;
  touched.reserve(1 << 16);
  auto add_list = [&](int64_t u) {
    const int32_t* p = sh->indices + sh->indptr[u];
    const int32_t* e = sh->indices + sh->indptr[u + 1];
    for (; p < e; ++p) {
      const int32_t w = *p;
      uint64_t& word = bits[(size_t)w >> 6];
      if (word == 0) touched.push_back(w >> 6);
      word |= 1ull << (w & 63);
    }
  };
  for (;;) {
    const int64_t i = sh->next_build.fetch_add(1, std::memory_order_relaxed);
    if (i >= sh->n_train) return;
    // the slot is free once the committer has finished vertex i - W
    for (int spins = 0; sh->committed.load(std::memory_order_acquire) + W <= i; ++spins) {
      if (sh->stop.load(std::memory_order_relaxed)) return;
      if (spins > 256) std::this_thread::yield(); else __builtin_ia32_pause();
    }
    const int64_t nid = sh->train[i];
    touched.clear();
    add_list(nid);                                                       // dg.py:23  nids = [in(nid)]
    for (int64_t q = sh->indptr[nid]; q < sh->indptr[nid + 1]; ++q) add_list(sh->indices[q]);   // :24-26
    std::sort(touched.begin(), touched.end());
    Dg2Slot& s = sh->slots[(size_t)(i % W)];
    s.word.assign(touched.begin(), touched.end());
    s.mask.resize(touched.size());
    s.com.assign((size_t)sh->P, 0);
    s.pending.clear();
    // dg.py:47-50 needs belongs[w] as of this vertex's turn. A vertex is assigned exactly at its own turn, so the
    // members assigned by then are the train vertices with a smaller train index: those below the committer's
    // published progress `c` have their final value (counted here), the few in flight go to the committer.
    const int64_t c = sh->committed.load(std::memory_order_acquire);
    for (size_t k = 0; k < touched.size(); ++k) {
      uint64_t& word = bits[(size_t)touched[k]];
      const uint64_t m0 = word;
      s.mask[k] = m0;
      word = 0;
      const int64_t base = (int64_t)touched[k] << 6;
      for (uint64_t m = m0; m; m &= m - 1) {
        const int64_t w = base + __builtin_ctzll(m);
        const int64_t tp = sh->train_pos[(size_t)w];
        if (tp < 0 || tp >= i) continue;
        if (tp < c) ++s.com[(size_t)sh->belongs[w]];
        else s.pending.push_back((int32_t)w);
      }
    }
    s.ready.store(i, std::memory_order_release);
  }
}

}  // namespace

// np.argsort(v) as dg.py:31 gets it on numpy's scalar path (exported for the tests: pinned against numpy itself)
extern "C" int pg_np_argsort_f64(const double* v, int32_t n, int32_t* order) {
  if (!v || !order || n < 0 || n > 127) return PG_ERR_INVALID;
  np_argsort_f64(v, n, order);
  return PG_OK;
}

extern "C" int pg_dg_partition(int64_t V, const int64_t* indptr, const int32_t* indices,
                               const int64_t* train_nids, int64_t n_train, int32_t P, int32_t hops,
                               int8_t* belongs_out, uint8_t* r_mask_out, int64_t* p_vnum_out,
                               int64_t* r_vnum_out) {
  const char* e = getenv("PG_DG_THREADS");
  return pg_dg_partition_mt(V, indptr, indices, train_nids, n_train, P, hops, belongs_out, r_mask_out, p_vnum_out,
                            r_vnum_out, e ? atoi(e) : 1);
}

extern "C" int pg_dg_partition_mt(int64_t V, const int64_t* indptr, const int32_t* indices,
                                  const int64_t* train_nids, int64_t n_train, int32_t P, int32_t hops,
                                  int8_t* belongs_out, uint8_t* r_mask_out, int64_t* p_vnum_out,
                                  int64_t* r_vnum_out, int32_t n_threads) {
  if (V <= 0 || !indptr || !indices || n_train < 0 || (n_train > 0 && !train_nids) || !belongs_out) return PG_ERR_INVALID;
  if (P < 2 || P > 127 || hops < 1) return PG_ERR_INVALID;  // belongs is int8 (dg.py:63); argsort[-2:] needs P>=2
  if (hops == 2 && n_threads > 1 && n_train > 0) {
    for (int64_t i = 0; i < n_train; ++i)
      if (train_nids[i] < 0 || train_nids[i] >= V) return PG_ERR_INVALID;
    const int T = n_threads > 64 ? 64 : n_threads;
    Dg2Shared sh;
    sh.V = V; sh.indptr = indptr; sh.indices = indices; sh.train = train_nids; sh.n_train = n_train;
    sh.P = P; sh.belongs = belongs_out;
    sh.train_pos.assign((size_t)V, -1);
    for (int64_t i = n_train - 1; i >= 0; --i) sh.train_pos[(size_t)train_nids[i]] = (int32_t)i;   // first occurrence assigns
    if (n_train >= INT32_MAX) return PG_ERR_UNSUPPORTED;
    std::memset(belongs_out, 0xFF, (size_t)V);
    sh.slots = std::vector<Dg2Slot>((size_t)std::min<int64_t>(n_train, 64 * (int64_t)T));
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(dg2_builder, &sh);
    // ---- the committer: dg.py:71-83 in train order ---------------------------------------------------------
    const size_t n_words = (size_t)(V + 63) / 64;
    std::vector<std::vector<uint64_t>> rbits(P, std::vector<uint64_t>(n_words, 0));   // r_belongs (dg.py:64)
    std::vector<int64_t> p_vnum(P, 0), r_vnum(P, 0), com(P);
    std::vector<double> score(P);
    std::vector<int32_t> order(P);
    const double avg = (double)V * 0.65 / (double)P;
    const int64_t W = (int64_t)sh.slots.size();
    constexpr bool dbg = false;       // (committer / builder cycle counts: flip for a diagnosis build)
    uint64_t c_wait = 0, c_work = 0, n_bits = 0, n_wordsum = 0;
    for (int64_t i = 0; i < n_train; ++i) {
      Dg2Slot& sl = sh.slots[(size_t)(i % W)];
      const uint64_t tw0 = dbg ? __builtin_ia32_rdtsc() : 0;
      for (int spins = 0; sl.ready.load(std::memory_order_acquire) != i; ++spins)
        if (spins > 1024) std::this_thread::yield(); else __builtin_ia32_pause();
      const uint64_t tw1 = dbg ? __builtin_ia32_rdtsc() : 0;
      c_wait += tw1 - tw0;
      const int64_t nid = train_nids[i];
      const size_t nw = sl.word.size();
      for (int p = 0; p < P; ++p) com[p] = 1 + sl.com[(size_t)p];    // dg.py:47-50: the builder's share ...
      for (int32_t w : sl.pending) {                                  // ... + the members assigned while it was in flight
        const int8_t b = belongs_out[w];
        if (b >= 0) ++com[b];
      }
      for (int p = 0; p < P; ++p)                                       // dg.py:51-55
        score[p] = (double)com[p] * (-(double)p_vnum[p] + avg) / (double)(r_vnum[p] + 1);
      np_argsort_f64(score.data(), P, order.data());                    // dg.py:30-35 np.argsort(score)[-2:]
      const int32_t i0 = order[P - 2], i1 = order[P - 1];
      const int32_t ind = (score[i0] != score[i1]) ? i1 : ((p_vnum[i0] < p_vnum[i1]) ? i0 : i1);
      if (belongs_out[nid] == -1) {                                     // dg.py:76-83
        belongs_out[nid] = (int8_t)ind;
        ++p_vnum[ind];
        uint64_t* rb = rbits[ind].data();
        int64_t fresh = 0;
        for (size_t k = 0; k < nw; ++k) {
          uint64_t& w = rb[(size_t)sl.word[k]];
          fresh += __builtin_popcountll(sl.mask[k] & ~w);
          w |= sl.mask[k];
        }
        uint64_t& ws = rb[(size_t)nid >> 6];
        const uint64_t bit = 1ull << (nid & 63);
        if (!(ws & bit)) { ws |= bit; ++fresh; }
        r_vnum[ind] += fresh;
      }
      sh.committed.store(i + 1, std::memory_order_release);
      if (dbg) {
        c_work += __builtin_ia32_rdtsc() - tw1;
        n_wordsum += nw;
        for (size_t k = 0; k < nw; ++k) n_bits += (uint64_t)__builtin_popcountll(sl.mask[k]);
      }
    }
    if (dbg)
      fprintf(stderr, "[dg2] committer: Gcycles waiting for builders %.2f, working %.2f | set members %.3e in %.3e words\n",
              c_wait * 1e-9, c_work * 1e-9, (double)n_bits, (double)n_wordsum);
    sh.stop.store(true, std::memory_order_relaxed);
    for (auto& t : th) t.join();
    if (r_mask_out)
      for (int p = 0; p < P; ++p)
        for (int64_t v = 0; v < V; ++v) r_mask_out[(size_t)p * V + v] = (rbits[p][(size_t)v >> 6] >> (v & 63)) & 1u;
    if (p_vnum_out) std::copy(p_vnum.begin(), p_vnum.end(), p_vnum_out);
    if (r_vnum_out) std::copy(r_vnum.begin(), r_vnum.end(), r_vnum_out);
    return PG_OK;
  }
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
    // dg.py:30-35 — np.argsort(score) (numpy's scalar introsort: stable up to 16 partitions), then the top two
    np_argsort_f64(score.data(), P, order.data());
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
