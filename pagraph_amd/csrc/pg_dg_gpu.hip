// dg with the neighbour sets built on the GPU (round 6) — PaGraph/partition/dg.py:14-103, hops 1 and 2, bit-identical to
// pg_dg_partition (tests: all G4 fixtures + 10^6-vertex graphs against the sequential code; checked once at 10^8 vertices /
// 10^9 edges, P = 8, hops 2: 56 s here, 1 915 s there, the same partition).
//
// What is sequential in dg is ONE decision per train vertex that reads p_vnum / r_vnum as the previous decision left them
// (dg.py:47-55,71-83). What is expensive is building N(v) — the de-duplicated one- or two-hop in-neighbourhood: sum(deg^2) =
// 4.7e10 adjacency entries on the 10M / 100M graph, 6.9e11 on config 5's — and that depends on the graph only. Round 3 put
// host threads on it (68 s at 10M through one committer). Here the device does it, in BATCHES of consecutive train vertices,
// against a SNAPSHOT of the assignment state taken at the batch's start, and hands the host exactly what the snapshot
// cannot know:
//
//   com0[v][p]   = |{u in N(v) : u assigned to p in the snapshot}|                       (dg.py:47-50, the settled part)
//   corr(v)      = the members of N(v) that belong to THIS batch and precede v            (assigned between snapshot and v's turn)
//   fresh(v, p)  = {u in N(v) : the redundancy set of partition p lacked u in the snapshot}, for the CANDIDATE partitions p of v
//
// The host then walks the batch in train order with the reference's float64 score and numpy's argsort (pg_np_argsort_f64):
// com[p] = 1 + com0[v][p] + |{u in corr(v) : belongs[u] == p}|; after the arg-max, r_vnum[ind] grows by the members of
// fresh(v, ind) that are still missing from the HOST's exact r_belongs[ind] bitmap (test-and-set) — fresh(v, ind) is a
// superset of N(v) \ r_belongs[ind] whatever the snapshot's age, so the count is exact. Every (vertex, candidate partition)
// pair has a list of its own — a range the vertex's workgroup reserves in that partition's buffer, after counting, with ONE
// atomic per partition — and the host reads the one list it needs per vertex (RMAT has no communities: four vertices of five
// have more than one candidate, and the partitions that lose lack far more members than the one that wins — 95 % of the
// entries are never looked at; nothing is sorted). The decisions go back, and one kernel applies them to the snapshot from
// the same lists (no second expansion). Early batches are tiny (the first hubs' sets are the whole graph and every partition
// lacks all of it); the batch doubles while the lists stay small, and a batch whose lists overflow their buffers is simply
// redone at half the size.
//
// De-duplication of a two-hop multiset: one bitmap per workgroup in HBM — 288 GB pays for a few hundred private ones even at
// 10^8 vertices (25 MB each). The expansion is bound by exactly that traffic: a multiset is sparse in the id space (mean 7 000
// entries, 10^8 ids), so every access is a 128-byte line of its own (PMC, profiles/r06/dg_gpu_expand_pmc.txt: 180 B fetched +
// 41 B written per entry visit in the first version, 222 + 47 per entry now). The first version walked every multiset twice —
// once to set bits and count, once more to clear them and list the fresh members — and paid that line twice. Now a word holds
// 16 members under a 16-bit GENERATION tag (the workgroup's count of multisets so far): a word whose tag is stale counts as
// empty, nothing is ever cleared (a wrap of the tag, every 65 000 multisets of a workgroup, zeroes its bitmap), and the members
// that lack some partition are put aside during the one walk (a per-workgroup scratch list) and filtered by the candidates
// afterwards. A vertex whose list does not fit the scratch is walked twice more (count, then place), each time under a
// generation of its own.
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#include "pg_common.h"

using namespace pg;

namespace {

constexpr int kDgThreads = 256;
constexpr int kMaxP = 16;              // a member's partition mask is 16 bits wide (scratch entries, the state word)
constexpr int kLongQueue = 1024;       // such lists queued per vertex (more: the finding wave walks them itself)
constexpr int kLongList = 1024;        // adjacency lists above this are walked by the whole workgroup, the others per wave

struct DgExpandArgs {
  const int64_t* indptr;
  const int32_t* indices;
  const int64_t* bv;          // batch vertices (train order)
  int32_t n;                  // batch size
  int32_t P;
  int32_t hops;
  const uint32_t* state;      // the snapshot, one word per vertex (ONE random read per member instead of two): bits 0-15: bit p =
                              // the vertex is in r_belongs[p]; bits 16-23: belongs as int8 (>= 0 partition, -1 unassigned,
                              // -2 member of this batch)
  uint32_t* pool;             // gridDim.x de-duplication bitmaps: 16 members per word under a 16-bit generation tag
  int64_t words;              // uint32 words per bitmap
  uint32_t* gens;             // [gridDim.x] the generation each workgroup's bitmap is at (kept across launches)
  unsigned long long* scratch;    // [gridDim.x][scr_cap] members of the current multiset that lack a partition: vertex << 16 | mask
  uint32_t scr_cap;
  int32_t* com0;              // [n][P]
  uint16_t* cand;             // [n] the partitions whose fresh members are listed for the vertex (see candidates())
  float a[kMaxP];             // (avg - p_vnum[p]) / (r_vnum[p] + 1) at the batch's start (dg.py:54-55 without com)
  float theta;                // a partition is a candidate when its best possible score >= theta x the best certain one
  int32_t all_candidates;     // 1: every partition for every vertex (near the end of the run, when a[] moves fast)
  uint32_t* fresh;            // [P][cap_fresh] members; fresh(v, p) = fresh[p][fbase[v][p] .. + fcnt[v][p])
  uint32_t* fcnt;             // [n][P]
  uint32_t* fbase;            // [n][P]
  unsigned long long* corr;   // keys: batch index << 32 | vertex
  unsigned long long cap_fresh, cap_corr;
  unsigned long long* counters;   // [1] corr, [2] next batch index, [3] multisets walked again, [4 + p] entries on partition p's lists
};
constexpr int kCounters = 4 + 16;

constexpr uint32_t kGenWrap = 0xFFF0u;     // a bitmap is zeroed when its generation gets here (two are used per vertex at most)
constexpr uint32_t kGenStart = 0xFF00u;    // where a run starts: every workgroup wraps early — the tests walk that path too

// first visit of member w in generation `gen`? (sets its bit; a word with another tag is an empty word)
__device__ __forceinline__ bool first_visit(uint32_t* bm, uint32_t gen, int32_t w) {
  uint32_t* p = bm + ((uint32_t)w >> 4);
  const uint32_t bit = 1u << (w & 15), tag = gen << 16;
  // (not from the CU's own cache: the line may still be there as an earlier multiset left it)
  uint32_t cur = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    const uint32_t base = (cur & 0xFFFF0000u) == tag ? cur : tag;
    if (base & bit) return false;
    const uint32_t seen = atomicCAS(p, cur, base | bit);
    if (seen == cur) return true;
    cur = seen;
  }
}

__device__ __forceinline__ void emit(unsigned long long* buf, unsigned long long* counter, unsigned long long cap, bool pred,
                                     unsigned long long key) {
  const unsigned long long m = __ballot(pred);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)m) - 1;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
  base = __shfl(base, leader);
  if (pred) {
    const unsigned long long at = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
    if (at < cap) buf[at] = key;
  }
}

// the walk, one member of the multiset: first visit -> which partition the snapshot has it in, or is it a batch member; a
// member some partition's redundancy set lacks is put aside with the mask of those partitions
template <bool DEDUP>
__device__ __forceinline__ void visit1(const DgExpandArgs& a, uint32_t* bm, uint32_t gen, int32_t* s_com, uint32_t* s_nscr,
                                       unsigned long long* scr, uint16_t full, int32_t i, int64_t v, bool live, int32_t w) {
  // the snapshot first: a member that is neither assigned nor of this batch and that every partition's set already holds
  // contributes nothing — it needs no de-duplication either, and its bitmap line is never touched (late in a run that is
  // every vertex outside the train set and most of the rest)
  int8_t b = -1;
  uint16_t lacks = 0;
  if (live) {
    const uint32_t st = a.state[w];
    b = (int8_t)(st >> 16);
    lacks = (uint16_t)(~st) & full;
  }
  bool first = live && (b != -1 || lacks != 0);
  if (DEDUP && first) first = first_visit(bm, gen, w);
  if (!first) {
    b = -1;
    lacks = 0;
  }
  if (b >= 0) atomicAdd(&s_com[b], 1);
  const bool pending = first && b == -2 && (int64_t)w < v;
  if (pending) atomicAdd(&s_com[kMaxP], 1);
  emit(a.corr, a.counters + 1, a.cap_corr, pending, ((unsigned long long)i << 32) | (uint32_t)w);
  const unsigned long long m = __ballot(lacks != 0);
  if (m) {
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(s_nscr, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    if (lacks) {
      const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      if (at < a.scr_cap) scr[at] = ((unsigned long long)(uint32_t)w << 16) | lacks;
    }
  }
}

// fresh(v, p), in two rounds over the members (both wave-uniform). tally: how many members does candidate p lack?
__device__ __forceinline__ void tally(int P, uint16_t cand, uint16_t miss, uint32_t* s_fcnt) {
  if (!__ballot(miss != 0)) return;
  for (int p = 0; p < P; ++p) {
    if (!((cand >> p) & 1)) continue;
    const unsigned long long m = __ballot((miss >> p) & 1);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(&s_fcnt[p], (uint32_t)__popcll(m));
  }
}
// place: the member goes into the range the vertex has reserved on each of those partitions' buffers
__device__ __forceinline__ void place(const DgExpandArgs& a, uint16_t cand, uint16_t miss, int32_t w, uint32_t* s_pos,
                                      const unsigned long long* s_base) {
  if (!__ballot(miss != 0)) return;
  const int lane = threadIdx.x & 63;
  for (int p = 0; p < a.P; ++p) {
    if (!((cand >> p) & 1)) continue;
    const bool pred = (miss >> p) & 1;
    const unsigned long long m = __ballot(pred);
    if (!m) continue;
    const int leader = __ffsll((long long)m) - 1;
    uint32_t off = 0;
    if (lane == leader) off = atomicAdd(&s_pos[p], (uint32_t)__popcll(m));
    off = __shfl(off, leader);
    if (pred) {
      const unsigned long long at = s_base[p] + off + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
      if (at < a.cap_fresh) a.fresh[(size_t)p * a.cap_fresh + at] = (uint32_t)w;
    }
  }
}

// the walks of a multiset whose put-aside list did not fit the scratch, each under a generation of its own (every member is
// met for the first time again): which candidates lack the member?
template <bool DEDUP>
__device__ __forceinline__ uint16_t revisit(const DgExpandArgs& a, uint32_t* bm, uint32_t gen, bool live, int32_t w, uint16_t cand) {
  const uint16_t miss = live ? (uint16_t)((uint16_t)(~a.state[w]) & cand) : (uint16_t)0;
  if (DEDUP && miss != 0 && !first_visit(bm, gen, w)) return 0;
  return miss;
}

// Which partitions can still win vertex i (dg.py:51-55,30-35)? score[p] = com[p] * a[p] with com[p] between 1 + com0[p] and
// 1 + com0[p] + (batch members among the neighbours), and a[p] a little below its value at the batch's start. A partition whose
// best case stays under theta x the best certain score is left out of the fresh list — the list then names, per member, only the
// CANDIDATES that lack it, instead of every partition that will never get the vertex anyway (a vertex 600 000 vertices of
// partition 3's set lack costs every neighbour of theirs an entry otherwise: 8.7e9 entries on the 10M / 100M graph, 70 GB over
// PCIe). It is a guess, and the host checks it: a decision outside the candidates ends the batch in front of that vertex.
__device__ __forceinline__ uint16_t candidates(const DgExpandArgs& a, const int32_t* s_com, int32_t i, uint16_t full) {
  if (a.all_candidates || i == 0) return full;       // (a batch's first vertex always has them all: the run cannot stall)
  float best = 0.f;
  for (int q = 0; q < a.P; ++q) best = fmaxf(best, (float)(1 + s_com[q]) * a.a[q]);
  uint16_t c = 0;
  for (int p = 0; p < a.P; ++p)
    if ((float)(1 + s_com[p] + s_com[kMaxP]) * a.a[p] >= a.theta * best) c |= (uint16_t)(1u << p);
  return c;
}

// walk the multiset of batch vertex v: in(v) (hops 1), plus in(u) for every u in in(v) (hops 2: dg.py:22-27). `f(live, w)` is
// called wave-uniformly (every lane of a wave calls it the same number of times; `live` says whether the lane holds a member).
template <typename F>
__device__ __forceinline__ void walk(const DgExpandArgs& a, int64_t v, int32_t* s_long, int32_t* s_nlong, F f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = kDgThreads / 64;
  const int64_t b0 = a.indptr[v], e0 = a.indptr[v + 1];
  // the vertex's own list, by the whole workgroup
  for (int64_t q = b0 + threadIdx.x; q - threadIdx.x < e0; q += kDgThreads) {
    const bool live = q < e0;
    f(live, live ? a.indices[q] : 0);
  }
  if (a.hops < 2) return;
  if (threadIdx.x == 0) *s_nlong = 0;
  __syncthreads();
  // short lists: one wave per neighbour; long ones are queued for the whole workgroup
  for (int64_t j = b0 + wave; j < e0; j += nwave) {
    const int32_t u = a.indices[j];
    const int64_t b1 = a.indptr[u], e1 = a.indptr[u + 1];
    if (e1 - b1 > kLongList) {
      int at = 0;
      if (lane == 0) {
        at = atomicAdd(s_nlong, 1);
        if (at < kLongQueue) s_long[at] = u;
      }
      if (__shfl(at, 0) < kLongQueue) continue;      // queued for the whole workgroup (a full queue: this wave walks it)
    }
    for (int64_t q = b1 + lane; q - lane < e1; q += 64) {
      const bool live = q < e1;
      f(live, live ? a.indices[q] : 0);
    }
  }
  __syncthreads();
  const int nl = *s_nlong < kLongQueue ? *s_nlong : kLongQueue;
  for (int k = 0; k < nl; ++k) {
    const int32_t u = s_long[k];
    const int64_t b1 = a.indptr[u], e1 = a.indptr[u + 1];
    for (int64_t q = b1 + threadIdx.x; q - threadIdx.x < e1; q += kDgThreads) {
      const bool live = q < e1;
      f(live, live ? a.indices[q] : 0);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(kDgThreads) void k_dg_expand(const DgExpandArgs a) {
  __shared__ int32_t s_com[kMaxP + 1];          // per partition: settled members; [kMaxP]: batch members in front of v
  __shared__ int32_t s_long[kLongQueue];
  __shared__ int32_t s_nlong;
  __shared__ int32_t s_i;
  __shared__ uint32_t s_nscr;
  __shared__ uint32_t s_fcnt[kMaxP], s_pos[kMaxP];
  __shared__ unsigned long long s_base[kMaxP];
  __shared__ uint16_t s_cand;
  uint32_t* bm = a.pool + (size_t)blockIdx.x * (size_t)a.words;
  unsigned long long* scr = a.scratch + (size_t)blockIdx.x * (size_t)a.scr_cap;
  const uint16_t full = (uint16_t)((1u << a.P) - 1u);
  uint32_t gen = a.gens[blockIdx.x];            // (every thread keeps the same count)
  for (;;) {
    if (threadIdx.x == 0) {
      s_i = (int32_t)atomicAdd(a.counters + 2, 1ull);
      s_nscr = 0;
    }
    if (threadIdx.x <= kMaxP) s_com[threadIdx.x] = 0;
    if (threadIdx.x < kMaxP) s_fcnt[threadIdx.x] = 0;
    __syncthreads();
    const int32_t i = s_i;
    if (i >= a.n) break;
    if (a.hops >= 2) {
      if (gen >= kGenWrap) {
        // the tag wraps: an empty bitmap, tag 0 (no generation has it)
        for (int64_t q = threadIdx.x; q < a.words; q += kDgThreads)
          __hip_atomic_store(&bm[q], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __syncthreads();
        gen = 0;
      }
      ++gen;
    }
    const int64_t v = a.bv[i];
    if (a.hops >= 2)
      walk(a, v, s_long, &s_nlong, [&](bool live, int32_t w) { visit1<true>(a, bm, gen, s_com, &s_nscr, scr, full, i, v, live, w); });
    else
      walk(a, v, s_long, &s_nlong, [&](bool live, int32_t w) { visit1<false>(a, bm, gen, s_com, &s_nscr, scr, full, i, v, live, w); });
    __syncthreads();
    if (threadIdx.x == 0) {
      s_cand = candidates(a, s_com, i, full);
      a.cand[i] = s_cand;
    }
    if (threadIdx.x < a.P) a.com0[(size_t)i * a.P + threadIdx.x] = s_com[threadIdx.x];
    __syncthreads();
    const uint16_t cand = s_cand;
    const uint32_t ns = s_nscr;
    const bool aside = ns <= a.scr_cap;
    // round 1: the members each candidate lacks, counted
    if (aside) {
      for (uint32_t q = threadIdx.x; q - threadIdx.x < ns; q += kDgThreads)
        tally(a.P, cand, q < ns ? (uint16_t)(scr[q] & 0xFFFFull) & cand : (uint16_t)0, s_fcnt);
    } else {
      if (threadIdx.x == 0) atomicAdd(a.counters + 3, 1ull);
      if (a.hops >= 2) {
        ++gen;                                   // (kGenWrap leaves room for two more)
        walk(a, v, s_long, &s_nlong, [&](bool live, int32_t w) { tally(a.P, cand, revisit<true>(a, bm, gen, live, w, cand), s_fcnt); });
      } else {
        walk(a, v, s_long, &s_nlong, [&](bool live, int32_t w) { tally(a.P, cand, revisit<false>(a, bm, gen, live, w, cand), s_fcnt); });
      }
    }
    __syncthreads();
    // one reservation per partition and vertex
    if (threadIdx.x < a.P) {
      const uint32_t c = s_fcnt[threadIdx.x];
      const unsigned long long base = c ? atomicAdd(a.counters + 4 + threadIdx.x, (unsigned long long)c) : 0ull;
      s_base[threadIdx.x] = base;
      s_pos[threadIdx.x] = 0;
      a.fcnt[(size_t)i * a.P + threadIdx.x] = c;
      a.fbase[(size_t)i * a.P + threadIdx.x] = (uint32_t)(base < 0xFFFFFFFFull ? base : 0xFFFFFFFFull);
    }
    __syncthreads();
    // round 2: into the reserved ranges
    if (aside) {
      for (uint32_t q = threadIdx.x; q - threadIdx.x < ns; q += kDgThreads) {
        const unsigned long long e = q < ns ? scr[q] : 0ull;
        place(a, cand, (uint16_t)(e & 0xFFFFull) & cand, (int32_t)(e >> 16), s_pos, s_base);
      }
    } else if (a.hops >= 2) {
      ++gen;
      walk(a, v, s_long, &s_nlong, [&](bool live, int32_t w) { place(a, cand, revisit<true>(a, bm, gen, live, w, cand), w, s_pos, s_base); });
    } else {
      walk(a, v, s_long, &s_nlong, [&](bool live, int32_t w) { place(a, cand, revisit<false>(a, bm, gen, live, w, cand), w, s_pos, s_base); });
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) a.gens[blockIdx.x] = gen;
}

__global__ void k_dg_mark(const int64_t* bv, int32_t n, uint32_t* state, int8_t value) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) state[bv[i]] = (state[bv[i]] & 0xFF00FFFFu) | ((uint32_t)(uint8_t)value << 16);
}

// the batch's decisions applied to the snapshot: belongs, and r_belongs[ind] |= N(v) + {v} from the list of (v, ind) itself.
// One 64-thread block per decided vertex.
__global__ __launch_bounds__(64) void k_dg_apply(const int64_t* bv, const int8_t* ind, int32_t P, const uint32_t* fresh,
                                                 unsigned long long cap_fresh, const uint32_t* fbase, const uint32_t* fcnt,
                                                 uint32_t* state) {
  const int32_t i = blockIdx.x, p = ind[i];
  if (threadIdx.x == 0) {
    // (other blocks set low bits of the same word meanwhile: two atomics, neither touches what the other writes)
    const int64_t v = bv[i];
    atomicAnd(&state[v], 0xFF00FFFFu);
    atomicOr(&state[v], ((uint32_t)(uint8_t)p << 16) | (1u << p));
  }
  const uint32_t* lst = fresh + (size_t)p * cap_fresh + fbase[(size_t)i * P + p];
  const uint32_t c = fcnt[(size_t)i * P + p];
  for (uint32_t q = threadIdx.x; q < c; q += 64) atomicOr(&state[lst[q]], 1u << p);
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? PG_OK : PG_ERR_NOMEM; }
  template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};
struct HostBuf {
  void* p = nullptr;
  ~HostBuf() { if (p) (void)hipHostFree(p); }
  int alloc(size_t bytes) { return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? PG_OK : PG_ERR_NOMEM; }
  template <typename T> T* as() { return reinterpret_cast<T*>(p); }
};

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

extern "C" int pg_dg_partition_gpu(int64_t V, const int64_t* indptr_dev, const int32_t* indices_dev,
                                   const int64_t* train_nids, int64_t n_train, int32_t P, int32_t hops,
                                   int8_t* belongs_out, uint8_t* r_mask_out, int64_t* vnum_out,
                                   pg_dg_gpu_stats_t* stats, pg_stream_t stream) {
  int64_t* p_vnum_out = vnum_out;                          // [2][P]: p_vnum, then r_vnum (dg.py:65-66)
  int64_t* r_vnum_out = vnum_out ? vnum_out + P : nullptr;
  if (V <= 0 || !indptr_dev || !indices_dev || n_train < 0 || (n_train > 0 && !train_nids) || !belongs_out) return PG_ERR_INVALID;
  if (P < 2 || P > 127 || hops < 1) return PG_ERR_INVALID;
  // what this path covers; the caller falls back to pg_dg_partition_mt otherwise (same result, host only)
  if (P > kMaxP || hops > 2 || V >= (1ll << 28)) return PG_ERR_UNSUPPORTED;
  for (int64_t i = 0; i < n_train; ++i) {
    if (train_nids[i] < 0 || train_nids[i] >= V) return PG_ERR_INVALID;
    if (i && train_nids[i] <= train_nids[i - 1]) return PG_ERR_UNSUPPORTED;      // ascending, distinct (dg.py:122 np.nonzero)
  }
  hipStream_t st = as_stream(stream);
  const double t_begin = now_s();
  int dev = 0, cus = 0;
  PG_HIP(hipGetDevice(&dev));
  PG_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int64_t words = (V + 15) / 16;
  // bitmaps: up to 4 workgroups per CU, at most ~16 GB of them
  // (2, 4 or 8 workgroups per CU expand at the same rate: profiles/r06/dg_gpu_sweep.txt; again with the tagged bitmaps at 10^8
  // vertices: 512 / 687 / 1 024 / 1 536 workgroups 38.4 / 38.7 / 40.8 / 42.0 s — the lines' traffic is the bound, not the waves in flight)
  int n_wg = hops >= 2 ? cus * 4 : cus * 8;
  if (hops >= 2) n_wg = (int)std::max<int64_t>(cus, std::min<int64_t>(n_wg, (16ll << 30) / (words * 4)));
  // the scratch list of a multiset's members that lack a partition: V / 256 entries (a larger list: two more walks), so that
  // small graphs — the tests' — take both paths
  const uint32_t scr_cap = (uint32_t)std::min<int64_t>(512 << 10, std::max<int64_t>(16, V / 256));
  const int32_t b_max = 1 << 15;       // (8 192 / 16 384 / 32 768 / 65 536: 108 / 95 / 93 / 97-100 s at 10^8 vertices; 6.0 / 5.5 / - / 6.2-6.4 s at 10^7)
  // entries per partition's list buffer: a single vertex's lists always fit (|N(v)| <= V each)
  const unsigned long long cap_fresh = (unsigned long long)std::max<int64_t>(V + 4096, std::min<int64_t>(32ll << 20, 64 * V));
  const unsigned long long cap_corr = 16ull << 20;

  DevBuf d_gens, d_scratch;
  if (d_gens.alloc((size_t)n_wg * 4) || d_scratch.alloc((size_t)n_wg * scr_cap * 8)) return PG_ERR_NOMEM;
  {
    std::vector<uint32_t> g0((size_t)n_wg, kGenStart);
    PG_HIP(hipMemcpy(d_gens.p, g0.data(), (size_t)n_wg * 4, hipMemcpyHostToDevice));
  }
  DevBuf d_state, d_pool, d_bv, d_com0, d_cand, d_fresh, d_corr, d_corr2, d_cnt, d_ind, d_tmp, d_fcnt, d_fbase;
  HostBuf h_fresh, h_corr, h_com0, h_cand, h_cnt, h_bv, h_ind, h_fcnt, h_fbase;
  if (d_state.alloc((size_t)V * 4) || d_pool.alloc(hops >= 2 ? (size_t)n_wg * words * 4 : 4) ||
      d_bv.alloc((size_t)b_max * 8) || d_com0.alloc((size_t)b_max * P * 4) || d_cand.alloc((size_t)b_max * 2) ||
      d_fresh.alloc((size_t)P * cap_fresh * 4) || d_fcnt.alloc((size_t)b_max * P * 4) || d_fbase.alloc((size_t)b_max * P * 4) ||
      d_corr.alloc(cap_corr * 8) || d_corr2.alloc(cap_corr * 8) || d_cnt.alloc(kCounters * 8) || d_ind.alloc((size_t)b_max))
    return PG_ERR_NOMEM;
  if (h_fresh.alloc((size_t)P * cap_fresh * 4) || h_fcnt.alloc((size_t)b_max * P * 4) || h_fbase.alloc((size_t)b_max * P * 4) ||
      h_corr.alloc(cap_corr * 8) || h_com0.alloc((size_t)b_max * P * 4) || h_cand.alloc((size_t)b_max * 2) ||
      h_cnt.alloc(kCounters * 8) || h_bv.alloc((size_t)b_max * 8) || h_ind.alloc((size_t)b_max))
    return PG_ERR_NOMEM;
  size_t tmp_bytes = 0;
  {
    unsigned long long* k = nullptr;
    PG_HIP(rocprim::radix_sort_keys(nullptr, tmp_bytes, k, k, (size_t)cap_corr, 0, 64, st));
  }
  if (d_tmp.alloc(tmp_bytes)) return PG_ERR_NOMEM;
  PG_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_state.p), 0x00FF0000, (size_t)V, st));      // unassigned, in no set
  if (hops >= 2) PG_HIP(hipMemsetAsync(d_pool.p, 0, (size_t)n_wg * words * 4, st));

  // host state (exact): dg.py:62-66
  const size_t n_words64 = (size_t)(V + 63) / 64;
  std::vector<std::vector<uint64_t>> rbits(P, std::vector<uint64_t>(n_words64, 0));
  std::vector<int64_t> p_vnum(P, 0), r_vnum(P, 0), com(P);
  std::vector<double> score(P);
  std::vector<int32_t> order(P);
  const double avg = (double)V * 0.65 / (double)P;
  std::memset(belongs_out, 0xFF, (size_t)V);

  pg_dg_gpu_stats_t s{};
  int64_t i0 = 0;
  int32_t bsz = 1;
  while (i0 < n_train) {
    const int32_t b = (int32_t)std::min<int64_t>(std::min<int64_t>(bsz, b_max), n_train - i0);
    std::memcpy(h_bv.p, train_nids + i0, (size_t)b * 8);
    const double t0 = now_s();
    PG_HIP(hipMemcpyAsync(d_bv.p, h_bv.p, (size_t)b * 8, hipMemcpyHostToDevice, st));
    PG_HIP(hipMemsetAsync(d_cnt.p, 0, kCounters * 8, st));
    hipLaunchKernelGGL(k_dg_mark, dim3((b + 255) / 256), dim3(256), 0, st, d_bv.as<int64_t>(), b, d_state.as<uint32_t>(), (int8_t)-2);
    DgExpandArgs a{};
    a.indptr = indptr_dev; a.indices = indices_dev; a.bv = d_bv.as<int64_t>(); a.n = b; a.P = P; a.hops = hops;
    a.state = d_state.as<uint32_t>(); a.pool = d_pool.as<uint32_t>(); a.words = words;
    a.gens = d_gens.as<uint32_t>(); a.scratch = d_scratch.as<unsigned long long>(); a.scr_cap = scr_cap;
    a.cand = d_cand.as<uint16_t>();
    {
      // dg.py:54-55 without com, as of now; near the end of the run (avg - p_vnum within a few batches of zero, or past it)
      // it moves too fast inside a batch for a guess: every partition is a candidate then
      bool fast = false;
      for (int p = 0; p < P; ++p) {
        const double room = avg - (double)p_vnum[p];
        a.a[p] = (float)(room / (double)(r_vnum[p] + 1));
        if (room < 8.0 * (double)b) fast = true;
      }
      a.theta = 0.95f;         // (0.8: 1 wrong guess and 1.11e9 list entries on the 10M / 100M graph; 0.95: 14 and 0.94e9, 9 % faster)
      a.all_candidates = (fast || b == 1) ? 1 : 0;
    }
    a.com0 = d_com0.as<int32_t>(); a.fresh = d_fresh.as<uint32_t>(); a.corr = d_corr.as<unsigned long long>();
    a.fcnt = d_fcnt.as<uint32_t>(); a.fbase = d_fbase.as<uint32_t>();
    a.cap_fresh = cap_fresh; a.cap_corr = cap_corr; a.counters = d_cnt.as<unsigned long long>();
    hipLaunchKernelGGL(k_dg_expand, dim3((unsigned)std::min<int>(n_wg, b)), dim3(kDgThreads), 0, st, a);
    PG_LAUNCH_CHECK();
    PG_HIP(hipMemcpyAsync(h_cnt.p, d_cnt.p, kCounters * 8, hipMemcpyDeviceToHost, st));
    PG_HIP(hipStreamSynchronize(st));
    const unsigned long long* n_list = h_cnt.as<unsigned long long>() + 4;      // entries on each partition's lists
    const unsigned long long n_corr = h_cnt.as<unsigned long long>()[1];
    unsigned long long n_fresh = 0, n_longest = 0;
    for (int p = 0; p < P; ++p) {
      n_fresh += n_list[p];
      n_longest = std::max(n_longest, n_list[p]);
    }
    s.second_walks += (int64_t)h_cnt.as<unsigned long long>()[3];
    s.seconds_expand += now_s() - t0;
    ++s.batches;

    if (n_longest > cap_fresh || n_corr > cap_corr) {
      // the lists did not fit: the same vertices again in a smaller batch (one vertex always fits: cap_fresh > V)
      hipLaunchKernelGGL(k_dg_mark, dim3((b + 255) / 256), dim3(256), 0, st, d_bv.as<int64_t>(), b, d_state.as<uint32_t>(), (int8_t)-1);
      ++s.batches_redone;
      if (b == 1) return PG_ERR_UNSUPPORTED;        // (cannot happen: see cap_fresh / a single vertex has no corr entries)
      bsz = std::max(1, b / 2);
      continue;
    }
    const double t1 = now_s();
    // group the corrections by batch index (a whole-key sort: rocPRIM's bit-range variant left lists of ~1 K keys ungrouped
    // on ROCm 7.0) and bring everything over
    unsigned long long* co = d_corr.as<unsigned long long>();
    if (b > 1 && n_corr > 1) {
      size_t tb = tmp_bytes;
      PG_HIP(rocprim::radix_sort_keys(d_tmp.p, tb, d_corr.as<unsigned long long>(), d_corr2.as<unsigned long long>(),
                                      (size_t)n_corr, 0, 64, st));
      co = d_corr2.as<unsigned long long>();
    }
    for (int p = 0; p < P; ++p)
      if (n_list[p])
        PG_HIP(hipMemcpyAsync(h_fresh.as<uint32_t>() + (size_t)p * cap_fresh, d_fresh.as<uint32_t>() + (size_t)p * cap_fresh,
                              (size_t)n_list[p] * 4, hipMemcpyDeviceToHost, st));
    PG_HIP(hipMemcpyAsync(h_fcnt.p, d_fcnt.p, (size_t)b * P * 4, hipMemcpyDeviceToHost, st));
    PG_HIP(hipMemcpyAsync(h_fbase.p, d_fbase.p, (size_t)b * P * 4, hipMemcpyDeviceToHost, st));
    if (n_corr) PG_HIP(hipMemcpyAsync(h_corr.p, co, (size_t)n_corr * 8, hipMemcpyDeviceToHost, st));
    PG_HIP(hipMemcpyAsync(h_com0.p, d_com0.p, (size_t)b * P * 4, hipMemcpyDeviceToHost, st));
    PG_HIP(hipMemcpyAsync(h_cand.p, d_cand.p, (size_t)b * 2, hipMemcpyDeviceToHost, st));
    PG_HIP(hipStreamSynchronize(st));
    s.seconds_lists += now_s() - t1;
    s.fresh_entries += (int64_t)n_fresh;
    s.corr_entries += (int64_t)n_corr;
    // ---- the committer: dg.py:71-83 in train order ------------------------------------------------------------------
    const double t2 = now_s();
    const uint32_t* hf = h_fresh.as<uint32_t>();
    const uint32_t* hfc = h_fcnt.as<uint32_t>();
    const uint32_t* hfb = h_fbase.as<uint32_t>();
    const unsigned long long* hc = h_corr.as<unsigned long long>();
    const int32_t* hcom = h_com0.as<int32_t>();
    int8_t* hind = h_ind.as<int8_t>();
    const uint16_t* hcand = h_cand.as<uint16_t>();
    unsigned long long qc = 0;
    int32_t done = 0;                      // vertices of this batch that were decided
    for (int32_t i = 0; i < b; ++i) {
      const int64_t nid = train_nids[i0 + i];
      for (int p = 0; p < P; ++p) com[p] = 1 + hcom[(size_t)i * P + p];
      for (; qc < n_corr && (int32_t)(hc[qc] >> 32) == i; ++qc) {
        // (belongs is V bytes touched at random: the line of a correction a few vertices ahead is asked for now)
        if (qc + 24 < n_corr) __builtin_prefetch(&belongs_out[(uint32_t)hc[qc + 24]], 0, 1);
        const int8_t bb = belongs_out[(uint32_t)hc[qc]];
        if (bb >= 0) ++com[bb];
      }
      for (int p = 0; p < P; ++p)
        score[p] = (double)com[p] * (-(double)p_vnum[p] + avg) / (double)(r_vnum[p] + 1);
      pg_np_argsort_f64(score.data(), P, order.data());
      const int32_t o0 = order[P - 2], o1 = order[P - 1];
      const int32_t ind = (score[o0] != score[o1]) ? o1 : ((p_vnum[o0] < p_vnum[o1]) ? o0 : o1);
      if (!((hcand[i] >> ind) & 1)) {
        // the device did not list the members partition `ind` lacks (it guessed that `ind` could not win): the batch ends
        // here, this vertex opens the next one — where it has every partition as a candidate
        ++s.candidate_misses;
        break;
      }
      hind[i] = (int8_t)ind;
      belongs_out[nid] = (int8_t)ind;
      ++p_vnum[ind];
      uint64_t* rb = rbits[ind].data();
      int64_t fresh = 0;
      const uint32_t* lst = hf + (size_t)ind * cap_fresh + hfb[(size_t)i * P + ind];
      const uint32_t nl = hfc[(size_t)i * P + ind];
      for (uint32_t q = 0; q < nl; ++q) {
        // (the list streams; the bitmap word it names does not: 12.5 MB per partition at 10^8 vertices, touched at random)
        if (q + 16 < nl) __builtin_prefetch(&rb[lst[q + 16] >> 6], 1, 1);
        const uint64_t u = lst[q];
        uint64_t& w = rb[u >> 6];
        const uint64_t ub = 1ull << (u & 63);
        if (!(w & ub)) { w |= ub; ++fresh; }
      }
      uint64_t& ws = rb[(size_t)nid >> 6];
      const uint64_t nb = 1ull << (nid & 63);
      if (!(ws & nb)) { ws |= nb; ++fresh; }
      r_vnum[ind] += fresh;
      done = i + 1;
    }
    if (done == b && qc != n_corr) {        // (a list that is not grouped by batch index: cannot happen)
      fprintf(stderr, "[pg_dg_partition_gpu] internal: batch at %lld of %d: corr %llu / %llu\n", (long long)i0, b, qc, n_corr);
      return PG_ERR_HIP;
    }
    if (done == 0) return PG_ERR_HIP;                           // (a batch's first vertex has every candidate)
    s.seconds_commit += now_s() - t2;
    // ---- decisions back to the snapshot --------------------------------------------------------------------------------
    const double t3 = now_s();
    PG_HIP(hipMemcpyAsync(d_ind.p, h_ind.p, (size_t)done, hipMemcpyHostToDevice, st));
    if (done < b)
      hipLaunchKernelGGL(k_dg_mark, dim3((b - done + 255) / 256), dim3(256), 0, st, d_bv.as<int64_t>() + done, b - done,
                         d_state.as<uint32_t>(), (int8_t)-1);
    hipLaunchKernelGGL(k_dg_apply, dim3((unsigned)done), dim3(64), 0, st, d_bv.as<int64_t>(), d_ind.as<int8_t>(), P,
                       d_fresh.as<uint32_t>(), cap_fresh, d_fbase.as<uint32_t>(), d_fcnt.as<uint32_t>(), d_state.as<uint32_t>());
    PG_LAUNCH_CHECK();
    PG_HIP(hipStreamSynchronize(st));        // (h_ind / h_bv are reused by the next batch)
    s.seconds_apply += now_s() - t3;
    i0 += done;
    s.largest_batch = std::max<int64_t>(s.largest_batch, done);
    // grow while the lists are far from their buffers and the guesses hold; a batch that ended early sets the size for the next
    if (done < b) bsz = std::max(1, std::max(done, b / 4));
    else if (n_longest < cap_fresh / 8 && n_corr < cap_corr / 8) bsz = std::min<int32_t>(b_max, std::max(bsz, b) * 2);
    else if (n_longest > cap_fresh / 2 || n_corr > cap_corr / 2) bsz = std::max(1, b / 2);
  }
  if (r_mask_out)
    for (int p = 0; p < P; ++p)
      for (int64_t v = 0; v < V; ++v) r_mask_out[(size_t)p * V + v] = (rbits[p][(size_t)v >> 6] >> (v & 63)) & 1u;
  if (p_vnum_out) std::copy(p_vnum.begin(), p_vnum.end(), p_vnum_out);
  if (r_vnum_out) std::copy(r_vnum.begin(), r_vnum.end(), r_vnum_out);
  s.seconds_total = now_s() - t_begin;
  s.workgroups = n_wg;
  if (stats) *stats = s;
  return PG_OK;
}
