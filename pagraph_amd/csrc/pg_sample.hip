// Per-layer frontier-expand neighbour sampler on a CSC adjacency in HBM.
//
// Stands in for DGL 0.4.1's C++/OpenMP NeighborSampler as the reference calls
// it at examples/profile/pa_gcn.py:71-76 (fan-out k, neighbor_type='in',
// num_hops) and, with fan-out = infinity, at PaGraph/partition/utils.py:11-18.
// DGL's source is not in the reference checkout: the semantics implemented here
// are the build-defined spec of DESIGN.md ("Sampler spec"), restated
// independently in oracle/pgc_oracle.c.
//
// Fan-out <= 64 (round 3): five launches per minibatch — k_sx (sample + relabel), k_bm_rank (bitmap -> ascending ids and
// word ranks), scans as decoupled look-backs inside them; see the block comment above k_sx. Wider fan-outs keep the
// round-1 chain, per block b (layer b+1 = destinations, layer b = sources), stream-ordered:
//   k_sample_wide wave per destination vertex: deg <= k takes the whole adjacency list (coalesced), else Floyd's
//                 k-subset with Philox4x32-10 draws keyed (seed; v, epoch, batch, layer, j), a vertex's tentative
//                 picks in an LDS strip; picks go to an ELL buffer and are marked in a V-bit bitmap (atomicOr) — the
//                 per-layer dedup.
//   k_scan2       exclusive scan of per-destination pick counts -> block indptr (+ the per-block popcounts' scan)
//   k_bm_count / k_bm_emit
//                 popcount prefix over the bitmap words: emits the layer's vertex ids ASCENDING (the spec's canonical
//                 order) and the per-word rank table; LDS holds the per-wave partials.
//   k_relabel     edge source id -> position in the layer = word_rank[w] + popcount(bits below) — no hash table, no sort.
// k_pack finally concatenates the layers (layer 0 first) into node_mapping and writes the sizes straight into pinned
// host memory. All sizes stay on the device; launches are sized by worst-case capacities.
#include <new>
#include <vector>

#include <cstdlib>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "pg_common.h"

namespace pg {

constexpr int kScanThreads = 1024;
constexpr int kMaxFanout = 4096;      // k_sample_wide: 4 waves x 4 * fanout bytes of LDS per block (64 KB)
constexpr int kWordsPerBlock = 1024;  // bitmap words handled by one 256-thread block (4 per thread)

// ---- block-wide exclusive scan of one int per thread (blockDim <= 1024) ----
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
// returns exclusive prefix; *total = block sum. lds: >= 16 ints
__device__ __forceinline__ int block_excl_scan(int v, int* lds, int* total) {
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave, nw = blockDim.x / kWave;
  const int incl = wave_incl_scan(v, lane);
  if (lane == kWave - 1) lds[w] = incl;
  __syncthreads();
  if (w == 0) {
    int s = lane < nw ? lds[lane] : 0;
    s = wave_incl_scan(s, lane);
    if (lane < nw) lds[lane] = s;
  }
  __syncthreads();
  const int base = w == 0 ? 0 : lds[w - 1];
  *total = lds[nw - 1];
  __syncthreads();
  return base + incl - v;
}

// ---------------------------------------------------------------------------
// what changes from one pg_sampler_sample call to the next. The launch sequence itself is fixed per output
// slot, so it is captured once into a hipGraph; the host writes this block (pinned, one per slot) and
// k_seed_layer copies it to the device for the rest of the chain.
struct SampleParams {
  const int64_t* seeds;
  int32_t n_seeds;
  uint32_t seed_lo, seed_hi, epoch, batch;
};

struct SampleArgs {
  const SampleParams* prm;  // device copy
  const int64_t* indptr;
  const int32_t* indices;
  const int64_t* dst_ids;  // layer b+1 vertex ids
  const int32_t* n_dst;    // device count
  int32_t* nbr;            // ELL [cap_dst, k] picked neighbour ids
  int32_t* cnt;            // [cap_dst]
  unsigned long long* bitmap;
  int32_t k;
  uint32_t layer;
};

// Fan-out above one wave (the reference accepts any --num-neighbors, pa_gcn.py:146-147): the same spec — Floyd's
// k-subset, draw j = word pair (j & 1) of Philox call j >> 1 — with a vertex's k tentative picks in LDS (one k-entry
// strip per wave) instead of one per lane. Draws are generated 64 at a time; the "already taken?" test of pick j scans
// the j resolved picks before it with the whole wave (ceil(j / 64) LDS reads per lane). O(k^2 / 64) per vertex, which
// a wide fan-out pays once per destination; the k <= 64 kernel above is unchanged.
__global__ __launch_bounds__(256) void k_sample_wide(const SampleArgs a) {
  extern __shared__ uint32_t sel_lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int k = a.k;
  volatile uint32_t* sel = sel_lds + (size_t)(threadIdx.x / kWave) * k;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
  const int n = *a.n_dst;
  for (int64_t p = wave0; p < n; p += nwaves) {
    const int64_t v = a.dst_ids[p];
    const int64_t beg = a.indptr[v];
    const int64_t deg = a.indptr[v + 1] - beg;
    int32_t* out = a.nbr + p * k;
    if (deg <= k) {
      for (int j = lane; j < deg; j += kWave) {
        const int32_t u = a.indices[beg + j];
        out[j] = u;
        atomicOr(&a.bitmap[u >> 6], 1ull << (u & 63));
      }
      if (lane == 0) a.cnt[p] = (int32_t)deg;
      continue;
    }
    for (int j = lane; j < k; j += kWave) {
      uint32_t r[4];
      Philox::gen((uint32_t)v, a.prm->epoch, a.prm->batch, (a.layer << 24) | (uint32_t)(j >> 1), a.prm->seed_lo,
                  a.prm->seed_hi, r);
      const uint64_t r64 = (j & 1) ? ((uint64_t)r[3] << 32 | r[2]) : ((uint64_t)r[1] << 32 | r[0]);
      sel[j] = (uint32_t)bounded(r64, (uint64_t)(deg - k + j) + 1);      // deg <= V < 2^31
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int j = 1; j < k; ++j) {
      const uint32_t tj = sel[j];
      bool dup = false;
      for (int i = lane; i < j; i += kWave) dup |= sel[i] == tj;
      if (__ballot(dup) != 0ull && lane == 0) sel[j] = (uint32_t)(deg - k + j);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    for (int j = lane; j < k; j += kWave) {
      const int32_t u = a.indices[beg + (int64_t)sel[j]];
      out[j] = u;
      atomicOr(&a.bitmap[u >> 6], 1ull << (u & 63));
    }
    if (lane == 0) a.cnt[p] = k;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the strip is rewritten for the wave's next vertex
  }
}

// exclusive scan of cnt[0:n] -> indptr[0:n+1]; single block, loops over tiles
template <bool CLEAR>
__global__ __launch_bounds__(kScanThreads) void k_scan_cnt(int32_t* __restrict__ cnt,
                                                           const int32_t* __restrict__ n_dev,
                                                           int32_t* __restrict__ indptr,
                                                           int32_t* __restrict__ total_out, int32_t cap_rows) {
  __shared__ int lds[16];
  const int n = *n_dev;
  int carry = 0;
  for (int base = 0; base < n; base += kScanThreads) {
    const int i = base + threadIdx.x;
    const int v = i < n ? cnt[i] : 0;
    if (CLEAR && i < n) cnt[i] = 0;
    int tot;
    const int ex = block_excl_scan(v, lds, &tot);
    if (i < n) indptr[i] = carry + ex;
    carry += tot;
  }
  // rows n..cap_rows are empty: a fixed-shape consumer (hipGraph replay) may run over all cap_rows
  for (int i = n + threadIdx.x; i <= cap_rows; i += kScanThreads) indptr[i] = carry;
  if (threadIdx.x == 0) *total_out = carry;
}

// the block's two single-block scans in one launch (both latency bound, one less link in the sampler's chain):
// per-destination pick counts -> indptr (as k_scan_cnt) and per-word-block popcounts -> exclusive offsets + layer
// size (as k_bm_scan)
__global__ __launch_bounds__(kScanThreads) void k_scan2(const int32_t* __restrict__ cnt,
                                                        const int32_t* __restrict__ n_dev,
                                                        int32_t* __restrict__ indptr, int32_t* __restrict__ total_out,
                                                        int32_t cap_rows, int32_t* __restrict__ partial, int n_blocks,
                                                        int32_t* __restrict__ count_out) {
  __shared__ int lds[16];
  const int n = *n_dev;
  int carry = 0;
  for (int base = 0; base < n; base += kScanThreads) {
    const int i = base + threadIdx.x;
    const int v = i < n ? cnt[i] : 0;
    int tot;
    const int ex = block_excl_scan(v, lds, &tot);
    if (i < n) indptr[i] = carry + ex;
    carry += tot;
  }
  // rows n..cap_rows are empty: a fixed-shape consumer (hipGraph replay) may run over all cap_rows
  for (int i = n + threadIdx.x; i <= cap_rows; i += kScanThreads) indptr[i] = carry;
  if (threadIdx.x == 0) *total_out = carry;
  __syncthreads();
  carry = 0;
  for (int base = 0; base < n_blocks; base += kScanThreads) {
    const int i = base + threadIdx.x;
    const int v = i < n_blocks ? partial[i] : 0;
    int tot;
    const int ex = block_excl_scan(v, lds, &tot);
    if (i < n_blocks) partial[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *count_out = carry;
}

// ---- bitmap -> ascending ids ------------------------------------------------
// (+ optionally: ANOTHER bitmap of the same size back to zero — the one the next layer's picks are marked in)
__global__ __launch_bounds__(256) void k_bm_count(const unsigned long long* __restrict__ bm, int64_t n_words,
                                                  int32_t* __restrict__ partial, unsigned long long* __restrict__ clr_bm) {
  __shared__ int lds[16];
  const int64_t w0 = (int64_t)blockIdx.x * kWordsPerBlock + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (w0 + i < n_words) {
      c += __popcll(bm[w0 + i]);
      if (clr_bm) clr_bm[w0 + i] = 0ull;
    }
  int tot;
  (void)block_excl_scan(c, lds, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// exclusive scan of the per-block partials in place; total -> *count_out (int32) and *count64 (optional)
__global__ __launch_bounds__(kScanThreads) void k_bm_scan(int32_t* __restrict__ partial, int n_blocks,
                                                          int32_t* __restrict__ count_out,
                                                          int64_t* __restrict__ count64) {
  __shared__ int lds[16];
  int carry = 0;
  for (int base = 0; base < n_blocks; base += kScanThreads) {
    const int i = base + threadIdx.x;
    const int v = i < n_blocks ? partial[i] : 0;
    int tot;
    const int ex = block_excl_scan(v, lds, &tot);
    if (i < n_blocks) partial[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    if (count_out) *count_out = carry;
    if (count64) *count64 = carry;
  }
}

__global__ __launch_bounds__(256) void k_bm_emit(const unsigned long long* __restrict__ bm, int64_t n_words,
                                                 const int32_t* __restrict__ partial,
                                                 int64_t* __restrict__ out_ids, int64_t cap,
                                                 uint32_t* __restrict__ word_rank) {
  __shared__ int lds[16];
  const int64_t w0 = (int64_t)blockIdx.x * kWordsPerBlock + threadIdx.x * 4;
  unsigned long long w[4];
  int c = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    w[i] = (w0 + i < n_words) ? bm[w0 + i] : 0ull;
    c += __popcll(w[i]);
  }
  int tot;
  int pos = partial[blockIdx.x] + block_excl_scan(c, lds, &tot);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (w0 + i < n_words) {
      if (word_rank) word_rank[w0 + i] = (uint32_t)pos;
      unsigned long long x = w[i];
      const int64_t vbase = (w0 + i) << 6;
      while (x) {
        const int b = __ffsll((long long)x) - 1;
        x &= x - 1;
        if (pos < cap) out_ids[pos] = vbase + b;
        ++pos;
      }
    }
  }
}

// ELL picks -> CSR block with sources relabelled to layer positions
__global__ __launch_bounds__(256) void k_relabel(const int32_t* __restrict__ nbr, const int32_t* __restrict__ cnt,
                                                 const int32_t* __restrict__ n_dst, int32_t k,
                                                 const int32_t* __restrict__ indptr,
                                                 const unsigned long long* __restrict__ bm,
                                                 const uint32_t* __restrict__ word_rank,
                                                 int32_t* __restrict__ blk_src) {
  const int64_t total = (int64_t)(*n_dst) * k;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / k;
    const int j = (int)(i - p * k);
    if (j < cnt[p]) {
      const int32_t u = nbr[i];
      const int64_t w = u >> 6;
      const unsigned long long below = bm[w] & ((1ull << (u & 63)) - 1ull);
      blk_src[indptr[p] + j] = (int32_t)(word_rank[w] + __popcll(below));
    }
  }
}

struct PackArgs {
  const int64_t* layer_ids[PG_MAX_LAYERS];
  const int32_t* layer_cnt[PG_MAX_LAYERS];
  const int32_t* blk_edges[PG_MAX_LAYERS];
  int64_t* node_mapping;
  int32_t* layer_offsets;
  int32_t* sizes_pinned;
  int32_t* sizes_dev;                 // optional device copy of the same numbers
  int64_t cap_nodes;
  int32_t num_layers;
  int32_t padded;                     // 1: layer l starts at pad_off[l], unused entries = -1
  int32_t pad_off[PG_MAX_LAYERS + 1];
};

__global__ __launch_bounds__(256) void k_pack(const PackArgs a) {
  int off[PG_MAX_LAYERS + 1], cnt[PG_MAX_LAYERS];
  off[0] = 0;
#pragma unroll
  for (int l = 0; l < PG_MAX_LAYERS; ++l) {
    cnt[l] = l < a.num_layers ? *a.layer_cnt[l] : 0;
    off[l + 1] = a.padded ? a.pad_off[l < a.num_layers ? l + 1 : a.num_layers] : off[l] + cnt[l];
  }
  if (a.padded) {
#pragma unroll
    for (int l = 0; l < PG_MAX_LAYERS; ++l) off[l] = a.pad_off[l < a.num_layers ? l : a.num_layers];
  }
  const int64_t total = a.padded ? a.pad_off[a.num_layers] : off[PG_MAX_LAYERS];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total && i < a.cap_nodes;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t v = -1;
#pragma unroll
    for (int l = 0; l < PG_MAX_LAYERS; ++l)
      if (l < a.num_layers && i >= off[l] && i < off[l] + cnt[l]) v = a.layer_ids[l][i - off[l]];
    a.node_mapping[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int l = 0; l < PG_MAX_LAYERS; ++l) {
      if (l <= a.num_layers) a.layer_offsets[l] = a.padded ? a.pad_off[l] : off[l];
      if (l < a.num_layers) a.sizes_pinned[l] = cnt[l];
      if (l + 1 < a.num_layers) a.sizes_pinned[PG_MAX_LAYERS + l] = *a.blk_edges[l];
      if (a.sizes_dev) {
        if (l < a.num_layers) a.sizes_dev[l] = cnt[l];
        if (l + 1 < a.num_layers) a.sizes_dev[PG_MAX_LAYERS + l] = *a.blk_edges[l];
      }
    }
    __threadfence_system();
  }
}

// ===========================================================================================================
// Round 3: the chain above as FIVE launches for fan-outs of at most one wave (was twelve), all plain launches with the
// call's scalars as kernel arguments (no pinned parameter block, no hipGraph, nothing for the host to wait for):
//   K1  S(top)            sample the seeds' in-neighbours            K2  R   bitmap -> ascending ids + word ranks
//   K3  S(next) + X(top)  sample layer L-1 + relabel block L-1       K4  R
//   K5  X(0) + sizes      relabel block 0, padding, sizes
// What made twelve: two single-block scans per layer (k_scan2, k_bm_scan) and two passes over the bitmap (count, emit).
// Both scans are now decoupled look-backs inside the kernel that produces the numbers: a block publishes its aggregate as
// one 8-byte {value, tag} granule (one sc1 store; tag = a per-launch number, so nothing is ever reset) and sums the granules
// of ALL its predecessors (at most 1024: one wave, a few loads per lane) — no serial chain, one hand-off deep. Blocks only
// ever wait for blocks dispatched before them.
//   S: a group of G = 2^ceil(log2 k) lanes per destination (k = 2: 32 destinations per wave; rounds 1-2 gave a vertex a
//      whole wave and idles 62 lanes) -> pick counts -> block scan + look-back -> the block's CSR offsets are known
//      while the picks are still in flight, so picks land in CSR position straight away (as vertex ids; no ELL buffer,
//      no cnt array, no k_scan_cnt).
//   R: popcount of the block's words -> look-back -> emit ids (straight into node_mapping for the fixed-shape layout)
//      and the per-word rank table; zeroes the OTHER bitmap, the one the next S marks into.
//   X: blk_src[e] = rank of the id it holds, in place; rides in the launch of the next layer's S (independent work).
// The spec (DESIGN "Sampler spec") and therefore every output bit is unchanged: tests compare with oracle/pgc_oracle.c.
constexpr int kMaxLookback = 1024;     // blocks per S / R launch (every block reads all its predecessors' granules)

// sum of the values of granules [0, b) once each carries `tag`; wave 0 of the block calls it, result in every lane
__device__ __forceinline__ int lookback_sum(const unsigned long long* agg, int b, uint32_t tag, int lane, int32_t* err) {
  int sum = 0;
  for (int j = lane; j < b; j += kWave) {
    unsigned long long g;
    int spins = 0;
    while (true) {
      g = __hip_atomic_load(agg + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((uint32_t)(g >> 32) == tag) break;
      if (++spins > (1 << 22)) {      // bounded (seconds): a lost predecessor must not hang the GPU — and must not go unseen
        atomicAdd(err, 1);            // pg_sampler_status reports it; the sample it belongs to is garbage
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    sum += (int)(uint32_t)g;
  }
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
  return sum;
}

struct SampleGArgs {
  const int64_t* indptr;
  const int32_t* indices;
  const int64_t* dst_ids;      // ids of layer b+1 (the top layer: the caller's seeds)
  const int32_t* n_dev;        // device count of layer b+1, or NULL -> n_imm
  int32_t n_imm;
  int32_t cap_rows;            // the block's indptr is padded with empty rows up to cap_rows
  int32_t k, g_shift, iters;   // group of 1 << g_shift lanes per destination, `iters` destinations per group
  uint32_t layer, epoch, batch, seed_lo, seed_hi;
  unsigned long long* bitmap;
  int32_t* blk_indptr;
  int32_t* blk_src;            // picks in CSR position, as vertex ids until X relabels them
  int32_t* ecnt;               // edges of the block
  unsigned long long* agg;
  uint32_t tag;
  int32_t* err;                // look-back time-outs (device counter)
  // top layer only
  int64_t* top_ids;            // the sampler's copy of the seeds' layer (NULL below the top)
  int32_t* top_cnt;
  int64_t* nm_top;             // fixed-shape layout: node_mapping at the top layer's offset (ids, then -1 up to nm_cap)
  int32_t nm_cap;
  Bnd bnd;                     // PG_BOUNDS: [0] vertices of the graph (a destination id is followed into indptr)
};

__device__ __forceinline__ void sample_body(const SampleGArgs& a, const int blk, int* lds) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int G = 1 << a.g_shift, gl = tid & (G - 1), grp = tid >> a.g_shift;   // lane inside the group, group inside the block
  const int vpi = 256 >> a.g_shift, vpb = vpi * a.iters;
  const int n = a.n_dev ? *a.n_dev : a.n_imm;
  const int k = a.k;
  const int64_t p_base = (int64_t)blk * vpb + grp;
  // pass 1: pick counts -> the block's aggregate
  int64_t beg0 = 0, deg0 = 0, v0 = -1;
  int mine = 0;
  for (int it = 0; it < a.iters; ++it) {
    const int64_t p = p_base + (int64_t)it * vpi;
    int64_t beg = 0, deg = 0, v = -1;
    if (p < n) {
      v = PG_IDX(a.dst_ids[p], a.bnd, 0, PG_K_SX_SAMPLE, 1);
      beg = a.indptr[v];
      deg = a.indptr[v + 1] - beg;
    }
    if (it == 0) { beg0 = beg; deg0 = deg; v0 = v; }
    if (gl == 0) mine += (int)(deg < k ? deg : k);
  }
  int tot;
  (void)block_excl_scan(mine, lds, &tot);
  if (tid == 0)
    __hip_atomic_store(a.agg + blk, ((unsigned long long)a.tag << 32) | (uint32_t)tot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  // pass 2: picks (their loads are in flight while wave 0 collects the predecessors' aggregates)
  int carry = -1;
  for (int it = 0; it < a.iters; ++it) {
    const int64_t p = p_base + (int64_t)it * vpi;
    int64_t beg = beg0, deg = deg0, v = v0;
    if (it > 0) {
      beg = deg = 0; v = -1;
      if (p < n) {
        v = PG_IDX(a.dst_ids[p], a.bnd, 0, PG_K_SX_SAMPLE, 2);
        beg = a.indptr[v];
        deg = a.indptr[v + 1] - beg;
      }
    }
    const int c = (int)(deg < k ? deg : k);
    // Floyd's k-subset inside the group (spec rule 3): draw j = word pair (j & 1) of Philox call j >> 1
    const bool floyd = deg > k;
    uint64_t t = 0;
    if (floyd && gl < k) {
      uint32_t r[4];
      Philox::gen((uint32_t)v, a.epoch, a.batch, (a.layer << 24) | (uint32_t)(gl >> 1), a.seed_lo, a.seed_hi, r);
      const uint64_t r64 = (gl & 1) ? ((uint64_t)r[3] << 32 | r[2]) : ((uint64_t)r[1] << 32 | r[0]);
      t = bounded(r64, (uint64_t)(deg - k + gl) + 1);
    }
    uint64_t sel = floyd ? t : (uint64_t)gl;
    const int gbase = lane & ~(G - 1);
    const unsigned long long gmask = G == 64 ? ~0ull : (((1ull << G) - 1ull) << gbase);
    for (int j = 1; j < k; ++j) {
      const uint64_t tj = __shfl(t, gbase + j);
      const unsigned long long hit = __ballot(floyd && gl < j && sel == tj) & gmask;
      if (floyd && gl == j && hit != 0ull) sel = (uint64_t)(deg - k + j);
    }
    int32_t u = -1;
    if (gl < c) u = a.indices[beg + (int64_t)sel];
    if (carry < 0) {
      // first iteration: the exclusive prefix of this block (block 0 has none)
      if (tid < kWave) {
        const int s = blk > 0 ? lookback_sum(a.agg, blk, a.tag, lane, a.err) : 0;
        if (tid == 0) lds[16] = s;
      }
      __syncthreads();
      carry = lds[16];
    }
    int itot;
    const int ex = block_excl_scan(gl == 0 ? c : 0, lds, &itot);
    const int pos = __shfl(carry + ex, gbase);
    if (gl == 0 && p <= a.cap_rows) a.blk_indptr[p] = pos;        // rows past n are empty: pos = the block total
    if (gl == 0 && p == a.cap_rows) *a.ecnt = pos;
    if (u >= 0) {
      a.blk_src[pos + gl] = u;
      atomicOr(&a.bitmap[u >> 6], 1ull << (u & 63));
    }
    if (a.top_ids && gl == 0) {
      if (p < n) a.top_ids[p] = v;
      if (a.nm_top && p < a.nm_cap) a.nm_top[p] = p < n ? v : -1;
    }
    carry += itot;
  }
  if (a.top_cnt && blk == 0 && tid == 0) *a.top_cnt = n;
}

struct RelabelXArgs {
  int32_t* blk_src;
  const int32_t* ecnt;
  const unsigned long long* bm;
  const uint32_t* word_rank;
  int64_t* nm_tail;            // fixed-shape layout: node_mapping at layer b's offset; [count, cap) := -1
  const int32_t* lcnt_b;
  int32_t cap_b;
  // the chain's last launch also publishes the sizes (fixed-shape layout; the DGL layout runs k_pack)
  int32_t final_sizes;
  int32_t num_layers;
  const int32_t* lcnt;         // [num_layers]
  const int32_t* ecnt_all;     // [num_layers - 1]
  int32_t* layer_offsets;
  int32_t* sizes_pinned;
  int32_t* sizes_dev;
  int32_t pad_off[PG_MAX_LAYERS + 1];
  Bnd bnd;                     // PG_BOUNDS: [0] vertices of the graph (blk_src holds vertex ids until relabelled), [1] edge capacity
};

__device__ __forceinline__ void relabel_body(const RelabelXArgs& a, const int blk, const int nblk) {
  const int64_t tid = (int64_t)blk * blockDim.x + threadIdx.x, nth = (int64_t)nblk * blockDim.x;
  const int nnz = (int)PG_IDX((long long)*a.ecnt, a.bnd, 1, PG_K_SX_RELABEL, 2);
  for (int64_t e = tid; e < nnz; e += nth) {
    const int32_t u = PG_IDX(a.blk_src[e], a.bnd, 0, PG_K_SX_RELABEL, 1);
    const int64_t w = u >> 6;
    const unsigned long long below = a.bm[w] & ((1ull << (u & 63)) - 1ull);
    a.blk_src[e] = (int32_t)(a.word_rank[w] + __popcll(below));
  }
  if (a.nm_tail) {
    const int c = *a.lcnt_b;
    for (int64_t i = c + tid; i < a.cap_b; i += nth) a.nm_tail[i] = -1;
  }
  if (a.final_sizes && blk == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int l = 0; l < PG_MAX_LAYERS; ++l) {
      if (l <= a.num_layers) a.layer_offsets[l] = a.pad_off[l];
      if (l < a.num_layers) {
        const int c = a.lcnt[l];
        a.sizes_pinned[l] = c;
        if (a.sizes_dev) a.sizes_dev[l] = c;
      }
      if (l + 1 < a.num_layers) {
        const int e = a.ecnt_all[l];
        a.sizes_pinned[PG_MAX_LAYERS + l] = e;
        if (a.sizes_dev) a.sizes_dev[PG_MAX_LAYERS + l] = e;
      }
    }
    __threadfence_system();
  }
}

// blocks [0, s_blocks): S of one layer; the rest: X of the layer above it (either side may be empty)
__global__ __launch_bounds__(256) void k_sx(const SampleGArgs s, const RelabelXArgs x, const int s_blocks) {
  __shared__ int lds[20];
  if ((int)blockIdx.x < s_blocks) sample_body(s, (int)blockIdx.x, lds);
  else relabel_body(x, (int)blockIdx.x - s_blocks, (int)gridDim.x - s_blocks);
}

struct RankArgs {
  const unsigned long long* bm;
  unsigned long long* other_bm;   // zeroed here: the next S marks into it
  // how: every word (a V/8-byte store per layer: 1 MB at 8.5 M vertices), or — clear_ids set — only the words of the ids
  // that were marked there, i.e. the layer the PREVIOUS rank launch emitted (<= a few 10^4 scattered stores whatever V is)
  const int64_t* clear_ids;
  const int32_t* clear_cnt;
  int64_t clear_cap;
  int64_t n_words;
  int32_t m;                      // the block covers 1024 * m words, 4 per thread and round
  unsigned long long* agg;
  uint32_t tag;
  int32_t* err;
  int64_t* out_ids;
  int64_t cap;
  uint32_t* word_rank;
  int32_t* count_out;
  int64_t* nm_out;                // fixed-shape layout: node_mapping at this layer's offset (NULL otherwise)
  Bnd bnd;                        // PG_BOUNDS: [0] vertices of the graph (an id of the clear list is followed into the bitmap)
};

__global__ __launch_bounds__(256) void k_bm_rank(const RankArgs a) {
  __shared__ int lds[20];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), blk = blockIdx.x;
  const int64_t wblk = (int64_t)blk * kWordsPerBlock * a.m;
  unsigned long long w[4] = {0ull, 0ull, 0ull, 0ull};
  int mine = 0;
  const bool clear_all = a.clear_ids == nullptr;
  if (!clear_all) {
    int64_t n = *a.clear_cnt;
    if (n > a.clear_cap) n = a.clear_cap;
    for (int64_t i = (int64_t)blk * blockDim.x + tid; i < n; i += (int64_t)gridDim.x * blockDim.x)
      a.other_bm[PG_IDX(a.clear_ids[i], a.bnd, 0, PG_K_BM_RANK, 1) >> 6] = 0ull;
  }
  for (int r = 0; r < a.m; ++r) {
    const int64_t w0 = wblk + (int64_t)r * kWordsPerBlock + tid * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      w[i] = (w0 + i < a.n_words) ? a.bm[w0 + i] : 0ull;
      mine += __popcll(w[i]);
      if (clear_all && w0 + i < a.n_words) a.other_bm[w0 + i] = 0ull;
    }
  }
  int tot;
  (void)block_excl_scan(mine, lds, &tot);
  if (tid == 0)
    __hip_atomic_store(a.agg + blk, ((unsigned long long)a.tag << 32) | (uint32_t)tot, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
  if (tid < kWave) {
    const int s = blk > 0 ? lookback_sum(a.agg, blk, a.tag, lane, a.err) : 0;
    if (tid == 0) lds[16] = s;
  }
  __syncthreads();
  int carry = lds[16];
  if (blk == (int)gridDim.x - 1 && tid == 0) *a.count_out = carry + tot;
  for (int r = 0; r < a.m; ++r) {
    const int64_t w0 = wblk + (int64_t)r * kWordsPerBlock + tid * 4;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (a.m > 1) w[i] = (w0 + i < a.n_words) ? a.bm[w0 + i] : 0ull;   // one round: still in registers
      c += __popcll(w[i]);
    }
    int rtot;
    int pos = carry + block_excl_scan(c, lds, &rtot);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (w0 + i < a.n_words) {
        a.word_rank[w0 + i] = (uint32_t)pos;
        unsigned long long x = w[i];
        const int64_t vbase = (w0 + i) << 6;
        while (x) {
          const int b = __ffsll((long long)x) - 1;
          x &= x - 1;
          if (pos < a.cap) {
            a.out_ids[pos] = vbase + b;
            if (a.nm_out) a.nm_out[pos] = vbase + b;
          }
          ++pos;
        }
      }
    }
    carry += rtot;
  }
}

// ---- block transposes (source-major copy of a block, for the gather-form backward aggregation) -------
// One (key = source, value = destination row) pair per edge, edges in destination order; a STABLE radix sort
// by key (rocPRIM) then leaves every source's destinations ascending, which fixes the summation order of
// the backward pass. Entries past the block's real edge count get the key `pad_key` (> every source).
// tcnt[s] += #edges whose source is s (zero on entry; the scan that follows clears it again).
__global__ __launch_bounds__(256) void k_t_keys(const int32_t* __restrict__ indptr, const int32_t* __restrict__ src,
                                                const int32_t* __restrict__ n_dst_dev,
                                                const int32_t* __restrict__ nnz_dev, int32_t cap_edges,
                                                int32_t pad_key, int32_t* __restrict__ key,
                                                int32_t* __restrict__ val, int32_t* __restrict__ tcnt,
                                                int32_t* __restrict__ heavy, Bnd bnd) {
  // PG_BOUNDS: [0] destination rows + 1, [1] edge capacity + 1, [2] source rows (a block edge is followed into tcnt)
  const int n = (int)PG_IDX((long long)*n_dst_dev, bnd, 0, PG_K_T_KEYS, 1);
  const int nnz = (int)PG_IDX((long long)*nnz_dev, bnd, 1, PG_K_T_KEYS, 2);
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  for (int v = tid; v < n; v += nth) {
    for (int e = indptr[v]; e < indptr[v + 1]; ++e) {
      const int sr = PG_IDX(src[e], bnd, 2, PG_K_T_KEYS, 3);
      key[e] = sr;
      val[e] = v;
      atomicAdd(tcnt + sr, 1);
    }
  }
  for (int e = nnz + tid; e < cap_edges; e += nth) {
    key[e] = pad_key;
    val[e] = 0;
  }
  if (tid == 0 && heavy) heavy[0] = 0;
}

// sources with more than PG_HEAVY_ROW edges (hubs; in the benchmark graph: the vertex every isolated train
// vertex aliases to, utils.py:34) are listed so that the backward pass can give each of them a whole block
__global__ __launch_bounds__(256) void k_t_heavy(const int32_t* __restrict__ tptr, const int32_t* __restrict__ n_src_dev,
                                                 int32_t* __restrict__ heavy, int32_t heavy_cap) {
  const int n = *n_src_dev;
  for (int sr = blockIdx.x * blockDim.x + threadIdx.x; sr < n; sr += gridDim.x * blockDim.x) {
    if (tptr[sr + 1] - tptr[sr] > PG_HEAVY_ROW) {
      const int i = atomicAdd(heavy, 1);
      if (i < heavy_cap) heavy[1 + i] = sr;
    }
  }
}

// The same transpose in ONE workgroup, for blocks of at most 1024 * ITEMS edges and source rows (the 12 000-edge seeds'
// block of the benchmark's step): no global atomics (the hub's counter was one contended address), no multi-pass device sort
// (seven small launches running beside the compute stream's HBM-bound kernel), one CU.
// Round 6: a counting sort by source, entirely in LDS (145 KB of the CU's 160). Rounds 2-5 packed (source, destination) keys and
// sorted them stably with rocprim::block_radix_sort (37.8 us, the longest kernel of every step); a hand-written stable LSD sort
// (two 7-bit passes, ranks from seven-ballot match masks) measured 34 us — phase stamps showed the passes VALU-bound: one CU
// issues 64 lanes per clock, and 14 ballots + mask updates per key are ~55 vector instructions x 12 288 keys. A (source,
// destination) pair is unique, so stability is not needed if every source's destinations are simply put in ascending order:
//   1. destination of every edge: the thread of destination v writes v over its edges [indptr[v], indptr[v + 1])   (LDS array B)
//   2. histogram of the sources (returning LDS adds: the edge's ARRIVAL NUMBER inside its source), exclusive scan = tptr
//      (written out coalesced, no barrier behind it); sources with more than PG_HEAVY_ROW edges are the backward pass's hub list
//   3. placement in ANY order: start of the source + arrival number; a source's ONLY edge goes straight to C     (B -> B | C)
//   4. every other edge counts the destinations of its source that are smaller than its own = its rank         (B -> C)
//      (sum over sources of count^2 reads: sources of a fan-out-2 block have one or two edges)
//   5. hubs: their destinations set bits in a bitmap; the set bits in order are the sorted list                 (B -> C)
//   6. tdst = C, coalesced.
#ifdef PG_T_STAMPS      // tools/exp_t_stamps.sh: where one launch of k_t_block spends its time (100 MHz wall clock, thread 0)
__device__ long long g_t_stamps[16];
#define T_STAMP(k) do { if (threadIdx.x == 0) g_t_stamps[k] = (long long)wall_clock64(); } while (0)
#else
#define T_STAMP(k) do { } while (0)
#endif
template <int ITEMS>
__global__ __launch_bounds__(1024) void k_t_block(const int32_t* __restrict__ indptr, const int32_t* __restrict__ src,
                                                  const int32_t* __restrict__ n_dst_dev,
                                                  const int32_t* __restrict__ nnz_dev, int32_t cap_edges,
                                                  int32_t cap_rows, int32_t* __restrict__ tptr,
                                                  int32_t* __restrict__ tdst, int32_t* __restrict__ heavy,
                                                  int32_t heavy_cap, Bnd bnd) {
  // PG_BOUNDS: [0] destination rows + 1, [1] edge capacity + 1, [2] source rows (a block edge becomes an LDS index)
  constexpr int kN = 1024 * ITEMS;
  constexpr int kWords = kN / 32;
  constexpr int kHubs = kN / (PG_HEAVY_ROW + 1) + 1;       // more sources than this cannot have > PG_HEAVY_ROW edges each
  // A is indexed by source id both at random (adds) and ITEMS-in-a-row per thread (scan): one word of padding per 32 keeps the
  // row-wise accesses of a wave off the same banks
#define T_PAD(i) ((i) + ((i) >> 5))
  __shared__ int32_t A[T_PAD(kN) + 2];            // per source: edge count -> exclusive start -> (after placement) end
  __shared__ int32_t B[kN];                       // per edge: its destination; then the destinations grouped by source
  __shared__ int32_t C[kN];                       // the destinations grouped by source, ascending inside a source = tdst
  __shared__ uint32_t bits[kWords];
  __shared__ int32_t wave_sum[16];
  __shared__ int32_t hub_start[kHubs], hub_count[kHubs];
  __shared__ int32_t n_hubs;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
  const int n = (int)PG_IDX((long long)*n_dst_dev, bnd, 0, PG_K_T_BLOCK, 1);
  const int nnz = (int)PG_IDX((long long)*nnz_dev, bnd, 1, PG_K_T_BLOCK, 2);
  T_STAMP(0);
  if (tid == 0) {
    if (heavy) heavy[0] = 0;
    n_hubs = 0;
  }
  // one value per thread -> its exclusive prefix over the workgroup (contains barriers: called by every thread)
  auto block_exclusive = [&](int32_t x) -> int32_t {
    int32_t incl = x;
#pragma unroll
    for (int dlt = 1; dlt < kWave; dlt <<= 1) {
      const int32_t t = __shfl_up(incl, dlt);
      if (lane >= dlt) incl += t;
    }
    __syncthreads();                              // (wave_sum may still be read by the previous call)
    if (lane == kWave - 1) wave_sum[w] = incl;
    __syncthreads();
    int32_t base = 0;
    for (int ww = 0; ww < w; ++ww) base += wave_sum[ww];
    return base + incl - x;
  };
  // ---- loads: edge e = tid + 1024 j (coalesced), destination v = tid + 1024 j; all in flight together ------------------
  int32_t sr[ITEMS], ip0[ITEMS], ip1[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int e = tid + j * 1024;
    sr[j] = e < nnz ? (int32_t)PG_IDX(src[e], bnd, 2, PG_K_T_BLOCK, 3) : -1;
    ip0[j] = e < n ? indptr[e] : 0;
    ip1[j] = e < n ? indptr[e + 1] : 0;
  }
  for (int i = tid; i < T_PAD(kN) + 2; i += 1024) A[i] = 0;
  __syncthreads();
  T_STAMP(1);
  // ---- 1 + 2: destinations per edge, histogram of the sources -------------------------------------------------------------
  int32_t arr[ITEMS];                             // the edge's arrival number among the edges of its source
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int v = tid + j * 1024;
    for (int e = ip0[j]; e < ip1[j]; ++e) B[e] = v;
    arr[j] = sr[j] >= 0 ? atomicAdd(&A[T_PAD(sr[j])], 1) : 0;
  }
  __syncthreads();
  T_STAMP(2);
  int32_t d[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    const int e = tid + j * 1024;
    d[j] = e < nnz ? B[e] : 0;
  }
  // exclusive scan of the counts: thread t owns sources [t * ITEMS, (t + 1) * ITEMS)
  int32_t cnt[ITEMS], start[ITEMS];
  int32_t sum = 0;
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    cnt[j] = A[T_PAD(tid * ITEMS + j)];
    start[j] = sum;
    sum += cnt[j];
  }
  const int32_t base = block_exclusive(sum);      // (its barriers also separate the reads of B above from the writes below)
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    start[j] += base;
    A[T_PAD(tid * ITEMS + j)] = start[j];
    if (cnt[j] > PG_HEAVY_ROW) {
      const int h = atomicAdd(&n_hubs, 1);
      hub_start[h] = start[j];
      hub_count[h] = cnt[j];
      if (heavy) {
        const int i = atomicAdd(heavy, 1);
        if (i < heavy_cap) heavy[1 + i] = tid * ITEMS + j;
      }
    }
  }
  if (tid == 1023) A[T_PAD(kN)] = base + sum;      // = nnz
  __syncthreads();
  T_STAMP(3);
  // tptr out, coalesced (sources past the last real one: nnz). A keeps the starts from here on: no barrier behind this pass,
  // the stores drain under the LDS work below. (Written from the scan's registers instead — 12 words per thread — the stores
  // are 48 bytes apart inside an instruction: 5.9 us for the scan phase instead of 1.9.)
  for (int i = tid; i <= cap_rows; i += 1024) tptr[i] = A[T_PAD(i)];
  T_STAMP(4);
  // ---- 3: placement: start of the source + arrival number (any order inside a source). A source's only edge is final. ------
  int32_t beg[ITEMS], num[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    if (sr[j] < 0) continue;
    beg[j] = A[T_PAD(sr[j])];
    num[j] = A[T_PAD(sr[j] + 1)] - beg[j];
    if (num[j] == 1) C[beg[j]] = d[j];
    else B[beg[j] + arr[j]] = d[j];
  }
  __syncthreads();
  T_STAMP(5);
  // ---- 4: rank inside the source -------------------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < ITEMS; ++j) {
    if (sr[j] < 0 || num[j] == 1 || num[j] > PG_HEAVY_ROW) continue;       // (hubs: below)
    int32_t rank = 0;
    for (int32_t q = beg[j]; q < beg[j] + num[j]; ++q) rank += B[q] < d[j] ? 1 : 0;
    C[beg[j] + rank] = d[j];
  }
  T_STAMP(6);
  // ---- 5: hubs — set bits, then the set bits in order ---------------------------------------------------------------------
  const int nh = n_hubs;                          // (written before the barriers above)
  for (int h = 0; h < nh; ++h) {
    const int32_t beg = hub_start[h], c = hub_count[h];
    for (int i = tid; i < kWords; i += 1024) bits[i] = 0u;
    __syncthreads();
    for (int i = tid; i < c; i += 1024) {
      const int32_t v = B[beg + i];
      atomicOr(&bits[v >> 5], 1u << (v & 31));
    }
    __syncthreads();
    uint32_t word = tid < kWords ? bits[tid] : 0u;
    int32_t at = beg + block_exclusive(__popc(word));
    for (; word; word &= word - 1u) C[at++] = tid * 32 + (__ffs((int)word) - 1);
  }
  __syncthreads();
  T_STAMP(7);
  // ---- 6: out -------------------------------------------------------------------------------------------------------------
  for (int i = tid; i < cap_edges; i += 1024) tdst[i] = i < nnz ? C[i] : 0;
  T_STAMP(8);
#undef T_PAD
}

// copy seeds into the top layer buffer + set its count; publishes the call's parameters on the device
__global__ void k_seed_layer(const SampleParams p, SampleParams* __restrict__ prm_dev, int64_t* layer_ids,
                             int32_t* layer_cnt) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.n_seeds; i += gridDim.x * blockDim.x)
    layer_ids[i] = p.seeds[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *layer_cnt = p.n_seeds;
    *prm_dev = p;
  }
}

__global__ __launch_bounds__(256) void k_mark_neighbors(const int64_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices,
                                                        const int64_t* __restrict__ frontier, int64_t n,
                                                        unsigned long long* __restrict__ bm, int mark_self) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
  for (int64_t p = wave0; p < n; p += nwaves) {
    const int64_t v = frontier[p];
    const int64_t beg = indptr[v], end = indptr[v + 1];
    for (int64_t e = beg + lane; e < end; e += kWave) {
      const int32_t u = indices[e];
      atomicOr(&bm[u >> 6], 1ull << (u & 63));
    }
    if (mark_self && lane == 0) atomicOr(&bm[v >> 6], 1ull << (v & 63));
  }
}

static inline int grid_for(int64_t n, int per_block, int cap = 8192) {
  int64_t g = ceil_div<int64_t>(n, per_block);
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

}  // namespace pg

using namespace pg;

struct pg_sampler {
  struct SlotState {
    const void* key = nullptr;
    hipStream_t stream = nullptr;
    pg_nodeflow_desc_t desc{};
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int64_t calls = 0;
    bool graph_failed = false;
  };
  std::vector<SlotState*> tslot_states;   // pg_sampler_transpose's per-(slot, stream) graphs (multi-launch device sort only)
  SampleParams* prm_d = nullptr;     // device copy of the running call's parameters (wide fan-out chain)
  unsigned long long* agg = nullptr; // look-back granules {value, tag} of the running S / R launch
  int32_t* err = nullptr;            // device: look-back polls that gave up (pg_sampler_status)
  uint32_t tag = 0;                  // last tag handed to a launch (never 0)
  uint64_t rank_launches = 0;        // picks of launch i are marked in bitmap (i & 1); its R zeroes the other one
  int max_lookback = kMaxLookback;   // blocks per S / R launch
  int rank_m = 1, rank_blocks = 1;   // k_bm_rank: rounds per block, blocks (<= max_lookback)
  bool clear_by_ids = false;         // k_bm_rank clears the other bitmap through the previous layer's id list (large V)
  int64_t V = 0;
  const int64_t* indptr = nullptr;
  const int32_t* indices = nullptr;
  int32_t B = 0, k = 0, hops = 0;
  int64_t cap[PG_MAX_LAYERS] = {0};  // per-layer vertex capacity, layer 0 first
  int64_t n_words = 0;
  int n_bm_blocks = 0;
  // device buffers
  unsigned long long* bitmap = nullptr;
  unsigned long long* bitmap_b = nullptr;   // layers alternate between the two (the other one is being cleared)
  uint32_t* word_rank = nullptr;
  int32_t* partial = nullptr;
  int64_t* layer_ids[PG_MAX_LAYERS] = {nullptr};
  int32_t* counters = nullptr;  // [0..L] layer counts, [PG_MAX_LAYERS..] block edge counts
  int32_t* nbr = nullptr;       // ELL picks of the current block (reused)
  int32_t* cnt = nullptr;
  int32_t* tcnt = nullptr;      // per-source edge counts while a block is transposed (zero between uses)
  int32_t* tdummy = nullptr;    // sink for the scan's total when the transposes run outside the sampling chain
  int32_t* tkey = nullptr;      // [2][cap edges] sort keys in / out, then [cap edges] values in
  void* tsort_tmp = nullptr;    // rocPRIM radix sort scratch
  size_t tsort_bytes = 0;
  int64_t max_edges = 0;
};

static void sampler_free(pg_sampler* s) {
  if (!s) return;
  (void)hipFree(s->bitmap);
  (void)hipFree(s->bitmap_b);
  (void)hipFree(s->word_rank);
  (void)hipFree(s->partial);
  for (auto p : s->layer_ids) (void)hipFree(p);
  (void)hipFree(s->counters);
  (void)hipFree(s->nbr);
  (void)hipFree(s->cnt);
  (void)hipFree(s->tcnt);
  (void)hipFree(s->tdummy);
  (void)hipFree(s->tkey);
  (void)hipFree(s->tsort_tmp);
  (void)hipFree(s->prm_d);
  for (auto* c : s->tslot_states) {
    if (c->exec) (void)hipGraphExecDestroy(c->exec);
    if (c->graph) (void)hipGraphDestroy(c->graph);
    delete c;
  }
  (void)hipFree(s->agg);
  (void)hipFree(s->err);
  delete s;
}

extern "C" {

int pg_sampler_create(int64_t V, const int64_t* indptr, const int32_t* indices, int32_t max_seeds,
                      int32_t fanout, int32_t num_hops, pg_sampler_t** out) {
  if (!out || V <= 0 || V > INT32_MAX || !indptr || !indices || max_seeds <= 0 || fanout <= 0 ||
      num_hops <= 0 || num_hops + 1 > PG_MAX_LAYERS)
    return PG_ERR_INVALID;
  // fan-out <= 64: lane groups (k_sx); above: the picks of a vertex live in an LDS strip (k_sample_wide),
  // four strips of 4 * fanout bytes per block
  if (fanout > kMaxFanout) return PG_ERR_UNSUPPORTED;
  pg_sampler* s = new (std::nothrow) pg_sampler;
  if (!s) return PG_ERR_NOMEM;
  s->V = V; s->indptr = indptr; s->indices = indices;
  s->B = max_seeds; s->k = fanout; s->hops = num_hops;
  const int L = num_hops;  // top layer index
  s->cap[L] = max_seeds;
  for (int l = L - 1; l >= 0; --l) {
    const int64_t c = s->cap[l + 1] * fanout;
    s->cap[l] = c < V ? c : V;
  }
  s->n_words = ceil_div<int64_t>(V, 64);
  s->n_bm_blocks = (int)ceil_div<int64_t>(s->n_words, kWordsPerBlock);
  int64_t max_dst = 0;
  for (int l = 1; l <= L; ++l) max_dst = s->cap[l] > max_dst ? s->cap[l] : max_dst;
  // a NodeFlow block is per-minibatch data with 32-bit edge offsets (blk_indptr / blk_tptr); the graph's own CSC offsets
  // are 64-bit everywhere. Refuse a (batch, fan-out) whose worst-case block would not fit instead of overflowing.
  if (max_dst * fanout >= INT32_MAX) {
    delete s;
    return PG_ERR_OVERFLOW;
  }
  bool ok = true;
  ok &= hipMalloc(&s->bitmap, s->n_words * 8) == hipSuccess;
  ok &= hipMalloc(&s->bitmap_b, s->n_words * 8) == hipSuccess;
  ok &= hipMalloc(&s->word_rank, s->n_words * 4) == hipSuccess;
  ok &= hipMalloc(&s->partial, (size_t)s->n_bm_blocks * 4 + 4) == hipSuccess;
  for (int l = 0; l <= L; ++l) ok &= hipMalloc(&s->layer_ids[l], s->cap[l] * 8) == hipSuccess;
  ok &= hipMalloc(&s->counters, 2 * PG_MAX_LAYERS * 4) == hipSuccess;
  if (fanout > kWave) {   // ELL picks + counts: the wide chain only
    ok &= hipMalloc(&s->nbr, max_dst * fanout * 4) == hipSuccess;
    ok &= hipMalloc(&s->cnt, max_dst * 4) == hipSuccess;
  }
  // (test hook: PG_SAMPLER_LOOKBACK=<n> shrinks the block limit so that small graphs take the multi-round paths)
  if (const char* e = getenv("PG_SAMPLER_LOOKBACK")) {
    const int v = atoi(e);
    if (v >= 1 && v <= kMaxLookback) s->max_lookback = v;
  }
  // The rank launch zeroes the bitmap the next S marks into. Every word of it — a V/8-byte store per layer, fine at 8.5 M
  // vertices (1 MB) — or, from PG_SAMPLER_CLEAR_IDS_ABOVE bitmap words on (default 2^20 words = 64 M vertices; 0 = always),
  // only the words of the ids that were marked there: at 10^9 vertices the full clear alone is 125 MB of stores per layer,
  // as much HBM traffic as a whole training step, on a stream that runs beside the step's HBM-bound kernel.
  // The id list is the layer the previous rank launch emitted; with one hop that is the buffer this launch rewrites, so a
  // one-hop sampler keeps the full clear.
  {
    int64_t above = (int64_t)1 << 20;
    if (const char* e = getenv("PG_SAMPLER_CLEAR_IDS_ABOVE")) above = atoll(e);
    s->clear_by_ids = s->hops >= 2 && s->n_words > above;
  }
  s->rank_m = (int)ceil_div<int64_t>(s->n_words, (int64_t)kWordsPerBlock * s->max_lookback);
  if (s->rank_m < 1) s->rank_m = 1;
  s->rank_blocks = (int)ceil_div<int64_t>(s->n_words, (int64_t)kWordsPerBlock * s->rank_m);
  ok &= hipMalloc(&s->agg, (size_t)kMaxLookback * 8) == hipSuccess;
  if (ok) ok &= hipMemset(s->agg, 0, (size_t)kMaxLookback * 8) == hipSuccess;
  ok &= hipMalloc(&s->err, 4) == hipSuccess;
  if (ok) ok &= hipMemset(s->err, 0, 4) == hipSuccess;
  ok &= hipMalloc(&s->prm_d, sizeof(SampleParams)) == hipSuccess;
  ok &= hipMalloc(&s->tcnt, (s->cap[0] + 1) * 4) == hipSuccess;
  ok &= hipMalloc(&s->tdummy, 4) == hipSuccess;
  if (ok) ok &= hipMemset(s->bitmap, 0, s->n_words * 8) == hipSuccess;
  if (ok) ok &= hipMemset(s->bitmap_b, 0, s->n_words * 8) == hipSuccess;
  if (ok) ok &= hipMemset(s->counters, 0, 2 * PG_MAX_LAYERS * 4) == hipSuccess;
  if (ok) ok &= hipMemset(s->tcnt, 0, (s->cap[0] + 1) * 4) == hipSuccess;
  s->max_edges = max_dst * fanout;
  ok &= hipMalloc(&s->tkey, (size_t)s->max_edges * 3 * 4) == hipSuccess;
  if (ok) {
    int32_t* kk = s->tkey;
    ok &= rocprim::radix_sort_pairs(nullptr, s->tsort_bytes, kk, kk, kk, kk, (size_t)s->max_edges, 0, 32,
                                    (hipStream_t) nullptr) == hipSuccess;
    if (ok) ok &= hipMalloc(&s->tsort_tmp, s->tsort_bytes ? s->tsort_bytes : 4) == hipSuccess;
  }
  if (!ok) {
    sampler_free(s);
    return PG_ERR_NOMEM;
  }
  *out = s;
  return PG_OK;
}

int pg_sampler_destroy(pg_sampler_t* s) {
  sampler_free(s);
  return PG_OK;
}

int pg_sampler_capacity(const pg_sampler_t* s, int64_t* cap_nodes, int64_t* cap_blk_rows, int64_t* cap_blk_edges) {
  if (!s) return PG_ERR_INVALID;
  int64_t tot = 0;
  for (int l = 0; l <= s->hops; ++l) tot += s->cap[l];
  if (cap_nodes) *cap_nodes = tot;
  for (int b = 0; b < s->hops; ++b) {
    if (cap_blk_rows) cap_blk_rows[b] = s->cap[b + 1];
    if (cap_blk_edges) cap_blk_edges[b] = s->cap[b + 1] * s->k;
  }
  return PG_OK;
}

// does block b take the one-workgroup transpose (k_t_block)?
static bool transpose_one_workgroup(const pg_sampler* s, int b) {
  const int64_t cap_edges = s->cap[b + 1] * s->k;
  const int64_t larger = cap_edges > s->cap[b] + 1 ? cap_edges : s->cap[b] + 1;
  if (larger > 1024 * 12) return false;
  int vb = 1, kb = 1;
  while ((1ll << vb) < s->cap[b + 1]) ++vb;
  while ((1ll << kb) <= s->cap[b]) ++kb;
  return vb + kb <= 32;
}

// source-major copy of block b of the slot `o`. n_dst / n_src / nnz: DEVICE counters of that NodeFlow (the
// sampler's own while it is still sampling the slot, the slot's sizes_dev copy afterwards).
static int transpose_block(pg_sampler* s, const pg_nodeflow_desc_t* o, int b, const int32_t* n_dst, const int32_t* n_src,
                           const int32_t* nnz, int32_t* scan_total_dummy, hipStream_t ax) {
  const int32_t* indptr_b = o->blk_indptr + o->blk_indptr_off[b];
  const int32_t* src_b = o->blk_src + o->blk_src_off[b];
  int32_t* tptr_b = o->blk_tptr + o->blk_tptr_off[b];
  int32_t* tdst_b = o->blk_tdst + o->blk_src_off[b];
  int32_t* heavy_b = o->blk_theavy ? o->blk_theavy + o->blk_theavy_off[b] : nullptr;
  const int32_t cap_edges = (int32_t)(s->cap[b + 1] * s->k);
  const int32_t pad_key = (int32_t)s->cap[b];
  int32_t *key_in = s->tkey, *key_out = s->tkey + s->max_edges, *val_in = s->tkey + 2 * s->max_edges;
  const Bnd tb = bnd(s->cap[b + 1] + 1, (long long)cap_edges + 1, s->cap[b]);
  // small blocks (the seeds' block of a 2-layer step: 12 000 edges): one workgroup does it all
  const int64_t larger = cap_edges > s->cap[b] + 1 ? cap_edges : s->cap[b] + 1;
  if (larger <= 1024 * 12) {
    int vb = 1, kb = 1;
    while ((1ll << vb) < s->cap[b + 1]) ++vb;
    while ((1ll << kb) <= s->cap[b]) ++kb;
    if (vb + kb <= 32) {
      const int32_t hcap = (int32_t)(cap_edges / PG_HEAVY_ROW);
      if (larger <= 1024 * 4)
        hipLaunchKernelGGL(k_t_block<4>, dim3(1), dim3(1024), 0, ax, indptr_b, src_b, n_dst, nnz, cap_edges,
                           (int32_t)s->cap[b], tptr_b, tdst_b, heavy_b, hcap, tb);
      else
        hipLaunchKernelGGL(k_t_block<12>, dim3(1), dim3(1024), 0, ax, indptr_b, src_b, n_dst, nnz, cap_edges,
                           (int32_t)s->cap[b], tptr_b, tdst_b, heavy_b, hcap, tb);
      PG_LAUNCH_CHECK();
      return PG_OK;
    }
  }
  hipLaunchKernelGGL(k_t_keys, dim3(grid_for(s->cap[b + 1], 256, 1024)), dim3(256), 0, ax, indptr_b, src_b, n_dst, nnz,
                     cap_edges, pad_key, key_in, val_in, s->tcnt, heavy_b, tb);
  PG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_scan_cnt<true>, dim3(1), dim3(kScanThreads), 0, ax, s->tcnt, n_src, tptr_b, scan_total_dummy,
                     (int32_t)s->cap[b]);
  PG_LAUNCH_CHECK();
  int bits = 1;
  while ((1ll << bits) <= pad_key) ++bits;
  size_t bytes = s->tsort_bytes;
  PG_HIP(rocprim::radix_sort_pairs(s->tsort_tmp, bytes, key_in, key_out, val_in, tdst_b, (size_t)cap_edges, 0,
                                   (unsigned)bits, ax));
  if (heavy_b) {
    hipLaunchKernelGGL(k_t_heavy, dim3(grid_for(s->cap[b], 256, 1024)), dim3(256), 0, ax, tptr_b, n_src, heavy_b,
                       (int32_t)(cap_edges / PG_HEAVY_ROW));
    PG_LAUNCH_CHECK();
  }
  return PG_OK;
}

// fan-out above one wave: the round-1 chain (k_sample_wide -> count -> scans -> emit -> relabel per layer, k_pack), plain
// launches; the call's scalars reach the kernels through a device copy that k_seed_layer makes of its BY-VALUE argument
static int enqueue_chain_wide(pg_sampler* s, const pg_nodeflow_desc_t* o, hipStream_t st, const SampleParams& prm) {
  const int L = s->hops;
  int32_t* lcnt = s->counters;
  int32_t* ecnt = s->counters + PG_MAX_LAYERS;

  hipLaunchKernelGGL(k_seed_layer, dim3(grid_for(s->B, 256, 64)), dim3(256), 0, st, prm, s->prm_d, s->layer_ids[L],
                     lcnt + L);
  PG_LAUNCH_CHECK();
  for (int b = L - 1; b >= 0; --b) {
    const int64_t cap_dst = s->cap[b + 1];
    unsigned long long* bm = (s->rank_launches & 1) ? s->bitmap_b : s->bitmap;
    unsigned long long* other = (s->rank_launches & 1) ? s->bitmap : s->bitmap_b;
    ++s->rank_launches;
    SampleArgs a{};
    a.indptr = s->indptr; a.indices = s->indices;
    a.dst_ids = s->layer_ids[b + 1]; a.n_dst = lcnt + b + 1;
    a.nbr = s->nbr; a.cnt = s->cnt; a.bitmap = bm; a.k = s->k;
    a.prm = s->prm_d; a.layer = (uint32_t)b;
    hipLaunchKernelGGL(k_sample_wide, dim3(grid_for(cap_dst, 4)), dim3(256), (size_t)4 * s->k * sizeof(uint32_t), st, a);
    PG_LAUNCH_CHECK();
    int32_t* indptr_b = o->blk_indptr + o->blk_indptr_off[b];
    int32_t* src_b = o->blk_src + o->blk_src_off[b];
    // (+ the other bitmap back to zero: the next layer — or the next call — marks into it)
    hipLaunchKernelGGL(k_bm_count, dim3(s->n_bm_blocks), dim3(256), 0, st, bm, s->n_words, s->partial, other);
    PG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_scan2, dim3(1), dim3(kScanThreads), 0, st, s->cnt, lcnt + b + 1, indptr_b, ecnt + b,
                       (int32_t)cap_dst, s->partial, s->n_bm_blocks, lcnt + b);
    PG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_bm_emit, dim3(s->n_bm_blocks), dim3(256), 0, st, bm, s->n_words, s->partial,
                       s->layer_ids[b], s->cap[b], s->word_rank);
    PG_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_relabel, dim3(grid_for(cap_dst * s->k, 256)), dim3(256), 0, st, s->nbr, s->cnt,
                       lcnt + b + 1, s->k, indptr_b, bm, s->word_rank, src_b);
    PG_LAUNCH_CHECK();
    // source-major copy of this block (gather-form backward aggregation) in line, unless the caller asked to
    // run it later on a stream of its own choice (pg_sampler_transpose)
    if (!o->defer_transpose && o->blk_tptr && o->blk_tdst && ((o->transpose_mask >> b) & 1u)) {
      const int rc = transpose_block(s, o, b, lcnt + b + 1, lcnt + b, ecnt + b, s->counters + 2 * PG_MAX_LAYERS - 1, st);
      if (rc != PG_OK) return rc;
    }
  }
  return PG_OK;
}

// node_mapping / layer_offsets / sizes from the sampler's per-layer id arrays (the DGL layout, where a layer's offset is the
// sum of the real sizes below it; and the fixed-shape layout of the wide chain)
static int enqueue_pack(pg_sampler* s, const pg_nodeflow_desc_t* o, hipStream_t st) {
  const int L = s->hops;
  int64_t need = 0;
  for (int l = 0; l <= L; ++l) need += s->cap[l];
  PackArgs p{};
  for (int l = 0; l <= L; ++l) {
    p.layer_ids[l] = s->layer_ids[l];
    p.layer_cnt[l] = s->counters + l;
  }
  for (int b = 0; b < L; ++b) p.blk_edges[b] = s->counters + PG_MAX_LAYERS + b;
  p.node_mapping = o->node_mapping; p.layer_offsets = o->layer_offsets; p.sizes_pinned = o->sizes_pinned;
  p.sizes_dev = o->sizes_dev;
  p.cap_nodes = o->cap_nodes; p.num_layers = L + 1;
  p.padded = o->padded ? 1 : 0;
  p.pad_off[0] = 0;
  for (int l = 0; l <= L; ++l) p.pad_off[l + 1] = p.pad_off[l] + (int32_t)s->cap[l];
  hipLaunchKernelGGL(k_pack, dim3(grid_for(need, 256, 1024)), dim3(256), 0, st, p);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

// the five-launch chain (fan-out <= 64)
static int enqueue_chain(pg_sampler* s, const pg_nodeflow_desc_t* o, hipStream_t st, const SampleParams& prm) {
  const int L = s->hops;
  int32_t* lcnt = s->counters;
  int32_t* ecnt = s->counters + PG_MAX_LAYERS;
  int g_shift = 0;
  while ((1 << g_shift) < s->k) ++g_shift;
  const int vpi = 256 >> g_shift;
  int32_t pad_off[PG_MAX_LAYERS + 1];
  pad_off[0] = 0;
  for (int l = 0; l <= L; ++l) pad_off[l + 1] = pad_off[l] + (int32_t)s->cap[l];
  const bool padded = o->padded != 0;

  RelabelXArgs x{};
  int x_blocks = 0;
  for (int b = L - 1; b >= -1; --b) {
    SampleGArgs a{};
    int s_blocks = 0;
    unsigned long long* bm = (s->rank_launches & 1) ? s->bitmap_b : s->bitmap;
    unsigned long long* other = (s->rank_launches & 1) ? s->bitmap : s->bitmap_b;
    if (b >= 0) {
      const int64_t cap_dst = s->cap[b + 1];
      a.indptr = s->indptr; a.indices = s->indices;
      a.cap_rows = (int32_t)cap_dst;
      a.k = s->k; a.g_shift = g_shift;
      a.iters = (int)ceil_div<int64_t>(cap_dst + 1, (int64_t)vpi * s->max_lookback);
      if (a.iters < 1) a.iters = 1;
      s_blocks = (int)ceil_div<int64_t>(cap_dst + 1, (int64_t)vpi * a.iters);
      a.layer = (uint32_t)b; a.epoch = prm.epoch; a.batch = prm.batch; a.seed_lo = prm.seed_lo; a.seed_hi = prm.seed_hi;
      a.bitmap = bm;
      a.blk_indptr = o->blk_indptr + o->blk_indptr_off[b];
      a.blk_src = o->blk_src + o->blk_src_off[b];
      a.ecnt = ecnt + b;
      a.agg = s->agg; a.tag = ++s->tag ? s->tag : ++s->tag; a.err = s->err;
      a.bnd = bnd(s->V);
      if (b == L - 1) {
        a.dst_ids = prm.seeds; a.n_dev = nullptr; a.n_imm = prm.n_seeds;
        a.top_ids = s->layer_ids[L]; a.top_cnt = lcnt + L;
        a.nm_top = padded ? o->node_mapping + pad_off[L] : nullptr;
        a.nm_cap = (int32_t)s->cap[L];
      } else {
        a.dst_ids = s->layer_ids[b + 1]; a.n_dev = lcnt + b + 1;
      }
    }
    hipLaunchKernelGGL(k_sx, dim3(s_blocks + x_blocks), dim3(256), 0, st, a, x, s_blocks);
    PG_LAUNCH_CHECK();
    if (b + 1 < L && !o->defer_transpose && o->blk_tptr && o->blk_tdst && ((o->transpose_mask >> (b + 1)) & 1u)) {
      // block b+1 is final (its relabel rode in the launch above): its source-major copy, in line
      const int rc = transpose_block(s, o, b + 1, lcnt + b + 2, lcnt + b + 1, ecnt + b + 1,
                                     s->counters + 2 * PG_MAX_LAYERS - 1, st);
      if (rc != PG_OK) return rc;
    }
    if (b < 0) break;
    RankArgs r{};
    r.bm = bm; r.other_bm = other; r.n_words = s->n_words;
    if (s->clear_by_ids) {
      // `other` holds the marks of the previous rank launch's layer: layer b + 1 of this call, or — for the first rank
      // launch of a call — layer 0 of the previous call (its ids and count are still in place: this call has not emitted
      // layer 0 yet; before the very first call the count is 0 and the bitmap clean)
      const int lp = b + 1 < L ? b + 1 : 0;
      r.clear_ids = s->layer_ids[lp]; r.clear_cnt = lcnt + lp; r.clear_cap = s->cap[lp];
    }
    r.m = s->rank_m;
    r.agg = s->agg; r.tag = ++s->tag ? s->tag : ++s->tag; r.err = s->err;
    r.out_ids = s->layer_ids[b]; r.cap = s->cap[b]; r.word_rank = s->word_rank; r.count_out = lcnt + b;
    r.nm_out = padded ? o->node_mapping + pad_off[b] : nullptr;
    r.bnd = bnd(s->V);
    hipLaunchKernelGGL(k_bm_rank, dim3(s->rank_blocks), dim3(256), 0, st, r);
    PG_LAUNCH_CHECK();
    ++s->rank_launches;
    // the relabel of this block rides in the next launch
    x = RelabelXArgs{};
    x.blk_src = o->blk_src + o->blk_src_off[b];
    x.ecnt = ecnt + b;
    x.bm = bm; x.word_rank = s->word_rank;
    x.nm_tail = padded ? o->node_mapping + pad_off[b] : nullptr;
    x.lcnt_b = lcnt + b; x.cap_b = (int32_t)s->cap[b];
    x_blocks = grid_for(s->cap[b + 1] * s->k, 256, 256);
    x.bnd = bnd(s->V, s->cap[b + 1] * s->k + 1);
    if (b == 0 && padded) {
      x.final_sizes = 1; x.num_layers = L + 1; x.lcnt = lcnt; x.ecnt_all = ecnt;
      x.layer_offsets = o->layer_offsets; x.sizes_pinned = o->sizes_pinned; x.sizes_dev = o->sizes_dev;
      for (int l = 0; l <= L + 1; ++l) x.pad_off[l] = pad_off[l];
    }
  }
  return PG_OK;
}

int pg_sampler_sample(pg_sampler_t* s, const int64_t* seeds, int32_t n_seeds, uint64_t seed, uint32_t epoch,
                      uint32_t batch, const pg_nodeflow_desc_t* o, pg_stream_t stream) {
  if (!s || !o || !seeds || n_seeds <= 0 || n_seeds > s->B) return PG_ERR_INVALID;
  if (!o->node_mapping || !o->layer_offsets || !o->blk_indptr || !o->blk_src || !o->sizes_pinned)
    return PG_ERR_INVALID;
  int64_t need = 0;
  for (int l = 0; l <= s->hops; ++l) need += s->cap[l];
  if (o->cap_nodes < need) return PG_ERR_OVERFLOW;
  hipStream_t st = as_stream(stream);
  SampleParams prm{};
  prm.seeds = seeds; prm.n_seeds = n_seeds;
  prm.seed_lo = (uint32_t)seed; prm.seed_hi = (uint32_t)(seed >> 32);
  prm.epoch = epoch; prm.batch = batch;
  // Every launch takes the call's scalars as kernel arguments: nothing of this call lives in memory that a later call
  // rewrites, so the host never waits for the device here. One chain at a time per handle (shared scratch: bitmaps, rank
  // table, look-back granules) — calls on one stream, or on streams the caller orders.
  if (s->k > kWave) {
    const int rc = enqueue_chain_wide(s, o, st, prm);
    return rc != PG_OK ? rc : enqueue_pack(s, o, st);
  }
  const int rc = enqueue_chain(s, o, st, prm);
  if (rc != PG_OK || o->padded) return rc;
  return enqueue_pack(s, o, st);
}

int pg_sampler_status(pg_sampler_t* s, int32_t* lookback_timeouts) {
  if (!s || !lookback_timeouts) return PG_ERR_INVALID;
  PG_HIP(hipMemcpy(lookback_timeouts, s->err, 4, hipMemcpyDeviceToHost));     // synchronises: call it off the hot loop
  return PG_OK;
}

static int enqueue_transposes(pg_sampler* s, const pg_nodeflow_desc_t* o, hipStream_t st) {
  for (int b = 0; b < s->hops; ++b) {
    if (!((o->transpose_mask >> b) & 1u)) continue;
    const int rc = transpose_block(s, o, b, o->sizes_dev + b + 1, o->sizes_dev + b, o->sizes_dev + PG_MAX_LAYERS + b,
                                   s->tdummy, st);
    if (rc != PG_OK) return rc;
  }
  return PG_OK;
}

int pg_sampler_transpose(pg_sampler_t* s, const pg_nodeflow_desc_t* o, pg_stream_t stream) {
  if (!s || !o || !o->blk_indptr || !o->blk_src) return PG_ERR_INVALID;
  if (!o->transpose_mask) return PG_OK;
  if (!o->blk_tptr || !o->blk_tdst || !o->sizes_dev) return PG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  // one-workgroup transposes (k_t_block) are one plain launch per block: cheaper for the launch thread than replaying a
  // one-node graph (3-4 us against 10-13)
  bool single = true;
  for (int b = 0; b < s->hops; ++b)
    if ((o->transpose_mask >> b) & 1u) single = single && transpose_one_workgroup(s, b);
  if (single) return enqueue_transposes(s, o, st);
  // the multi-launch device sort of a large block: the launch sequence into a given slot from a given stream is fixed (all
  // sizes are read from the slot's device counters), so from the second call it is one hipGraph launch instead of ~8
  // kernel launches on the trainer's launch thread.
  constexpr bool no_graph = false;
  pg_sampler::SlotState* ss = nullptr;
  for (auto* c : s->tslot_states)
    if (c->key == o->node_mapping && c->stream == st && memcmp(&c->desc, o, sizeof(*o)) == 0) ss = c;
  if (!ss) {
    ss = new (std::nothrow) pg_sampler::SlotState;
    if (!ss) return PG_ERR_NOMEM;
    ss->key = o->node_mapping;
    ss->stream = st;
    ss->desc = *o;
    s->tslot_states.push_back(ss);
  }
  ++ss->calls;
  if (ss->exec) {
    PG_HIP(hipGraphLaunch(ss->exec, st));
    return PG_OK;
  }
  if (no_graph || ss->graph_failed || ss->calls < 2) return enqueue_transposes(s, o, st);
  if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    ss->graph_failed = true;
    return enqueue_transposes(s, o, st);
  }
  int rc = enqueue_transposes(s, o, st);
  hipGraph_t graph = nullptr;
  const hipError_t e_end = hipStreamEndCapture(st, &graph);
  if (rc == PG_OK && e_end == hipSuccess && graph && hipGraphInstantiate(&ss->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
    ss->graph = graph;
    PG_HIP(hipGraphLaunch(ss->exec, st));
    return PG_OK;
  }
  (void)hipGetLastError();
  if (graph) (void)hipGraphDestroy(graph);
  ss->exec = nullptr;
  ss->graph_failed = true;
  return enqueue_transposes(s, o, st);   // nothing ran during the failed capture
}

int pg_frontier_mark_neighbors(const int64_t* indptr, const int32_t* indices, const int64_t* frontier, int64_t n,
                               uint64_t* bitmap, int mark_self, pg_stream_t stream) {
  if (n < 0) return PG_ERR_INVALID;
  if (n == 0) return PG_OK;
  if (!indptr || !indices || !frontier || !bitmap) return PG_ERR_INVALID;
  hipLaunchKernelGGL(k_mark_neighbors, dim3(grid_for(n, 4, 16384)), dim3(256), 0, as_stream(stream), indptr,
                     indices, frontier, n, reinterpret_cast<unsigned long long*>(bitmap), mark_self);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_bitmap_to_ids(const uint64_t* bitmap, int64_t n_words, int64_t* out_ids, int64_t cap, int64_t* count_dev,
                     uint32_t* word_rank, void* scratch, pg_stream_t stream) {
  if (n_words < 0 || cap < 0) return PG_ERR_INVALID;
  if (!bitmap || !out_ids || !count_dev || !scratch) return PG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  const int nb = (int)ceil_div<int64_t>(n_words, kWordsPerBlock);
  if (nb == 0) {
    PG_HIP(hipMemsetAsync(count_dev, 0, 8, st));
    return PG_OK;
  }
  // scratch holds the per-block partials: caller provides >= 4*(nb+1) bytes
  int32_t* partial = reinterpret_cast<int32_t*>(scratch);
  const unsigned long long* bm = reinterpret_cast<const unsigned long long*>(bitmap);
  hipLaunchKernelGGL(k_bm_count, dim3(nb), dim3(256), 0, st, bm, n_words, partial, (unsigned long long*)nullptr);
  PG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bm_scan, dim3(1), dim3(kScanThreads), 0, st, partial, nb, (int32_t*)nullptr, count_dev);
  PG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bm_emit, dim3(nb), dim3(256), 0, st, bm, n_words, partial, out_ids, cap, word_rank);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // extern "C"

#ifdef PG_T_STAMPS
extern "C" int pg_debug_t_stamps(long long* out16) {
  PG_HIP(hipDeviceSynchronize());
  PG_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(pg::g_t_stamps), 16 * sizeof(long long)));
  return PG_OK;
}
#endif
