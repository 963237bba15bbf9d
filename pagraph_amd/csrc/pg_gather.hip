// Feature-cache gather for MI355X (gfx950): the HBM-bound hot kernel.
//
// Replaces the op sequence of PaGraph/storage/storage.py:176-204 (fetch_data)
// and :207-216 (fetch_from_cache): per layer ~10 torch launches, 2 boolean-mask
// compactions (each a device->host sync) and a temp copy of the hit rows —
// with ONE launch over all rows of all layers.
//
// Work decomposition (chosen by measurement, tools/gather_variants.hip; see
// DESIGN.md "k_gather"):
//   * a 256-thread block owns T consecutive output rows (T = 8 for big
//     launches, 4 at the minibatch shape so the grid stays >> 256 CUs);
//   * stage: thread t < T loads ids[row0+t] and its slot_map entry into LDS
//     (the index tile is read from HBM exactly once per block);
//   * hit/miss split (storage.py:179-182,117) runs in its own tiny kernel
//     k_split just before: 256 rows per block, per-wave __ballot of "slot < 0",
//     prefix-popcount ranks, ONE atomicAdd per block to reserve a range of the
//     miss list.  Fusing it into the copy kernel (one atomic per 4..8-row
//     tile) serialised ~12 ns per tile on the single counter word: 34 K rows
//     with 25 % misses took 74 us instead of 31 us (measured);
//   * copy: the tile's T x (dim/VEC) 16-byte pieces are one FLAT index space
//     striped over the block's 256 lanes — every lane of every load/store
//     instruction is busy (a row of 600 floats = 150 pieces would leave 42 of
//     64 lanes idle on its third wave-load), reads are contiguous 2.4 KB runs
//     of the cache rows, the tile's output is one contiguous T*dim*4-byte run.
//     Each lane issues U independent 16-byte loads before its first store;
//     (row, piece) advance incrementally, no division in the loop;
//   * stores are non-temporal: the gathered frame is consumed by the next
//     kernel once and must not evict cache rows from L2/MALL (+7..10 %);
//   * narrow fields (e.g. 'norm', dim 1) are copied lane-per-row.
#include <cstdlib>

#include <hip/hip_ext.h>

#include "pg_common.h"

namespace pg {

struct GatherArgs {
  const int64_t* ids;
  const int32_t* slots;  // optional: slot of row r, already translated by k_split
  const int32_t* slot_map;
  const int64_t* nid_map;
  int32_t* miss_pos;
  int64_t* miss_fullid;
  int32_t* miss_count;
  int64_t n;
  int32_t n_fields;
  int32_t vec[PG_MAX_FIELDS];  // 4 / 2 / 1 floats per access; 0 = lane-per-row (narrow field)
  pg_field_t f[PG_MAX_FIELDS];
  Bnd bnd;                     // PG_BOUNDS: [0] vertices of the partition (an id is followed into slot_map)
};

typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));

template <int VEC>
struct VecT;
template <>
struct VecT<4> { using type = vf4; };
template <>
struct VecT<2> { using type = vf2; };
template <>
struct VecT<1> { using type = float; };

constexpr int kGatherBlock = 256;

// flat copy of one field of the block's tile: rows [row0, row0+rows), slots in LDS
template <int VEC, int U>
__device__ __forceinline__ void copy_tile(const pg_field_t fd, const int32_t* s_slot, int64_t row0, int rows) {
  using V = typename VecT<VEC>::type;
  const int pieces = fd.dim / VEC;
  const int total = rows * pieces;
  const int step_r = kGatherBlock / pieces, step_c = kGatherBlock % pieces;
  int r = threadIdx.x / pieces, c = threadIdx.x % pieces;
  for (int i0 = threadIdx.x; i0 < total; i0 += kGatherBlock * U) {
    V v[U];
    V* dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      dst[u] = nullptr;
      if (i0 + u * kGatherBlock < total) {
        const int32_t s = s_slot[r];
        if (s >= 0) {
          v[u] = reinterpret_cast<const V*>(fd.cache + (int64_t)s * fd.cache_stride)[c];
          dst[u] = reinterpret_cast<V*>(fd.out + (row0 + r) * fd.out_stride) + c;
        }
      }
      r += step_r;
      c += step_c;
      if (c >= pieces) {
        c -= pieces;
        ++r;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dst[u]) __builtin_nontemporal_store(v[u], dst[u]);
  }
}

// Miss-list index dedup (north star: "index dedup" in the gather; the reference fetches a vertex once per LAYER it
// appears in, storage.py:176-200). A NodeFlow's layers are each duplicate-free, but a vertex can sit in several
// layers; for a cache HIT the repeat costs an L2/MALL read, for a MISS it costs a second trip over PCIe — the
// bound of the step at a partial cache (measured on the benchmark graph: 6.0 % of the miss rows of a minibatch
// are such repeats, tools/exp_dup_census.py). Non-seed layers are sorted by id (sampler spec rule 5), so a missed
// row of layer l looks its id up in the layers before it with a binary search over the id array itself — no hash
// table, no extra memory. A repeat is NOT appended to the miss list; it goes to the dup list as (row, earlier row)
// and is filled on the device from the earlier row's staged copy (pg_scatter_rows_dups) once that has landed.
struct SplitDedup {
  int32_t n_ranges;
  int32_t lo[PG_MAX_LAYERS + 1];
  uint32_t sorted_mask;
  int32_t* dup_pos;
  int32_t* dup_src;
  int32_t* dup_count;
};

// first row in [lo, hi) whose id is >= `id`, ids ascending with the padding (< 0) of a fixed-shape layer at the end
__device__ __forceinline__ int32_t lower_bound_ids(const int64_t* __restrict__ ids, int32_t lo, int32_t hi, int64_t id) {
  while (lo < hi) {
    const int32_t mid = (int32_t)(((int64_t)lo + hi) >> 1);
    const int64_t v = ids[mid];
    if (v >= 0 && v < id) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// hit/miss split: appends (row, nid_map[id]) of every row with slot_map[id] < 0 to the miss list
template <bool DEDUP>
__global__ __launch_bounds__(256) void k_split(const int64_t* __restrict__ ids, int64_t n,
                                               const int32_t* __restrict__ slot_map,
                                               const int64_t* __restrict__ nid_map, int32_t* __restrict__ miss_pos,
                                               int64_t* __restrict__ miss_fullid, int32_t* __restrict__ miss_count,
                                               int32_t* __restrict__ slots_out,
                                               unsigned long long* __restrict__ stats, const SplitDedup dd, const Bnd bnd) {
  // PG_BOUNDS: [0] vertices of the partition (an id is followed into slot_map and nid_map)
  __shared__ int32_t s_wave[4], s_valid[4], s_dups[4];
  __shared__ int32_t s_base;
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t id = 0;
  bool miss = false;
  if (row < n) {
    id = ids[row];
    if (id >= 0) id = PG_IDX(id, bnd, 0, PG_K_SPLIT, 1);
    const int32_t s = id < 0 ? -2 : slot_map[id];   // id < 0: padding of a fixed-shape NodeFlow
    miss = s == -1;
    if (slots_out) slots_out[row] = s;  // coalesced; k_gather then skips the random slot_map lookup
  }
  int32_t n_dup_wave = 0;
  if constexpr (DEDUP) {
    int32_t dup_of = -1;
    if (miss) {
      int r = 0;
      while (r + 1 < dd.n_ranges && row >= dd.lo[r + 1]) ++r;
      // earliest layer first: the row found there is a primary miss itself (it has nothing before it to repeat)
      for (int q = 0; q < r && dup_of < 0; ++q) {
        if (!((dd.sorted_mask >> q) & 1u)) continue;
        const int32_t a = lower_bound_ids(ids, dd.lo[q], dd.lo[q + 1], id);
        if (a < dd.lo[q + 1] && ids[a] == id) dup_of = a;
      }
    }
    const unsigned long long dmask = __ballot(dup_of >= 0);
    n_dup_wave = (int32_t)__popcll(dmask);
    if (dmask) {
      int32_t base = 0;
      if (lane == 0) base = atomicAdd(dd.dup_count, n_dup_wave);
      base = __shfl(base, 0);
      if (dup_of >= 0) {
        const int32_t k = base + (int32_t)__popcll(dmask & ((1ull << lane) - 1ull));
        dd.dup_pos[k] = (int32_t)row;
        dd.dup_src[k] = dup_of;
        miss = false;            // stays out of the miss list; its slot stays -1 (k_gather skips it)
      }
    }
  }
  const unsigned long long mmask = __ballot(miss);
  const unsigned long long vmask = __ballot(row < n && id >= 0);
  if (lane == 0) {
    s_wave[w] = (int32_t)__popcll(mmask);
    s_valid[w] = (int32_t)__popcll(vmask);
    s_dups[w] = n_dup_wave;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int32_t tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = tot ? atomicAdd(miss_count, tot) : 0;
    if (stats) {  // storage.py:219-221 log_miss_rate, kept on the device; a repeated miss counts as a miss, as there
      const int32_t missed = tot + s_dups[0] + s_dups[1] + s_dups[2] + s_dups[3];
      atomicAdd(&stats[0], (unsigned long long)(s_valid[0] + s_valid[1] + s_valid[2] + s_valid[3]));
      if (missed) atomicAdd(&stats[1], (unsigned long long)missed);
    }
  }
  __syncthreads();
  if (miss) {
    int32_t off = s_base + __popcll(mmask & ((1ull << lane) - 1ull));
    for (int i = 0; i < w; ++i) off += s_wave[i];
    miss_pos[off] = (int32_t)row;
    miss_fullid[off] = nid_map[id];
    // a consumer that reads rows where they lie (pg_spmm_fwd_rows) finds miss row `off` of the staged block here
    if (slots_out) slots_out[row] = -(off + 3);
  }
}

template <int T, int U, bool FULL>
__global__ __launch_bounds__(kGatherBlock) void k_gather(const GatherArgs a) {
  static_assert(T <= kWave, "the tile's index stage and miss ballot live in wave 0");
  __shared__ int32_t s_slot[T];
  const int64_t row0 = (int64_t)blockIdx.x * T;
  const int rows = (int)((a.n - row0) < T ? (a.n - row0) : T);
  const int t = threadIdx.x;

  if (t < kWave) {  // wave 0: stage id -> slot
    const bool valid = t < rows;
    int64_t id = 0;
    int32_t slot = -2;
    if (valid) {
      if (!FULL && a.slots) {
        slot = a.slots[row0 + t];
      } else {
        id = a.ids[row0 + t];
        if (id >= 0) id = PG_IDX(id, a.bnd, 0, PG_K_GATHER, 1);
        slot = id < 0 ? -2 : (FULL ? (int32_t)id : a.slot_map[id]);
      }
      s_slot[t] = slot;
    }
    // narrow fields: lane-per-row (field loops are fully unrolled with constant indices so the
    // by-value kernel-argument struct stays in SGPRs instead of being copied to an alloca)
#pragma unroll
    for (int f = 0; f < PG_MAX_FIELDS; ++f) {
      if (f >= a.n_fields || a.vec[f] != 0) continue;
      const pg_field_t fd = a.f[f];
      if (valid && slot >= 0) {
        const float* src = fd.cache + (int64_t)slot * fd.cache_stride;
        float* dst = fd.out + (row0 + t) * fd.out_stride;
        for (int c = 0; c < fd.dim; ++c) dst[c] = src[c];
      }
    }
  }
  __syncthreads();

#pragma unroll
  for (int f = 0; f < PG_MAX_FIELDS; ++f) {
    if (f >= a.n_fields) continue;
    const int vec = a.vec[f];
    if (vec == 4) copy_tile<4, U>(a.f[f], s_slot, row0, rows);
    else if (vec == 2) copy_tile<2, U>(a.f[f], s_slot, row0, rows);
    else if (vec == 1) copy_tile<1, U>(a.f[f], s_slot, row0, rows);
  }
}

// out[pos[j], :] = staged[src_row ? src_row[j] : j, :]  — wave per row (storage.py:199-200)
template <int VEC>
__global__ __launch_bounds__(256) void k_scatter(const float* __restrict__ staged,
                                                 const int32_t* __restrict__ pos, int64_t n,
                                                 const int32_t* __restrict__ n_dev, int32_t dim,
                                                 float* __restrict__ out, int32_t out_stride, int32_t pos_lo,
                                                 const int32_t* __restrict__ src_row, int32_t staged_stride) {
  using V = typename VecT<VEC>::type;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t nn = n_dev ? (int64_t)*n_dev : n;
  const int64_t waves = (int64_t)gridDim.x * (blockDim.x / kWave);
  const int pieces = dim / VEC;
  for (int64_t j = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; j < nn; j += waves) {
    const int32_t p = pos[j] - pos_lo;
    if (p < 0) continue;          // a row below pos_lo stays where it is: its consumer reads the staged block
    const int64_t sj = src_row ? (int64_t)src_row[j] : j;   // dup list: staged row of the earlier occurrence
    if (sj < 0) continue;
    const V* src = reinterpret_cast<const V*>(staged + sj * staged_stride);
    V* dst = reinterpret_cast<V*>(out + (int64_t)p * out_stride);
    for (int c = lane; c < pieces; c += kWave) dst[c] = src[c];
  }
}

// zero-copy miss path: out[pos[j], :] = table_pinned[fullid[j], :] read over PCIe.
// Launched with a SMALL grid on purpose: the waves spend microseconds stalled on PCIe reads, and
// a one-wave-per-row grid starves the compute stream that is supposed to overlap with it
// (measured, tools/exp_overlap.py: 8.4 K rows, 405 us alone; + 155 us of GEMMs on another stream
// = 474 us with 2100 blocks, 431 us with 128 blocks, 389 us with 48 blocks; PCIe stays saturated
// at ~52 GB/s down to 32 blocks and collapses at 16). 48 blocks x 4 waves x 2 rows in flight.
template <int VEC>
__global__ __launch_bounds__(256) void k_scatter_host(const float* __restrict__ table,
                                                      int64_t table_stride,
                                                      const int32_t* __restrict__ pos,
                                                      const int64_t* __restrict__ fullid, int64_t n,
                                                      const int32_t* __restrict__ n_dev, int32_t dim,
                                                      float* __restrict__ out, int32_t out_stride,
                                                      int32_t start_num, int32_t pos_lo) {
  using V = typename VecT<VEC>::type;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t nn = n_dev ? (int64_t)*n_dev : n;
  const int64_t j0 = nn * start_num / 256;      // rows [0, j0) belong to the CPU worker (pg_missq_set_cpu_share)
  const int64_t waves = (int64_t)gridDim.x * (blockDim.x / kWave);
  const int pieces = dim / VEC;
  for (int64_t j = j0 + (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; j < nn; j += 2 * waves) {
    const int64_t j2 = j + waves;
    const bool two = j2 < nn;
    // pos == NULL: row j of the miss list goes to row j of `out` (a staged block in miss-list order); else to row
    // pos[j] - pos_lo, rows below pos_lo skipped (their consumer reads the staged block)
    const int64_t p0 = pos ? (int64_t)pos[j] - pos_lo : j;
    const int64_t p1 = pos ? (int64_t)pos[two ? j2 : j] - pos_lo : (two ? j2 : j);
    const bool w0 = p0 >= 0, w1 = two && p1 >= 0;
    const V* src0 = reinterpret_cast<const V*>(table + fullid[j] * table_stride);
    const V* src1 = reinterpret_cast<const V*>(table + fullid[two ? j2 : j] * table_stride);
    V* dst0 = reinterpret_cast<V*>(out + (w0 ? p0 : 0) * out_stride);
    V* dst1 = reinterpret_cast<V*>(out + (w1 ? p1 : 0) * out_stride);
    if (!w0 && !w1) continue;
    for (int c = lane; c < pieces; c += kWave) {
      const V a = src0[c];
      const V b = src1[c];
      if (w0) dst0[c] = a;
      if (w1) dst1[c] = b;
    }
  }
}
// k_scatter_host_wide: the same job with few waves that each keep MANY reads in flight. In the training loop the zero-copy
// reads share the chip with the HBM-bound kernels of the other streams, and what those pay for is the NUMBER of waves
// parked on PCIe reads, not the bytes in flight (round 3, share 0 of every miss list through this kernel: 24 blocks
// 0.219 ms/step, 48 -> 0.250, 96 -> 0.279, 192 -> 0.288 with the fused gather+aggregate kernel at 86 us instead of 19).
// Each wave therefore takes kRowsWide rows at a time and issues every piece of all of them (rows of <= 4 * 64 pieces)
// before the first store: 12 KB in flight per wave at F = 600.
constexpr int kRowsWide = 4;
template <int VEC, int P>
__global__ __launch_bounds__(256) void k_scatter_host_wide(const float* __restrict__ table, int64_t table_stride,
                                                           const int32_t* __restrict__ pos,
                                                           const int64_t* __restrict__ fullid, int64_t n,
                                                           const int32_t* __restrict__ n_dev, int32_t dim,
                                                           float* __restrict__ out, int32_t out_stride,
                                                           int32_t start_num, int32_t pos_lo) {
  using V = typename VecT<VEC>::type;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t nn = n_dev ? (int64_t)*n_dev : n;
  const int64_t j0 = nn * start_num / 256;
  const int64_t waves = (int64_t)gridDim.x * (blockDim.x / kWave);
  const int pieces = dim / VEC;
  for (int64_t jb = j0 + ((int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave) * kRowsWide; jb < nn;
       jb += waves * kRowsWide) {
    V x[kRowsWide][P];
    int64_t p[kRowsWide];
#pragma unroll
    for (int r = 0; r < kRowsWide; ++r) {
      const int64_t j = jb + r;
      p[r] = -1;
      if (j < nn) {
        p[r] = pos ? (int64_t)pos[j] - pos_lo : j;
        if (p[r] >= 0) {
          const V* src = reinterpret_cast<const V*>(table + fullid[j] * table_stride);
#pragma unroll
          for (int u = 0; u < P; ++u) {
            const int c = lane + u * kWave;
            if (c < pieces) x[r][u] = src[c];
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kRowsWide; ++r) {
      if (p[r] < 0) continue;
      V* dst = reinterpret_cast<V*>(out + p[r] * out_stride);
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int c = lane + u * kWave;
        if (c < pieces) dst[c] = x[r][u];
      }
    }
  }
}


__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = v;
}
__global__ void k_slot_assign(int32_t* slot_map, const int64_t* nids, int64_t rows) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x)
    slot_map[nids[i]] = (int32_t)i;
}
__global__ void k_slot_export(const int32_t* slot_map, int64_t n, uint8_t* flag, int64_t* l2c) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t s = slot_map[i];
    if (flag) flag[i] = s >= 0;
    if (l2c) l2c[i] = s >= 0 ? s : 0;
  }
}

static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// widest access both sides allow; 0 selects the lane-per-row path for narrow rows
static int pick_vec(const pg_field_t& f, bool need_cache) {
  if (f.dim < 16) return 0;
  auto ok = [&](int v) {
    const size_t b = (size_t)v * 4;
    return f.dim % v == 0 && f.out_stride % v == 0 && aligned(f.out, b) &&
           (!need_cache || (f.cache_stride % v == 0 && aligned(f.cache, b)));
  };
  if (ok(4)) return 4;
  if (ok(2)) return 2;
  return 1;
}

static inline int grid_1d(int64_t n, int block, int cap = 4096) {
  int64_t g = ceil_div<int64_t>(n, block);
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

template <bool FULL>
static int launch_gather(GatherArgs& a, hipStream_t st, pg_timer* timer = nullptr) {
  // timer: its events are attached to THIS dispatch (hipExtLaunchKernelGGL), i.e. they read the kernel's own
  // begin / end timestamps — what rocprofv3 reports — not the gaps to neighbouring event packets
  // tile height: 8 rows/block once the launch has >= 32K blocks anyway, 4 rows/block at the
  // minibatch shape (~34K rows -> ~8.5K blocks) to keep every CU fed through the tail.
  if (!FULL && a.slot_map) a.bnd = bnd(bounds_elems(a.slot_map, 4));
  if (a.n >= (int64_t)1 << 18) {
    const int64_t blocks = ceil_div<int64_t>(a.n, 8);
    if (timer)
      hipExtLaunchKernelGGL((k_gather<8, 4, FULL>), dim3((unsigned)blocks), dim3(kGatherBlock), 0, st, timer->start,
                            timer->stop, 0, a);
    else
      hipLaunchKernelGGL((k_gather<8, 4, FULL>), dim3((unsigned)blocks), dim3(kGatherBlock), 0, st, a);
  } else {
    const int64_t blocks = ceil_div<int64_t>(a.n, 4);
    if (timer)
      hipExtLaunchKernelGGL((k_gather<4, 3, FULL>), dim3((unsigned)blocks), dim3(kGatherBlock), 0, st, timer->start,
                            timer->stop, 0, a);
    else
      hipLaunchKernelGGL((k_gather<4, 3, FULL>), dim3((unsigned)blocks), dim3(kGatherBlock), 0, st, a);
  }
  PG_LAUNCH_CHECK();
  return PG_OK;
}

static int fill_args(GatherArgs& a, const pg_field_t* fields, int n_fields, bool need_cache) {
  if (n_fields < 0 || n_fields > PG_MAX_FIELDS || (n_fields > 0 && !fields)) return PG_ERR_INVALID;
  a.n_fields = n_fields;
  for (int i = 0; i < n_fields; ++i) {
    const pg_field_t& f = fields[i];
    if (f.dim <= 0 || !f.out || f.out_stride < f.dim) return PG_ERR_INVALID;
    if (need_cache && (!f.cache || f.cache_stride < f.dim)) return PG_ERR_INVALID;
    a.f[i] = f;
    a.vec[i] = pick_vec(f, f.cache != nullptr);
  }
  return PG_OK;
}

// batch_labels = labels[batch_nids] (examples/profile/pa_gcn.py:99-100); ids < 0 are the padding of a
// fixed-shape NodeFlow and get `fill` (the loss's ignore_index)
__global__ __launch_bounds__(256) void k_gather_labels(const int64_t* __restrict__ ids, int64_t n,
                                                       const int64_t* __restrict__ labels, int64_t n_labels,
                                                       int64_t fill, int64_t* __restrict__ out,
                                                       int32_t* __restrict__ n_valid) {
  int mine = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = ids[i];
    const int64_t l = (v >= 0 && v < n_labels) ? labels[v] : fill;
    out[i] = l;
    mine += (l != fill && l >= 0) ? 1 : 0;
  }
  if (n_valid) {   // rows the loss will count (label != fill): one atomic per wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && mine) atomicAdd(n_valid, mine);
  }
}

// the same with a self-cleaning count (round 4): the waves add to scratch[0], a block that finishes takes a ticket from
// scratch[1], and the LAST block publishes *n_valid = scratch[0] and zeroes both words for the next launch — no zero fill in
// front of the launch (a launch of its own on the load stream and ~4 us of the launch thread per batch). A batch is ~24 blocks.
__global__ __launch_bounds__(256) void k_gather_labels_sc(const int64_t* __restrict__ ids, int64_t n,
                                                          const int64_t* __restrict__ labels, int64_t n_labels,
                                                          int64_t fill, int64_t* __restrict__ out,
                                                          int32_t* __restrict__ n_valid, int32_t* __restrict__ scratch) {
  int mine = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = ids[i];
    const int64_t l = (v >= 0 && v < n_labels) ? labels[v] : fill;
    out[i] = l;
    mine += (l != fill && l >= 0) ? 1 : 0;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, kWave);
  __shared__ int s_cnt[4];
  if ((threadIdx.x & (kWave - 1)) == 0) s_cnt[threadIdx.x / kWave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    if (t) atomicAdd(&scratch[0], t);
    __threadfence();                                   // this block's sum is visible before its ticket is
    const int ticket = atomicAdd(&scratch[1], 1);
    if (ticket == (int)gridDim.x - 1) {                // every other block's sum is in: publish and clean up
      __threadfence();
      const int total = atomicExch(&scratch[0], 0);
      *n_valid = total;
      atomicExch(&scratch[1], 0);
    }
  }
}

static int launch_split(const int64_t* ids, int64_t n, const int32_t* slot_map, const int64_t* nid_map, int32_t* miss_pos,
                        int64_t* miss_fullid, int32_t* miss_count, int32_t* slots_out, uint64_t* stats,
                        const pg_dedup_t* dedup, hipStream_t st) {
  SplitDedup dd{};
  const unsigned grid = (unsigned)ceil_div<int64_t>(n, 256);
  unsigned long long* sp = reinterpret_cast<unsigned long long*>(stats);
  if (dedup) {
    if (dedup->n_ranges < 1 || dedup->n_ranges > PG_MAX_LAYERS || !dedup->dup_pos || !dedup->dup_src || !dedup->dup_count)
      return PG_ERR_INVALID;
    if (dedup->lo[0] != 0 || dedup->lo[dedup->n_ranges] != n) return PG_ERR_INVALID;
    for (int r = 0; r < dedup->n_ranges; ++r)
      if (dedup->lo[r] > dedup->lo[r + 1]) return PG_ERR_INVALID;
    dd.n_ranges = dedup->n_ranges;
    for (int r = 0; r <= dedup->n_ranges; ++r) dd.lo[r] = dedup->lo[r];
    dd.sorted_mask = dedup->sorted_mask;
    dd.dup_pos = dedup->dup_pos; dd.dup_src = dedup->dup_src; dd.dup_count = dedup->dup_count;
  }
  if (dedup && dedup->n_ranges > 1 && (dedup->sorted_mask & ((1u << (dedup->n_ranges - 1)) - 1u)))
    hipLaunchKernelGGL(k_split<true>, dim3(grid), dim3(256), 0, st, ids, n, slot_map, nid_map, miss_pos, miss_fullid,
                       miss_count, slots_out, sp, dd, bnd(bounds_elems(slot_map, 4)));
  else
    hipLaunchKernelGGL(k_split<false>, dim3(grid), dim3(256), 0, st, ids, n, slot_map, nid_map, miss_pos, miss_fullid,
                       miss_count, slots_out, sp, dd, bnd(bounds_elems(slot_map, 4)));
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // namespace pg

using namespace pg;

extern "C" {

int pg_slot_map_reset(int32_t* slot_map, int64_t node_num, pg_stream_t stream) {
  if (node_num < 0 || (node_num > 0 && !slot_map)) return PG_ERR_INVALID;
  if (node_num == 0) return PG_OK;
  hipLaunchKernelGGL(k_fill_i32, dim3(grid_1d(node_num, 256)), dim3(256), 0, as_stream(stream), slot_map,
                     node_num, -1);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_slot_map_assign(int32_t* slot_map, const int64_t* nids, int64_t rows, pg_stream_t stream) {
  if (rows < 0 || rows > INT32_MAX || (rows > 0 && (!slot_map || !nids))) return PG_ERR_INVALID;
  if (rows == 0) return PG_OK;
  hipLaunchKernelGGL(k_slot_assign, dim3(grid_1d(rows, 256)), dim3(256), 0, as_stream(stream), slot_map,
                     nids, rows);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_slot_map_export(const int32_t* slot_map, int64_t node_num, uint8_t* gpu_flag,
                       int64_t* localid2cacheid, pg_stream_t stream) {
  if (node_num < 0 || (node_num > 0 && !slot_map)) return PG_ERR_INVALID;
  if (node_num == 0) return PG_OK;
  hipLaunchKernelGGL(k_slot_export, dim3(grid_1d(node_num, 256)), dim3(256), 0, as_stream(stream),
                     slot_map, node_num, gpu_flag, localid2cacheid);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_gather_rows(const int64_t* ids, int64_t n, const int32_t* slot_map, const int64_t* nid_map,
                   const pg_field_t* fields, int n_fields, const pg_miss_list_t* miss, int32_t* slot_scratch,
                   uint64_t* stats, pg_timer_t* timer, const pg_dedup_t* dedup, pg_stream_t stream) {
  if (!miss) return PG_ERR_INVALID;
  int32_t* miss_pos = miss->pos;
  int64_t* miss_fullid = miss->fullid;
  int32_t* miss_count = miss->count;
  if (n < 0 || n > INT32_MAX || !miss_count) return PG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  PG_HIP(hipMemsetAsync(miss_count, 0, sizeof(int32_t), st));
  if (dedup && dedup->dup_count) PG_HIP(hipMemsetAsync(dedup->dup_count, 0, sizeof(int32_t), st));
  if (n == 0) return PG_OK;
  if (!ids || !slot_map || !nid_map || !miss_pos || !miss_fullid) return PG_ERR_INVALID;
  if (dedup && !slot_scratch) return PG_ERR_INVALID;      // the repeats are resolved through the slot array
  GatherArgs a{};
  a.ids = ids; a.slot_map = slot_map; a.nid_map = nid_map;
  a.miss_pos = miss_pos; a.miss_fullid = miss_fullid; a.miss_count = miss_count;
  a.slots = slot_scratch;
  a.n = n;
  // a partially cached server may have an empty cache (cache == NULL): every row misses
  int rc = fill_args(a, fields, n_fields, false);
  if (rc != PG_OK) return rc;
  rc = launch_split(ids, n, slot_map, nid_map, miss_pos, miss_fullid, miss_count, slot_scratch, stats, dedup, st);
  if (rc != PG_OK) return rc;
  // the timer brackets ONLY the copy kernel (what rocprofv3 reports as pg::k_gather)
  return launch_gather<false>(a, st, timer);
}

int pg_split_rows(const int64_t* ids, int64_t n, const int32_t* slot_map, const int64_t* nid_map,
                  const pg_miss_list_t* miss, int32_t* slots_out, uint64_t* stats, const pg_dedup_t* dedup,
                  pg_stream_t stream) {
  if (!miss) return PG_ERR_INVALID;
  int32_t* miss_pos = miss->pos;
  int64_t* miss_fullid = miss->fullid;
  int32_t* miss_count = miss->count;
  if (n < 0 || n > INT32_MAX || !miss_count) return PG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  PG_HIP(hipMemsetAsync(miss_count, 0, sizeof(int32_t), st));
  if (dedup && dedup->dup_count) PG_HIP(hipMemsetAsync(dedup->dup_count, 0, sizeof(int32_t), st));
  if (n == 0) return PG_OK;
  if (!ids || !slot_map || !nid_map || !miss_pos || !miss_fullid || !slots_out) return PG_ERR_INVALID;
  return launch_split(ids, n, slot_map, nid_map, miss_pos, miss_fullid, miss_count, slots_out, stats, dedup, st);
}

// slots of a launch over a FULLY cached table (storage.py:207-216's fetch_from_cache, for rows that are read in place): no
// miss list, so no counter to zero first — one launch instead of a 4-byte fill + the split (the launch thread bounds the
// step once the table is cached). Padding ids (< 0) -> -2; stats[0] += rows looked up (storage.py:203's try counter).
__global__ __launch_bounds__(256) void k_slots_full(const int64_t* __restrict__ ids, int64_t n,
                                                    const int32_t* __restrict__ slot_map, int32_t* __restrict__ slots_out,
                                                    unsigned long long* __restrict__ stats, const Bnd bnd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int valid = 0;
  if (i < n) {
    int64_t v = ids[i];
    valid = v >= 0;
    if (valid) v = PG_IDX(v, bnd, 0, PG_K_SPLIT, 2);
    slots_out[i] = valid ? slot_map[v] : -2;
  }
  if (stats) {
    const unsigned long long b = __ballot(valid);
    __shared__ int s_cnt[4];
    if ((threadIdx.x & (kWave - 1)) == 0) s_cnt[threadIdx.x / kWave] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
      const int t = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
      if (t) atomicAdd(&stats[0], (unsigned long long)t);
    }
  }
}

int pg_slots_full(const int64_t* ids, int64_t n, const int32_t* slot_map, int32_t* slots_out, uint64_t* stats,
                  pg_stream_t stream) {
  if (n < 0 || n > INT32_MAX) return PG_ERR_INVALID;
  if (n == 0) return PG_OK;
  if (!ids || !slot_map || !slots_out) return PG_ERR_INVALID;
  hipLaunchKernelGGL(k_slots_full, dim3((unsigned)ceil_div<int64_t>(n, 256)), dim3(256), 0, as_stream(stream), ids, n, slot_map,
                     slots_out, reinterpret_cast<unsigned long long*>(stats), bnd(bounds_elems(slot_map, 4)));
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_scatter_rows_dups(const float* staged, const int32_t* dup_pos, const int32_t* dup_staged_row, int64_t cap,
                         const int32_t* dup_count_dev, int32_t dim, float* out, int32_t out_stride, int32_t pos_lo,
                         pg_stream_t stream) {
  if (!dup_staged_row || !dup_count_dev) return PG_ERR_INVALID;
  return pg_scatter_rows_strided(staged, dim, dup_pos, dup_staged_row, cap, dup_count_dev, dim, out, out_stride, pos_lo, 64,
                                 stream);
}

int pg_scatter_rows_strided(const float* staged, int32_t staged_stride, const int32_t* pos, const int32_t* src_row,
                            int64_t n, const int32_t* n_dev, int32_t dim, float* out, int32_t out_stride, int32_t pos_lo,
                            int32_t max_blocks, pg_stream_t stream) {
  if (n < 0 || dim <= 0 || out_stride < dim || staged_stride < dim || pos_lo < 0) return PG_ERR_INVALID;
  if (n == 0) return PG_OK;
  if (!staged || !pos || !out) return PG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  int grid = grid_1d(n, 4, 8192);
  if (max_blocks > 0 && grid > max_blocks) grid = max_blocks;   // a dup list: a few hundred rows whatever the bound says
#define PG_SCATTER(V)                                                                                              \
  hipLaunchKernelGGL(k_scatter<V>, dim3(grid), dim3(256), 0, st, staged, pos, n, n_dev, dim, out, out_stride, pos_lo, \
                     src_row, staged_stride)
  if (dim % 4 == 0 && out_stride % 4 == 0 && staged_stride % 4 == 0 && aligned(staged, 16) && aligned(out, 16)) PG_SCATTER(4);
  else if (dim % 2 == 0 && out_stride % 2 == 0 && staged_stride % 2 == 0 && aligned(staged, 8) && aligned(out, 8)) PG_SCATTER(2);
  else PG_SCATTER(1);
#undef PG_SCATTER
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_gather_rows_presplit(const int32_t* slots, int64_t n, const pg_field_t* fields, int n_fields,
                            pg_timer_t* timer, pg_stream_t stream) {
  if (n < 0 || n > INT32_MAX) return PG_ERR_INVALID;
  if (n == 0) return PG_OK;
  if (!slots) return PG_ERR_INVALID;
  GatherArgs a{};
  a.slots = slots;
  a.n = n;
  int rc = fill_args(a, fields, n_fields, false);
  if (rc != PG_OK) return rc;
  return launch_gather<false>(a, as_stream(stream), timer);
}

int pg_gather_rows_full(const int64_t* ids, int64_t n, const pg_field_t* fields, int n_fields,
                        pg_stream_t stream) {
  if (n < 0 || n > INT32_MAX) return PG_ERR_INVALID;
  if (n == 0) return PG_OK;
  if (!ids) return PG_ERR_INVALID;
  GatherArgs a{};
  a.ids = ids; a.n = n;
  int rc = fill_args(a, fields, n_fields, true);
  if (rc != PG_OK) return rc;
  return launch_gather<true>(a, as_stream(stream));
}

int pg_gather_labels(const int64_t* ids, int64_t n, const int64_t* labels, int64_t n_labels, int64_t fill,
                     int64_t* out, int32_t* n_valid_out, pg_stream_t stream) {
  if (n < 0 || n_labels < 0) return PG_ERR_INVALID;
  // (the count is accumulated with one atomic per wave behind a 4-byte zero fill. Round 4 tried ONE workgroup that walks the
  // batch and writes the count — no fill launch: its own latency, one block's six dependent rounds, sits on the
  // sampler -> load -> compute chain and cost 5 us per step with the table cached: 0.1134-0.1148 against 0.1082-0.1093 ms)
  if (n_valid_out) PG_HIP(hipMemsetAsync(n_valid_out, 0, sizeof(int32_t), as_stream(stream)));
  if (n == 0) return PG_OK;
  if (!ids || !out || (!labels && n_labels > 0)) return PG_ERR_INVALID;
  int64_t g = ceil_div<int64_t>(n, 256);
  hipLaunchKernelGGL(k_gather_labels, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, as_stream(stream), ids, n,
                     labels, n_labels, fill, out, n_valid_out);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_gather_labels_sc(const int64_t* ids, int64_t n, const int64_t* labels, int64_t n_labels, int64_t fill,
                        int64_t* out, int32_t* n_valid_out, int32_t* scratch2, pg_stream_t stream) {
  if (n <= 0 || n_labels < 0 || !n_valid_out || !scratch2) return PG_ERR_INVALID;
  if (!ids || !out || (!labels && n_labels > 0)) return PG_ERR_INVALID;
  int64_t g = ceil_div<int64_t>(n, 256);
  hipLaunchKernelGGL(k_gather_labels_sc, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, as_stream(stream), ids, n,
                     labels, n_labels, fill, out, n_valid_out, scratch2);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_scatter_rows(const float* staged, const int32_t* pos, int64_t n, const int32_t* n_dev, int32_t dim,
                    float* out, int32_t out_stride, pg_stream_t stream) {
  return pg_scatter_rows_range(staged, pos, n, n_dev, dim, out, out_stride, 0, stream);
}

int pg_scatter_rows_range(const float* staged, const int32_t* pos, int64_t n, const int32_t* n_dev, int32_t dim,
                          float* out, int32_t out_stride, int32_t pos_lo, pg_stream_t stream) {
  return pg_scatter_rows_strided(staged, dim, pos, nullptr, n, n_dev, dim, out, out_stride, pos_lo, 0, stream);
}

int pg_scatter_rows_from_host(const float* table, int64_t table_stride, const int32_t* pos,
                              const int64_t* fullid, int64_t n_max, const int32_t* n_dev, int32_t dim,
                              float* out, int32_t out_stride, pg_stream_t stream) {
  return pg_scatter_rows_from_host_tail(table, table_stride, pos, fullid, n_max, n_dev, 0, dim, out, out_stride,
                                        stream);
}

int pg_scatter_rows_from_host_tail(const float* table, int64_t table_stride, const int32_t* pos,
                                   const int64_t* fullid, int64_t n_max, const int32_t* n_dev, int32_t start_num,
                                   int32_t dim, float* out, int32_t out_stride, pg_stream_t stream) {
  if (!pos) return PG_ERR_INVALID;
  return pg::scatter_host_tail(table, table_stride, pos, 0, fullid, n_max, n_dev, start_num, dim, out, out_stride, stream);
}

}  // extern "C"

// pos may be NULL (identity: a staged block in miss-list order); pos_lo as pg_scatter_rows_strided. Internal: the miss
// queue's device tail (pg_missq_device_tail).
int pg::scatter_host_tail(const float* table, int64_t table_stride, const int32_t* pos, int32_t pos_lo,
                          const int64_t* fullid, int64_t n_max, const int32_t* n_dev, int32_t start_num, int32_t dim,
                          float* out, int32_t out_stride, pg_stream_t stream) {
  if (n_max < 0 || dim <= 0 || out_stride < dim || table_stride < dim || start_num < 0 || start_num > 256 || pos_lo < 0)
    return PG_ERR_INVALID;
  if (n_max == 0) return PG_OK;
  if (!table || !fullid || !out) return PG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  // (round 3, measured inside the training loop with every miss row through this kernel — profiles/r03/zero_copy_grid.txt:
  // two-rows-per-wave kernel 8 / 16 / 24 / 48 / 96 / 192 blocks -> 0.372 / 0.231 / 0.217 / 0.250 / 0.279 / 0.288 ms/step;
  // wide kernel 4 / 8 / 12 / 16 / 24 / 48 -> 0.484 / 0.289 / 0.221 / 0.208 / 0.222 / 0.252)
  constexpr int kHostBlocks = 24, kWideBlocks = 16;
  const int grid = grid_1d(n_max, 8, kHostBlocks);
  if (dim % 4 == 0 && dim >= 256 && dim <= 1024 && out_stride % 4 == 0 && table_stride % 4 == 0 &&
      aligned(table, 16) && aligned(out, 16)) {
    const int gw = grid_1d(n_max, 4 * kRowsWide, kWideBlocks);
#define PG_WIDE(P)                                                                                                   \
  hipLaunchKernelGGL((k_scatter_host_wide<4, P>), dim3(gw), dim3(256), 0, st, table, table_stride, pos, fullid, n_max, \
                     n_dev, dim, out, out_stride, start_num, pos_lo)
    const int pieces = dim / 4;
    if (pieces <= 64) PG_WIDE(1);
    else if (pieces <= 128) PG_WIDE(2);
    else if (pieces <= 192) PG_WIDE(3);
    else PG_WIDE(4);
#undef PG_WIDE
    PG_LAUNCH_CHECK();
    return PG_OK;
  }
  if (dim % 4 == 0 && out_stride % 4 == 0 && table_stride % 4 == 0 && aligned(table, 16) && aligned(out, 16))
    hipLaunchKernelGGL(k_scatter_host<4>, dim3(grid), dim3(256), 0, st, table, table_stride, pos, fullid, n_max,
                       n_dev, dim, out, out_stride, start_num, pos_lo);
  else if (dim % 2 == 0 && out_stride % 2 == 0 && table_stride % 2 == 0 && aligned(table, 8) && aligned(out, 8))
    hipLaunchKernelGGL(k_scatter_host<2>, dim3(grid), dim3(256), 0, st, table, table_stride, pos, fullid, n_max,
                       n_dev, dim, out, out_stride, start_num, pos_lo);
  else
    hipLaunchKernelGGL(k_scatter_host<1>, dim3(grid), dim3(256), 0, st, table, table_stride, pos, fullid, n_max,
                       n_dev, dim, out, out_stride, start_num, pos_lo);
  PG_LAUNCH_CHECK();
  return PG_OK;
}
