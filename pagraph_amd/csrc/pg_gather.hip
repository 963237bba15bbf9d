// Feature-cache gather for MI355X (gfx950): the HBM-bound hot kernel.
//
// Replaces the op sequence of PaGraph/storage/storage.py:176-204 (fetch_data)
// and :207-216 (fetch_from_cache): per layer ~10 torch launches, 2 boolean-mask
// compactions (each a device->host sync) and a temp copy of the hit rows —
// with ONE launch over all rows of all layers.
//
// Work decomposition (wave64-native, no LDS, no block barrier):
//   * a wave owns RPW consecutive output rows; lane l < RPW loads ids[row0+l]
//     (one coalesced 8*RPW-byte read) and its slot_map entry;
//   * __ballot over "slot < 0" gives the wave's miss mask: one atomicAdd per
//     wave reserves a range in the miss list, each missing lane appends
//     (row, nid_map[id]) at its prefix-popcount rank (storage.py:179-182,117);
//   * rows are then copied U at a time: the row's slot is broadcast with
//     __shfl (wave-uniform), lanes stride over the row in 16-byte pieces
//     (dwordx4 when dim % 4 == 0, dwordx2 for F=602, dword otherwise), U
//     independent row loads in flight per lane before the first store.
//   * narrow fields (e.g. 'norm', dim 1) are copied lane-per-row: 64 gathered
//     loads, one coalesced store.
#include "pg_common.h"

namespace pg {

struct GatherArgs {
  const int64_t* ids;
  const int32_t* slot_map;
  const int64_t* nid_map;
  int32_t* miss_pos;
  int64_t* miss_fullid;
  int32_t* miss_count;
  int64_t n;
  int32_t n_fields;
  int32_t vec[PG_MAX_FIELDS];  // 4 / 2 / 1 floats per access; 0 = lane-per-row (narrow field)
  pg_field_t f[PG_MAX_FIELDS];
};

template <int VEC>
struct VecT;
template <>
struct VecT<4> { using type = float4; };
template <>
struct VecT<2> { using type = float2; };
template <>
struct VecT<1> { using type = float; };

// copy U rows (wave-uniform slots s[], output rows r[]) of one field, VEC floats per lane access
template <int VEC, int U>
__device__ __forceinline__ void copy_rows(const pg_field_t& fd, const int32_t (&s)[U],
                                          const int64_t (&r)[U], int lane) {
  using V = typename VecT<VEC>::type;
  const int pieces = fd.dim / VEC;
  const V* src[U];
  V* dst[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    src[u] = reinterpret_cast<const V*>(fd.cache + (int64_t)(s[u] < 0 ? 0 : s[u]) * fd.cache_stride);
    dst[u] = reinterpret_cast<V*>(fd.out + r[u] * fd.out_stride);
  }
  for (int c = lane; c < pieces; c += kWave) {
    V v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s[u] >= 0) v[u] = src[u][c];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s[u] >= 0) dst[u][c] = v[u];
  }
}

template <int RPW, int U, bool FULL>
__global__ __launch_bounds__(256) void k_gather(const GatherArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
  const int64_t row0 = wave * RPW;
  if (row0 >= a.n) return;

  // ---- stage: id -> slot, one row per lane ------------------------------
  const int64_t my_row = row0 + lane;
  const bool valid = lane < RPW && my_row < a.n;
  int64_t id = 0;
  int32_t slot = -2;  // -2: no row on this lane
  if (valid) {
    id = a.ids[my_row];
    slot = FULL ? (int32_t)id : a.slot_map[id];
  }
  if (!FULL) {
    // ---- hit/miss split (wave ballot + prefix popcount) -----------------
    const bool miss = valid && slot < 0;
    const unsigned long long mmask = __ballot(miss);
    if (mmask) {
      int32_t base = 0;
      if (lane == 0) base = atomicAdd(a.miss_count, (int32_t)__popcll(mmask));
      base = __shfl(base, 0);
      if (miss) {
        const int rank = __popcll(mmask & ((1ull << lane) - 1ull));
        a.miss_pos[base + rank] = (int32_t)my_row;
        a.miss_fullid[base + rank] = a.nid_map[id];
      }
    }
  }

  // ---- narrow fields: lane-per-row ---------------------------------------
  // (field loops are fully unrolled with constant indices so the by-value
  //  kernel-argument struct stays in SGPRs instead of being spilled to an alloca)
#pragma unroll
  for (int f = 0; f < PG_MAX_FIELDS; ++f) {
    if (f >= a.n_fields || a.vec[f] != 0) continue;
    const pg_field_t fd = a.f[f];
    if (valid && slot >= 0) {
      const float* src = fd.cache + (int64_t)slot * fd.cache_stride;
      float* dst = fd.out + my_row * fd.out_stride;
      for (int c = 0; c < fd.dim; ++c) dst[c] = src[c];
    }
  }

  // ---- wide fields: wave-per-row, U rows in flight ------------------------
  const int rows_here = (int)((a.n - row0) < RPW ? (a.n - row0) : RPW);
  for (int j = 0; j < rows_here; j += U) {
    int32_t s[U];
    int64_t r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int jj = j + u;
      // wave-uniform broadcast into an SGPR: the hit test below becomes a scalar branch
      s[u] = (jj < rows_here) ? __builtin_amdgcn_readlane(slot, jj) : -1;
      r[u] = row0 + jj;
    }
#pragma unroll
    for (int f = 0; f < PG_MAX_FIELDS; ++f) {
      if (f >= a.n_fields) continue;
      const int vec = a.vec[f];
      if (vec == 4) copy_rows<4, U>(a.f[f], s, r, lane);
      else if (vec == 2) copy_rows<2, U>(a.f[f], s, r, lane);
      else if (vec == 1) copy_rows<1, U>(a.f[f], s, r, lane);
    }
  }
}

// out[pos[j], :] = staged[j, :]  — wave per row (storage.py:199-200)
template <int VEC>
__global__ __launch_bounds__(256) void k_scatter(const float* __restrict__ staged,
                                                 const int32_t* __restrict__ pos, int64_t n,
                                                 const int32_t* __restrict__ n_dev, int32_t dim,
                                                 float* __restrict__ out, int32_t out_stride) {
  using V = typename VecT<VEC>::type;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t nn = n_dev ? (int64_t)*n_dev : n;
  const int64_t waves = (int64_t)gridDim.x * (blockDim.x / kWave);
  const int pieces = dim / VEC;
  for (int64_t j = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; j < nn; j += waves) {
    const V* src = reinterpret_cast<const V*>(staged + j * dim);
    V* dst = reinterpret_cast<V*>(out + (int64_t)pos[j] * out_stride);
    for (int c = lane; c < pieces; c += kWave) dst[c] = src[c];
  }
}

// zero-copy miss path: out[pos[j], :] = table_pinned[fullid[j], :] read over PCIe
template <int VEC>
__global__ __launch_bounds__(256) void k_scatter_host(const float* __restrict__ table,
                                                      int64_t table_stride,
                                                      const int32_t* __restrict__ pos,
                                                      const int64_t* __restrict__ fullid, int64_t n,
                                                      const int32_t* __restrict__ n_dev, int32_t dim,
                                                      float* __restrict__ out, int32_t out_stride) {
  using V = typename VecT<VEC>::type;
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t nn = n_dev ? (int64_t)*n_dev : n;
  const int64_t waves = (int64_t)gridDim.x * (blockDim.x / kWave);
  const int pieces = dim / VEC;
  for (int64_t j = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; j < nn; j += waves) {
    const V* src = reinterpret_cast<const V*>(table + fullid[j] * table_stride);
    V* dst = reinterpret_cast<V*>(out + (int64_t)pos[j] * out_stride);
    for (int c = lane; c < pieces; c += kWave) dst[c] = src[c];
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
static int launch_gather(GatherArgs& a, hipStream_t st) {
  // rows per wave: keep the grid >> 256 CUs at the real step shape (~42K rows)
  // and amortise the id->slot chain at large n.
  const int wpb = 4;  // waves per 256-thread block
  if (a.n >= (int64_t)1 << 19) {
    const int64_t blocks = ceil_div<int64_t>(a.n, 32 * wpb);
    hipLaunchKernelGGL((k_gather<32, 4, FULL>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  } else {
    const int64_t blocks = ceil_div<int64_t>(a.n, 8 * wpb);
    hipLaunchKernelGGL((k_gather<8, 4, FULL>), dim3((unsigned)blocks), dim3(256), 0, st, a);
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
                   const pg_field_t* fields, int n_fields, int32_t* miss_pos, int64_t* miss_fullid,
                   int32_t* miss_count, pg_stream_t stream) {
  if (n < 0 || n > INT32_MAX || !miss_count) return PG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  PG_HIP(hipMemsetAsync(miss_count, 0, sizeof(int32_t), st));
  if (n == 0) return PG_OK;
  if (!ids || !slot_map || !nid_map || !miss_pos || !miss_fullid) return PG_ERR_INVALID;
  GatherArgs a{};
  a.ids = ids; a.slot_map = slot_map; a.nid_map = nid_map;
  a.miss_pos = miss_pos; a.miss_fullid = miss_fullid; a.miss_count = miss_count;
  a.n = n;
  // a partially cached server may have an empty cache (cache == NULL): every row misses
  int rc = fill_args(a, fields, n_fields, false);
  if (rc != PG_OK) return rc;
  return launch_gather<false>(a, st);
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

int pg_scatter_rows(const float* staged, const int32_t* pos, int64_t n, const int32_t* n_dev, int32_t dim,
                    float* out, int32_t out_stride, pg_stream_t stream) {
  if (n < 0 || dim <= 0 || out_stride < dim) return PG_ERR_INVALID;
  if (n == 0) return PG_OK;
  if (!staged || !pos || !out) return PG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  const int grid = grid_1d(n, 4, 8192);
  if (dim % 4 == 0 && out_stride % 4 == 0 && aligned(staged, 16) && aligned(out, 16))
    hipLaunchKernelGGL(k_scatter<4>, dim3(grid), dim3(256), 0, st, staged, pos, n, n_dev, dim, out, out_stride);
  else if (dim % 2 == 0 && out_stride % 2 == 0 && aligned(staged, 8) && aligned(out, 8))
    hipLaunchKernelGGL(k_scatter<2>, dim3(grid), dim3(256), 0, st, staged, pos, n, n_dev, dim, out, out_stride);
  else
    hipLaunchKernelGGL(k_scatter<1>, dim3(grid), dim3(256), 0, st, staged, pos, n, n_dev, dim, out, out_stride);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_scatter_rows_from_host(const float* table, int64_t table_stride, const int32_t* pos,
                              const int64_t* fullid, int64_t n_max, const int32_t* n_dev, int32_t dim,
                              float* out, int32_t out_stride, pg_stream_t stream) {
  if (n_max < 0 || dim <= 0 || out_stride < dim || table_stride < dim) return PG_ERR_INVALID;
  if (n_max == 0) return PG_OK;
  if (!table || !pos || !fullid || !out) return PG_ERR_INVALID;
  hipStream_t st = as_stream(stream);
  const int grid = grid_1d(n_max, 4, 8192);
  if (dim % 4 == 0 && out_stride % 4 == 0 && table_stride % 4 == 0 && aligned(table, 16) && aligned(out, 16))
    hipLaunchKernelGGL(k_scatter_host<4>, dim3(grid), dim3(256), 0, st, table, table_stride, pos, fullid, n_max,
                       n_dev, dim, out, out_stride);
  else if (dim % 2 == 0 && out_stride % 2 == 0 && table_stride % 2 == 0 && aligned(table, 8) && aligned(out, 8))
    hipLaunchKernelGGL(k_scatter_host<2>, dim3(grid), dim3(256), 0, st, table, table_stride, pos, fullid, n_max,
                       n_dev, dim, out, out_stride);
  else
    hipLaunchKernelGGL(k_scatter_host<1>, dim3(grid), dim3(256), 0, st, table, table_stride, pos, fullid, n_max,
                       n_dev, dim, out, out_stride);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // extern "C"
