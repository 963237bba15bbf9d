// Block aggregation (SpMM with an implicit 0/1 matrix) over a NodeFlow block in
// CSR, destination-major.  Replaces DGL's fused copy_src+mean / copy_src+sum
// message passing invoked by nf.block_compute at
//   PaGraph/model/gcn_nssc.py:71-74 (mean), :139-142 (sum),
//   PaGraph/model/graphsage_nssc.py:98-111 (mean / sum).
//
// HBM-bound: every edge reads one source row (4*dim bytes), every destination
// writes one row.  A group of LPR lanes (LPR = 64 for dim 600, 16 for dim 64)
// owns one destination row, so a wave covers 64/LPR rows; lanes stride the row
// in 16-byte pieces and keep up to 4 accumulators (one per piece) in VGPRs.
// The per-destination sum runs in edge order, which makes the forward
// deterministic (and bit-equal to the sequential oracle).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "pg_common.h"

namespace pg {

template <int VEC>
struct SV;
template <>
struct SV<4> {
  using type = float4;
  __device__ static inline float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ static inline void add(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
  __device__ static inline void div(float4& a, float d) { a.x /= d; a.y /= d; a.z /= d; a.w /= d; }
};
template <>
struct SV<2> {
  using type = float2;
  __device__ static inline float2 zero() { return make_float2(0.f, 0.f); }
  __device__ static inline void add(float2& a, const float2& b) { a.x += b.x; a.y += b.y; }
  __device__ static inline void div(float2& a, float d) { a.x /= d; a.y /= d; }
};
template <>
struct SV<1> {
  using type = float;
  __device__ static inline float zero() { return 0.f; }
  __device__ static inline void add(float& a, const float& b) { a += b; }
  __device__ static inline void div(float& a, float d) { a /= d; }
};

constexpr int kMaxAcc = 4;

template <int VEC>
__global__ __launch_bounds__(256) void k_spmm_fwd(const int32_t* __restrict__ indptr,
                                                  const int32_t* __restrict__ src,
                                                  const float* __restrict__ h, int32_t h_stride, int64_t n_dst,
                                                  int32_t dim, int reduce, float* __restrict__ out,
                                                  int32_t out_stride, int lpr_log2) {
  using S = SV<VEC>;
  using V = typename S::type;
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & (lpr - 1);
  const int rows_per_wave = kWave >> lpr_log2;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const int64_t v = wave * rows_per_wave + (lane >> lpr_log2);
  if (v >= n_dst) return;
  const int pieces = dim / VEC;
  const int32_t beg = indptr[v], end = indptr[v + 1];
  V* orow = reinterpret_cast<V*>(out + v * out_stride);
  for (int c0 = 0; c0 < pieces; c0 += lpr * kMaxAcc) {
    V acc[kMaxAcc];
#pragma unroll
    for (int m = 0; m < kMaxAcc; ++m) acc[m] = S::zero();
    for (int32_t e = beg; e < end; ++e) {
      const V* hrow = reinterpret_cast<const V*>(h + (int64_t)src[e] * h_stride);
#pragma unroll
      for (int m = 0; m < kMaxAcc; ++m) {
        const int c = c0 + m * lpr + gl;
        if (c < pieces) S::add(acc[m], hrow[c]);
      }
    }
    const float d = (float)(end - beg);
#pragma unroll
    for (int m = 0; m < kMaxAcc; ++m) {
      const int c = c0 + m * lpr + gl;
      if (c < pieces) {
        if (reduce == PG_REDUCE_MEAN && end > beg) S::div(acc[m], d);
        orow[c] = acc[m];
      }
    }
  }
}

// grad_h[src[e], c] += grad_out[v, c] * scale(v)
__global__ __launch_bounds__(256) void k_spmm_bwd(const int32_t* __restrict__ indptr,
                                                  const int32_t* __restrict__ src,
                                                  const float* __restrict__ go, int32_t go_stride, int64_t n_dst,
                                                  int32_t dim, int reduce, float* __restrict__ gh,
                                                  int32_t gh_stride, int lpr_log2) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & (lpr - 1);
  const int rows_per_wave = kWave >> lpr_log2;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const int64_t v = wave * rows_per_wave + (lane >> lpr_log2);
  if (v >= n_dst) return;
  const int32_t beg = indptr[v], end = indptr[v + 1];
  if (end == beg) return;
  const float d = (float)(end - beg);
  const float* grow = go + v * go_stride;
  for (int c = gl; c < dim; c += lpr) {
    float g = grow[c];
    if (reduce == PG_REDUCE_MEAN) g /= d;
    for (int32_t e = beg; e < end; ++e) unsafeAtomicAdd(gh + (int64_t)src[e] * gh_stride + c, g);
  }
}

static inline bool al(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

static inline int log2_ceil_pow2(int x) {
  int l = 0;
  while ((1 << l) < x) ++l;
  return l;
}

}  // namespace pg

using namespace pg;

extern "C" {

int pg_spmm_fwd(const int32_t* indptr, const int32_t* src, const float* h, int32_t h_stride, int64_t n_dst,
                int32_t dim, int reduce, float* out, int32_t out_stride, pg_stream_t stream) {
  if (n_dst < 0 || dim <= 0 || h_stride < dim || out_stride < dim) return PG_ERR_INVALID;
  if (reduce != PG_REDUCE_MEAN && reduce != PG_REDUCE_SUM) return PG_ERR_INVALID;
  if (n_dst == 0) return PG_OK;
  if (!indptr || !out) return PG_ERR_INVALID;  // src / h may be NULL only for an edgeless block
  hipStream_t st = as_stream(stream);
  int vec = 1;
  if (dim % 4 == 0 && h_stride % 4 == 0 && out_stride % 4 == 0 && al(h, 16) && al(out, 16)) vec = 4;
  else if (dim % 2 == 0 && h_stride % 2 == 0 && out_stride % 2 == 0 && al(h, 8) && al(out, 8)) vec = 2;
  const int pieces = dim / vec;
  int l2 = log2_ceil_pow2(pieces < 64 ? pieces : 64);
  const int rows_per_block = 4 * (64 >> l2);
  const unsigned grid = (unsigned)ceil_div<int64_t>(n_dst, rows_per_block);
  if (vec == 4)
    hipLaunchKernelGGL(k_spmm_fwd<4>, dim3(grid), dim3(256), 0, st, indptr, src, h, h_stride, n_dst, dim, reduce, out, out_stride, l2);
  else if (vec == 2)
    hipLaunchKernelGGL(k_spmm_fwd<2>, dim3(grid), dim3(256), 0, st, indptr, src, h, h_stride, n_dst, dim, reduce, out, out_stride, l2);
  else
    hipLaunchKernelGGL(k_spmm_fwd<1>, dim3(grid), dim3(256), 0, st, indptr, src, h, h_stride, n_dst, dim, reduce, out, out_stride, l2);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_spmm_bwd(const int32_t* indptr, const int32_t* src, const float* grad_out, int32_t go_stride, int64_t n_dst,
                int32_t dim, int reduce, float* grad_h, int32_t gh_stride, pg_stream_t stream) {
  if (n_dst < 0 || dim <= 0 || go_stride < dim || gh_stride < dim) return PG_ERR_INVALID;
  if (reduce != PG_REDUCE_MEAN && reduce != PG_REDUCE_SUM) return PG_ERR_INVALID;
  if (n_dst == 0) return PG_OK;
  if (!indptr || !grad_out) return PG_ERR_INVALID;  // src / grad_h may be NULL only for an edgeless block
  int l2 = log2_ceil_pow2(dim < 64 ? dim : 64);
  const int rows_per_block = 4 * (64 >> l2);
  const unsigned grid = (unsigned)ceil_div<int64_t>(n_dst, rows_per_block);
  hipLaunchKernelGGL(k_spmm_bwd, dim3(grid), dim3(256), 0, as_stream(stream), indptr, src, grad_out, go_stride,
                     n_dst, dim, reduce, grad_h, gh_stride, l2);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // extern "C"
