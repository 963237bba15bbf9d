// Synthetic inputs generated directly in HBM (the GPU box has no dataset files):
//   * RMAT candidate edges — the role of the PaRMAT binary in README.md:36-41
//     (a,b,c = 0.45,0.22,0.22); quadrant choice per level from one 32-bit Philox
//     word against fixed-point thresholds, so host and device agree bit-exactly;
//   * U[0,1) fp32 features — PaGraph/data/preprocess.py:50-63 (random_feature)
//     and the torch.rand fallback of PaGraph/data/get_data.py:24-27.
#include "pg_common.h"

namespace pg {

__global__ __launch_bounds__(256) void k_rmat(uint32_t k0, uint32_t k1, int32_t scale, uint64_t ta, uint64_t tab,
                                              uint64_t tabc, int64_t first, int64_t n, int64_t* __restrict__ src,
                                              int64_t* __restrict__ dst) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t e = (uint64_t)(first + i);
    uint64_t s = 0, d = 0;
    uint32_t r[4];
    for (int lvl = 0; lvl < scale; ++lvl) {
      if ((lvl & 3) == 0) Philox::gen((uint32_t)e, (uint32_t)(e >> 32), (uint32_t)(lvl >> 2), 0x524D4154u, k0, k1, r);
      const uint64_t x = r[lvl & 3];
      const uint64_t sb = x >= tab ? 1u : 0u;                       // c or d quadrant: source bit set
      const uint64_t db = (x >= ta && x < tab) || x >= tabc ? 1u : 0u;  // b or d quadrant: destination bit set
      s = (s << 1) | sb;
      d = (d << 1) | db;
    }
    src[i] = (int64_t)s;
    dst[i] = (int64_t)d;
  }
}

__global__ __launch_bounds__(256) void k_features(uint32_t k0, uint32_t k1, int64_t row0, int64_t rows, int32_t dim,
                                                  float* __restrict__ out, int64_t stride) {
  const int groups = (dim + 3) / 4;
  const int64_t total = rows * groups;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rr = i / groups;
    const int g = (int)(i - rr * groups);
    const uint64_t row = (uint64_t)(row0 + rr);
    uint32_t r[4];
    Philox::gen((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)g, 0x46454154u, k0, k1, r);
    float* o = out + rr * stride + g * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (g * 4 + j < dim) o[j] = (float)(r[j] >> 8) * (1.0f / 16777216.0f);
  }
}

}  // namespace pg

using namespace pg;

extern "C" {

int pg_rmat_edges(uint64_t seed, int32_t scale, uint32_t a, uint32_t b, uint32_t c, int64_t first, int64_t n,
                  int64_t* src, int64_t* dst, pg_stream_t stream) {
  if (scale <= 0 || scale > 40 || n < 0 || first < 0) return PG_ERR_INVALID;
  if (n == 0) return PG_OK;
  if (!src || !dst) return PG_ERR_INVALID;
  const uint64_t ta = a, tab = (uint64_t)a + b, tabc = (uint64_t)a + b + c;
  if (tabc > 0xFFFFFFFFull) return PG_ERR_INVALID;
  int64_t g = ceil_div<int64_t>(n, 256);
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(k_rmat, dim3((unsigned)g), dim3(256), 0, as_stream(stream), (uint32_t)seed,
                     (uint32_t)(seed >> 32), scale, ta, tab, tabc, first, n, src, dst);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_random_features(uint64_t seed, int64_t row0, int64_t rows, int32_t dim, float* out, int64_t out_stride,
                       pg_stream_t stream) {
  if (rows < 0 || dim <= 0 || out_stride < dim || row0 < 0) return PG_ERR_INVALID;
  if (rows == 0) return PG_OK;
  if (!out) return PG_ERR_INVALID;
  int64_t g = ceil_div<int64_t>(rows * ((dim + 3) / 4), 256);
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(k_features, dim3((unsigned)g), dim3(256), 0, as_stream(stream), (uint32_t)seed,
                     (uint32_t)(seed >> 32), row0, rows, dim, out, out_stride);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // extern "C"
