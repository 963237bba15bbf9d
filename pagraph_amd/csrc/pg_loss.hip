// Loss head of the training step: torch.nn.CrossEntropyLoss() on the seed layer's logits
// (examples/profile/pa_gcn.py:80 `loss_fcn = torch.nn.CrossEntropyLoss()`, :101-104
// `loss = loss_fcn(pred, batch_labels)`; same in pa_gs.py) = log-softmax + NLL, mean over the rows
// whose label is not ignore_index.
//
// In the replayed step the library path (softmax fwd, nll fwd, two fills, nll bwd, softmax bwd) is six
// launches of 7-17 us each on a 6000 x 60 matrix — 60 us of a 300 us step for 1.4 MB of data. Here the
// forward makes one pass: a wave owns a row (lane = class), computes the log-sum-exp, the row's loss and
// the UNSCALED gradient softmax(x) - onehot(label); a one-block kernel then reduces the row losses in a
// fixed order (deterministic) and stores {loss, 1 / #valid rows}. (A "last block reduces" ticket inside
// the first kernel was tried: 1500 agent-scope fences + same-address atomics cost 118 us.) The backward
// is a single scale by grad_out / #valid.
#include "pg_common.h"

namespace pg {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

__global__ __launch_bounds__(256) void k_xent_rows(const float* __restrict__ logits, int32_t stride,
                                                   const int64_t* __restrict__ labels, int64_t n, int32_t C,
                                                   int64_t ignore_index, float* __restrict__ dlogits,
                                                   int32_t d_stride, float* __restrict__ row_loss) {
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int64_t r = (int64_t)blockIdx.x * 4 + w;
  if (r >= n) return;
  const float* x = logits + r * stride;
  const int64_t lab = labels[r];
  const bool valid = lab != ignore_index && lab >= 0 && lab < C;
  float m = -INFINITY;
  for (int c = lane; c < C; c += kWave) m = fmaxf(m, x[c]);
  m = wave_max(m);
  float s = 0.f;
  for (int c = lane; c < C; c += kWave) s += expf(x[c] - m);
  s = wave_sum(s);
  const float lse = m + logf(s);
  if (dlogits) {
    float* d = dlogits + r * d_stride;
    for (int c = lane; c < C; c += kWave) d[c] = valid ? expf(x[c] - lse) - (c == lab ? 1.f : 0.f) : 0.f;
  }
  if (lane == 0) row_loss[r] = valid ? lse - x[lab] : 0.f;
}

// one block: fixed-order (deterministic) sum of the row losses -> meta = {mean loss, 1 / #counted rows}
__global__ __launch_bounds__(1024) void k_xent_reduce(const float* __restrict__ row_loss,
                                                      const int64_t* __restrict__ labels, int64_t n, int32_t C,
                                                      int64_t ignore_index, float* __restrict__ meta) {
  __shared__ float s_sum[1024];
  __shared__ int s_cnt[1024];
  float acc = 0.f;
  int cnt = 0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const int64_t lab = labels[i];
    if (lab != ignore_index && lab >= 0 && lab < C) {
      acc += row_loss[i];
      ++cnt;
    }
  }
  s_sum[threadIdx.x] = acc;
  s_cnt[threadIdx.x] = cnt;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
      s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int c = s_cnt[0];
    meta[0] = c > 0 ? s_sum[0] / (float)c : NAN;   // torch: mean over zero rows = nan
    meta[1] = c > 0 ? 1.f / (float)c : 0.f;
  }
}

__global__ __launch_bounds__(256) void k_xent_bwd(const float* __restrict__ dlogits, int32_t d_stride, int64_t n,
                                                  int32_t C, const float* __restrict__ meta,
                                                  const float* __restrict__ grad_out, float* __restrict__ gx,
                                                  int32_t gx_stride) {
  const float scale = meta[1] * (grad_out ? *grad_out : 1.f);
  const int64_t total = n * C;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / C;
    const int c = (int)(t - r * C);
    gx[r * gx_stride + c] = dlogits[r * d_stride + c] * scale;
  }
}

}  // namespace pg

using namespace pg;

extern "C" {

int pg_xent_fwd(const float* logits, int32_t stride, const int64_t* labels, int64_t n, int32_t C,
                int64_t ignore_index, float* dlogits, int32_t d_stride, float* row_loss, float* meta,
                pg_stream_t stream) {
  if (n <= 0 || C <= 0 || stride < C || (dlogits && d_stride < C)) return PG_ERR_INVALID;
  if (!logits || !labels || !row_loss || !meta) return PG_ERR_INVALID;
  hipLaunchKernelGGL(k_xent_rows, dim3((unsigned)ceil_div<int64_t>(n, 4)), dim3(256), 0, as_stream(stream), logits,
                     stride, labels, n, C, ignore_index, dlogits, d_stride, row_loss);
  PG_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_xent_reduce, dim3(1), dim3(1024), 0, as_stream(stream), row_loss, labels, n, C, ignore_index,
                     meta);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_xent_bwd(const float* dlogits, int32_t d_stride, int64_t n, int32_t C, const float* meta,
                const float* grad_out, float* gx, int32_t gx_stride, pg_stream_t stream) {
  if (n <= 0 || C <= 0 || d_stride < C || gx_stride < C) return PG_ERR_INVALID;
  if (!dlogits || !meta || !gx) return PG_ERR_INVALID;
  int64_t g = ceil_div<int64_t>(n * C, 256);
  hipLaunchKernelGGL(k_xent_bwd, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, as_stream(stream), dlogits,
                     d_stride, n, C, meta, grad_out, gx, gx_stride);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

}  // extern "C"
