// Adam step of the trainer scripts (examples/profile/pa_gcn.py:137-139: torch.optim.Adam(model.parameters(),
// lr, weight_decay)) for the handful of small parameter tensors of the sampled GCN / GraphSAGE models
// (23 K - 40 K scalars in 4 - 8 tensors).  torch's capturable fused Adam spends two launches on it
// (per-tensor step counters += 1, then a multi-tensor kernel that gives each 65536-element chunk to ONE
// block: 4 blocks for these models, 15 us); here one launch spreads the elements over ~100 blocks and keeps
// ONE step counter on the device, bumped by the last block to finish (so that a replayed hipGraph
// advances it without the host).
#include <mutex>

#include "pg_common.h"

namespace pg {

struct AdamArgs {
  float* p[PG_ADAM_MAX_TENSORS];
  const float* g[PG_ADAM_MAX_TENSORS];
  float* m[PG_ADAM_MAX_TENSORS];
  float* v[PG_ADAM_MAX_TENSORS];
  int64_t end[PG_ADAM_MAX_TENSORS];   // exclusive prefix ends of the tensors' element ranges
  int32_t n_tensors;
  float lr, beta1, beta2, eps, weight_decay;
  int64_t* step;                      // device: completed steps
  uint32_t* ticket;                   // device, zero between launches
  int64_t* bump;                      // optional: one more device counter advanced with the step (the model's dropout step)
  int64_t* mirror;                    // optional, pinned host memory: the new step count, written by the last block
};

__global__ __launch_bounds__(256) void k_adam(const AdamArgs a) {
  const int64_t total = a.end[a.n_tensors - 1];
  const double t = (double)(*a.step + 1);
  const float bc1 = (float)(1.0 - pow((double)a.beta1, t));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)a.beta2, t));
  const float step_size = a.lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    // tensor of element i: constant-index selects only (a per-lane index into the by-value struct would
    // send it to scratch)
    float* pp = a.p[0];
    const float* gp = a.g[0];
    float *mp = a.m[0], *vp = a.v[0];
    int64_t base = 0;
#pragma unroll
    for (int j = 1; j < PG_ADAM_MAX_TENSORS; ++j) {
      if (j < a.n_tensors && i >= a.end[j - 1]) {
        pp = a.p[j]; gp = a.g[j]; mp = a.m[j]; vp = a.v[j];
        base = a.end[j - 1];
      }
    }
    const int64_t o = i - base;
    float g = gp[o];
    const float p = pp[o];
    if (a.weight_decay != 0.f) g += a.weight_decay * p;
    const float m = a.beta1 * mp[o] + (1.f - a.beta1) * g;
    const float v = a.beta2 * vp[o] + (1.f - a.beta2) * g * g;
    mp[o] = m;
    vp[o] = v;
    const float denom = sqrtf(v) / bc2_sqrt + a.eps;
    pp[o] = p - step_size * (m / denom);
  }
  // every block has read *step before the last one bumps it
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t done = atomicAdd(a.ticket, 1u) + 1;
    if (done == gridDim.x) {
      const int64_t ns = *a.step + 1;
      *a.step = ns;
      *a.ticket = 0;
      if (a.bump) *a.bump += 1;
      if (a.mirror) __hip_atomic_store(a.mirror, ns, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// The same step with the ordered partial sums of the weight-gradient kernels folded in (k_sum_partials x 2 + k_adam
// = 3 launches of 8-10 us each in the replayed GCN step -> 1): tensor i's gradient element o is either g[i][o] or
// the sum over `chunks[i]` partial rows of part[i][c * len[i] + off[i] + o], added in EXACTLY k_sum_partials' order
// (4 chunk groups x 8 rotating accumulators, pairwise combine) so that the trajectory is bit-identical to the
// three-launch version; the sum is also written to g[i] (p.grad stays meaningful). adam[i] == 0: reduce only (the
// loss scalar of pg_head lives in the same partial rows). Block = 64 elements x 4 chunk groups.
struct AdamPartArgs {
  float* p[PG_ADAM_MAX_TENSORS];
  float* g[PG_ADAM_MAX_TENSORS];
  float* m[PG_ADAM_MAX_TENSORS];
  float* v[PG_ADAM_MAX_TENSORS];
  const float* part[PG_ADAM_MAX_TENSORS];
  int64_t end[PG_ADAM_MAX_TENSORS];
  int32_t chunks[PG_ADAM_MAX_TENSORS];
  int32_t len[PG_ADAM_MAX_TENSORS];
  int32_t off[PG_ADAM_MAX_TENSORS];
  int32_t adam[PG_ADAM_MAX_TENSORS];
  // a parameter that is applied twice per step (GraphSAGE's NodeUpdate `lid` runs on every block >= lid,
  // graphsage_nssc.py:92-131) has a second set of partial rows: gradient = sum(part) + sum(part2), each in
  // k_sum_partials' order — the value AccumulateGrad's add of the two summed contributions gives
  const float* part2[PG_ADAM_MAX_TENSORS];
  int32_t chunks2[PG_ADAM_MAX_TENSORS];
  int32_t len2[PG_ADAM_MAX_TENSORS];
  int32_t off2[PG_ADAM_MAX_TENSORS];
  int32_t n_tensors;
  float lr, beta1, beta2, eps, weight_decay;
  int64_t* step;
  uint32_t* ticket;
  int64_t* bump;                        // optional: one more device counter advanced with the step (the model's dropout step)
  int64_t* mirror;                      // optional, pinned host memory: the new step count, written by the last block
};

__global__ __launch_bounds__(256) void k_adam_partials(const AdamPartArgs a) {
  __shared__ float red[4][64];
  const int64_t total = a.end[a.n_tensors - 1];
  // the bias corrections (two double-precision pow) once per block, while the partial rows are on their way
  __shared__ float s_bc[2];
  if (threadIdx.x == 255 && a.step) {
    const double t = (double)(*a.step + 1);
    s_bc[0] = (float)(1.0 - pow((double)a.beta1, t));
    s_bc[1] = (float)sqrt(1.0 - pow((double)a.beta2, t));
  }
  const int tx = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + tx;
  const bool ok = i < total;
  float* pp = a.p[0];
  float* gp = a.g[0];
  float *mp = a.m[0], *vp = a.v[0];
  const float* part = a.part[0];
  const float* part2 = a.part2[0];
  int chunks = a.chunks[0], len = a.len[0], off = a.off[0], is_adam = a.adam[0];
  int chunks2 = a.chunks2[0], len2 = a.len2[0], off2 = a.off2[0];
  int64_t base = 0;
#pragma unroll
  for (int j = 1; j < PG_ADAM_MAX_TENSORS; ++j) {
    if (j < a.n_tensors && i >= a.end[j - 1]) {
      pp = a.p[j]; gp = a.g[j]; mp = a.m[j]; vp = a.v[j];
      part = a.part[j]; chunks = a.chunks[j]; len = a.len[j]; off = a.off[j]; is_adam = a.adam[j];
      part2 = a.part2[j]; chunks2 = a.chunks2[j]; len2 = a.len2[j]; off2 = a.off2[j];
      base = a.end[j - 1];
    }
  }
  const int64_t o = i - base;
  // the parameter and its two moments depend on nothing but the element: their loads are issued here, ahead of the partial
  // rows' rounds and the two barriers of the ordered sum, instead of as one more round trip behind them
  float p_early = 0.f, m_early = 0.f, v_early = 0.f;
  if (grp == 0 && ok && is_adam) {
    p_early = pp[o];
    m_early = mp[o];
    v_early = vp[o];
  }
  // one ordered sum per set of partial rows (4 chunk groups x 8 rotating accumulators, pairwise combine)
  auto ordered_sum = [&](const float* pt, int nch, int rl, int of) -> float {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ok && pt) {
      const int per = (nch + 3) / 4;
      const int cb = grp * per, ce = (cb + per < nch) ? cb + per : nch;
      const float* col = pt + of + o;
      int c = cb;
      // 32 loads in flight per round (the head's partials are one row per block of pg_head: ~94 rows per chunk
      // group, twelve dependent rounds of 8 — the critical path of this launch); the additions keep k_sum_partials'
      // order: accumulator u & 7 takes rows c + u in ascending order
      for (; c + 31 < ce; c += 32) {
        float x[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) x[u] = col[(int64_t)(c + u) * rl];
#pragma unroll
        for (int u = 0; u < 32; ++u) acc[u & 7] += x[u];
      }
      for (; c + 7 < ce; c += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += col[(int64_t)(c + u) * rl];
      }
      for (; c < ce; ++c) acc[0] += col[(int64_t)c * rl];
    }
    __syncthreads();                                  // red is reused by the second set
    red[grp][tx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    return (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
  };
  const float g1 = ordered_sum(part, chunks, len, off);
  // (uniform per block only when no block straddles tensors with / without a second set: every thread takes the
  // call, the ones without a second set add nothing)
  const float g2 = ordered_sum(part2, chunks2, len2, off2);
  if (grp == 0 && ok) {
    float g;
    if (part) {
      g = part2 ? g1 + g2 : g1;
      gp[o] = g;
    } else {
      g = gp[o];
    }
    if (is_adam) {
      const float bc1 = s_bc[0], bc2_sqrt = s_bc[1];      // (ordered_sum's barriers lie between the write and this read)
      const float step_size = a.lr / bc1;
      const float p = p_early;
      if (a.weight_decay != 0.f) g += a.weight_decay * p;
      const float m = a.beta1 * m_early + (1.f - a.beta1) * g;
      const float v = a.beta2 * v_early + (1.f - a.beta2) * g * g;
      mp[o] = m;
      vp[o] = v;
      const float denom = sqrtf(v) / bc2_sqrt + a.eps;
      pp[o] = p - step_size * (m / denom);
    }
  }
  if (!a.step) return;                  // reduce only (uniform): the sums are in g[], nothing else moves
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t done = atomicAdd(a.ticket, 1u) + 1;
    if (done == gridDim.x) {
      const int64_t ns = *a.step + 1;
      *a.step = ns;
      *a.ticket = 0;
      if (a.bump) *a.bump += 1;
      if (a.mirror) __hip_atomic_store(a.mirror, ns, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace pg

using namespace pg;

// step counter -> host mirror (pg_adam_step_mirror): consulted at launch time, so the entry points keep their signatures
namespace {
struct MirrorReg {
  std::mutex m;
  int64_t* key[16] = {nullptr};
  int64_t* val[16] = {nullptr};
} g_mirror;
int64_t* mirror_of(int64_t* step_dev) {
  std::lock_guard<std::mutex> l(g_mirror.m);
  for (int i = 0; i < 16; ++i)
    if (g_mirror.key[i] == step_dev) return g_mirror.val[i];
  return nullptr;
}
}  // namespace

extern "C" int pg_adam_step_mirror(int64_t* step_dev, int64_t* mirror_host) {
  if (!step_dev) return PG_ERR_INVALID;
  std::lock_guard<std::mutex> l(g_mirror.m);
  int free_i = -1;
  for (int i = 0; i < 16; ++i) {
    if (g_mirror.key[i] == step_dev) {
      g_mirror.val[i] = mirror_host;
      if (!mirror_host) g_mirror.key[i] = nullptr;
      return PG_OK;
    }
    if (!g_mirror.key[i] && free_i < 0) free_i = i;
  }
  if (!mirror_host) return PG_OK;
  if (free_i < 0) return PG_ERR_NOMEM;
  g_mirror.key[free_i] = step_dev;
  g_mirror.val[free_i] = mirror_host;
  return PG_OK;
}

// ONE entry point (round 6; it used to be three with up to 24 positional arguments): the descriptor says per tensor
// whether the gradient is read as is or summed from one / two sets of partial rows, and per launch whether the update runs
// (PG_ADAM_FULL) or only the sums are written to grads[] (PG_ADAM_REDUCE_ONLY: the N > 1 step — the summed gradient goes
// straight into the flat buffer the all-reduce works on; a second, plain launch applies the update behind the collective).
extern "C" int pg_adam_step(const pg_adam_desc_t* d, pg_stream_t stream) {
  if (!d || d->n_tensors <= 0 || d->n_tensors > PG_ADAM_MAX_TENSORS) return PG_ERR_INVALID;
  if (d->mode != PG_ADAM_FULL && d->mode != PG_ADAM_REDUCE_ONLY) return PG_ERR_INVALID;
  const bool reduce_only = d->mode == PG_ADAM_REDUCE_ONLY;
  if (!reduce_only && (!d->step_dev || !d->ticket_dev)) return PG_ERR_INVALID;
  bool any_part = false;
  for (int i = 0; i < d->n_tensors; ++i) {
    const pg_adam_tensor_t& t = d->t[i];
    if (!t.grad || t.numel <= 0) return PG_ERR_INVALID;
    const bool upd = t.is_adam && !reduce_only;
    if (upd && (!t.param || !t.exp_avg || !t.exp_avg_sq)) return PG_ERR_INVALID;
    if (t.partials && (t.part_chunks <= 0 || t.part_off < 0 || (int64_t)t.part_off + t.numel > t.part_len)) return PG_ERR_INVALID;
    if (t.partials2 && (!t.partials || t.part2_chunks <= 0 || t.part2_off < 0 || (int64_t)t.part2_off + t.numel > t.part2_len))
      return PG_ERR_INVALID;
    if (reduce_only && !t.partials) return PG_ERR_INVALID;          // nothing to do for a tensor without partial rows
    any_part = any_part || t.partials != nullptr;
  }
  hipStream_t st = as_stream(stream);
  if (!any_part) {
    AdamArgs a{};
    int64_t tot = 0;
    for (int i = 0; i < d->n_tensors; ++i) {
      const pg_adam_tensor_t& t = d->t[i];
      if (!t.is_adam) return PG_ERR_INVALID;
      a.p[i] = t.param; a.g[i] = t.grad; a.m[i] = t.exp_avg; a.v[i] = t.exp_avg_sq;
      tot += t.numel;
      a.end[i] = tot;
    }
    a.n_tensors = d->n_tensors;
    a.lr = d->lr; a.beta1 = d->beta1; a.beta2 = d->beta2; a.eps = d->eps; a.weight_decay = d->weight_decay;
    a.step = d->step_dev; a.ticket = d->ticket_dev; a.bump = d->bump_dev; a.mirror = mirror_of(d->step_dev);
    int64_t g = ceil_div<int64_t>(tot, 256);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, st, a);
    PG_LAUNCH_CHECK();
    return PG_OK;
  }
  AdamPartArgs a{};
  int64_t tot = 0;
  for (int i = 0; i < d->n_tensors; ++i) {
    const pg_adam_tensor_t& t = d->t[i];
    a.p[i] = t.param; a.g[i] = t.grad; a.m[i] = t.exp_avg; a.v[i] = t.exp_avg_sq;
    a.part[i] = t.partials; a.chunks[i] = t.part_chunks; a.len[i] = t.part_len; a.off[i] = t.part_off;
    a.adam[i] = (t.is_adam && !reduce_only) ? 1 : 0;
    a.part2[i] = t.partials2; a.chunks2[i] = t.part2_chunks; a.len2[i] = t.part2_len; a.off2[i] = t.part2_off;
    tot += t.numel;
    a.end[i] = tot;
  }
  a.n_tensors = d->n_tensors;
  a.lr = d->lr; a.beta1 = d->beta1; a.beta2 = d->beta2; a.eps = d->eps; a.weight_decay = d->weight_decay;
  // reduce only: no step counter, no ticket, no bump, no mirror — the launch behind the collective advances them
  a.step = reduce_only ? nullptr : d->step_dev;
  a.ticket = reduce_only ? nullptr : d->ticket_dev;
  a.bump = reduce_only ? nullptr : d->bump_dev;
  a.mirror = reduce_only ? nullptr : mirror_of(d->step_dev);
  hipLaunchKernelGGL(k_adam_partials, dim3((unsigned)ceil_div<int64_t>(tot, 64)), dim3(256), 0, st, a);
  PG_LAUNCH_CHECK();
  return PG_OK;
}
