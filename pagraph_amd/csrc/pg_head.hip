// Output head of the sampled GCN in one pass: the last NodeFlow block's aggregation (with the model's
// dropout), the output NodeUpdate (a [C, K] linear layer, no activation: gcn_nssc.py:58), the loss
// (torch.nn.CrossEntropyLoss, pa_gcn.py:80,101-104) and every gradient that hangs off them:
//     agg[v]   = mean_{e of v} dropout(h)[src(e)]                     (gcn_nssc.py:66-74)
//     z[v]     = W agg[v] + b                                         (gcn_nssc.py:18)
//     loss     = mean_{counted v} (logsumexp z[v] - z[v][label v])
//     dZ       = (softmax(z) - onehot) * grad_scale / #counted
//     dAgg     = dZ W,   dW = dZ^T agg,   db = sum dZ
// As separate kernels this chain is nine launches of 5-8 us each on 6000 x 64 numbers (aggregate, GEMM,
// three for the loss, weight gradient + its reduction, input-gradient GEMM): ~55 us of a ~230 us step.
// Here a wave owns a destination row at a time: lane = input column for the aggregation and dAgg, lane =
// class for z / softmax / dZ / the dW row; W lives in registers both row- and column-wise; agg and dZ
// cross between the two lane roles through LDS. Per-block partial sums of dW, db and the loss go to
// scratch and k_sum_partials (pg_dense.hip) adds them in block order: deterministic.
#include "pg_common.h"

namespace pg {

constexpr int kHeadMax = 64;   // K (input width) and C (classes) both fit one wave

struct HeadDrop {
  uint32_t thr, tag, k0, k1;
  const uint64_t* step;
  float scale;
};

// GraphSAGE's output NodeUpdate z = fc_neigh(agg) + fc_self(h_self) (graphsage_nssc.py:24; round 4): the destination's OWN row is
// a second operand that is neither aggregated nor dropped. It rides in the input columns [K, K + Ks) of the same wave —
// K + Ks <= 64 — with its own weight matrix and bias; its gradient leaves as dself. Ks == 0: the GCN head.
struct HeadSelf {
  const float* h;        // [n_dst, >= Ks]: row v = destination v's own input
  const float* W;        // [C, Ks]
  const float* bias;     // [C] or null
  float* dself;          // [n_dst, Ks]
  int32_t stride, Ks;
};

// value of lane R of this lane's quad (DPP quad_perm broadcast)
template <int R>
__device__ __forceinline__ uint32_t quad_bcast(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, R * 0x55, 0xF, 0xF, true);
}

__device__ __forceinline__ float hw_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ float hw_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

// RPW = destination rows per wave; the host picks the smallest that keeps the launch within one round of
// resident blocks (the per-block costs — staging W, reducing the partial sums — are amortised over 4 * RPW rows).
template <int kHeadRows>
__global__ __launch_bounds__(256) void k_gcn_head(const int32_t* __restrict__ indptr, const int32_t* __restrict__ src,
                                                  const float* __restrict__ h, int32_t h_stride, int32_t K,
                                                  const float* __restrict__ W, const float* __restrict__ bias,
                                                  int32_t C, const int64_t* __restrict__ labels,
                                                  int64_t ignore_index, const int32_t* __restrict__ n_valid_dev,
                                                  const float* __restrict__ grad_scale_dev, HeadDrop d, int reduce,
                                                  int64_t n_dst, float* __restrict__ logits,
                                                  float* __restrict__ dagg, float* __restrict__ part, int dagg_per_edge,
                                                  int row_len, const ProfSucc succ, const HeadSelf self, const Bnd bnd) {
  // PG_BOUNDS: [0] rows of h (a block edge is followed into the source layer's activations)
  prof_succ_stamp(succ);     // a profiled predecessor's "my successor started" stamp (pg_common.h)
  const int Ks = self.Ks, Kt = K + Ks;          // input columns: [0, K) aggregated, [K, Kt) the destination's own row
  __shared__ __attribute__((aligned(16))) float s_rows[4][kHeadRows][kHeadMax];   // the waves' aggregated rows
  __shared__ int s_lab[4][kHeadRows];
  __shared__ float s_deg[4][kHeadRows];      // what dAgg is divided by when it leaves per edge (dagg_per_edge)
  __shared__ __attribute__((aligned(16))) float s_dl[4][kHeadRows][kHeadMax];   // dZ of the wave's rows (lane = class)
  // W staged with coalesced loads, zero padded to 64 x 64, row stride 65 (conflict-free row AND column reads);
  // the block's partial sums laid out [k][class] so that the 64 lanes of a wave hit 64 banks
  __shared__ float s_w[kHeadMax * (kHeadMax + 1)];
  __shared__ __attribute__((aligned(16))) float s_big[2 * kHeadMax * kHeadMax + 2 * kHeadMax + 4];
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int64_t wave_g = (int64_t)blockIdx.x * 4 + w;
  const bool is_c = lane < C, is_k = lane < K, is_s = lane >= K && lane < Kt;
  // W row of class `lane` straight from global memory into registers (one round of 16-byte loads; 15 KB that every block
  // finds in L2), zero padded to 64 x 64; wave 0 parks its copy in LDS for the column reads of dAgg (the column stays in
  // LDS: three 64-register arrays per lane would leave one wave per SIMD). Round 3: this replaced zero-fill -> barrier ->
  // strided copy with a div/mod per element -> barrier -> 64 LDS reads per lane.
  float wrow[kHeadMax], accw[kHeadMax];
  if ((K & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0 &&
      (Ks == 0 || ((Ks & 3) == 0 && (reinterpret_cast<uintptr_t>(self.W) & 15) == 0))) {
#pragma unroll
    for (int k = 0; k < kHeadMax; k += 4) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (is_c && k < K) v = *reinterpret_cast<const float4*>(W + (int64_t)lane * K + k);
      else if (is_c && k < Kt) v = *reinterpret_cast<const float4*>(self.W + (int64_t)lane * Ks + (k - K));
      wrow[k] = v.x; wrow[k + 1] = v.y; wrow[k + 2] = v.z; wrow[k + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kHeadMax; ++k)
      wrow[k] = (is_c && k < K) ? W[(int64_t)lane * K + k] : ((is_c && k < Kt) ? self.W[(int64_t)lane * Ks + (k - K)] : 0.f);
  }
#pragma unroll
  for (int k = 0; k < kHeadMax; ++k) {
    if (w == 0) s_w[lane * (kHeadMax + 1) + k] = wrow[k];      // read after the barrier that closes phase 1
    accw[k] = 0.f;
  }
  const float bz = ((is_c && bias) ? bias[lane] : 0.f) + ((is_c && Ks && self.bias) ? self.bias[lane] : 0.f);
  const int nv = *n_valid_dev;
  const float inv = nv > 0 ? 1.f / (float)nv : 0.f;
  const float grad_scale = grad_scale_dev ? *grad_scale_dev : 1.f;
  const uint32_t step = (d.thr && d.step) ? (uint32_t)*d.step : 0u;
  const int piece = lane >> 2, j4 = lane & 3;
  const uint32_t q = (uint32_t)(((piece >> 7) << 6) | (piece & 63));
  const int half = (piece >> 6) & 1;
  // phase 1: aggregate the wave's rows. Software-pipelined by hand: all rows' indptr loads, then all rows'
  // first two source ids, then all rows' feature loads are issued together — three memory round trips for
  // the wave instead of three per row (a data-dependent edge loop per row serialises them: 21 us of 32).
  int64_t lab[kHeadRows];
  int32_t beg[kHeadRows], end[kHeadRows], s0[kHeadRows], s1[kHeadRows];
#pragma unroll
  for (int it = 0; it < kHeadRows; ++it) {
    const int64_t v = wave_g * kHeadRows + it;
    const bool live = v < n_dst;
    beg[it] = live ? indptr[v] : 0;
    end[it] = live ? indptr[v + 1] : 0;
    lab[it] = live ? labels[v] : ignore_index;
  }
#pragma unroll
  for (int it = 0; it < kHeadRows; ++it) {
    s0[it] = beg[it] < end[it] ? PG_IDX(src[beg[it]], bnd, 0, PG_K_HEAD, 1) : -1;
    s1[it] = beg[it] + 1 < end[it] ? PG_IDX(src[beg[it] + 1], bnd, 0, PG_K_HEAD, 2) : -1;
  }
  float x0[kHeadRows], x1[kHeadRows];
#pragma unroll
  for (int it = 0; it < kHeadRows; ++it) {
    x0[it] = (s0[it] >= 0 && is_k) ? h[(int64_t)s0[it] * h_stride + lane] : 0.f;
    x1[it] = (s1[it] >= 0 && is_k) ? h[(int64_t)s1[it] * h_stride + lane] : 0.f;
  }
  auto dropped = [&](int32_t sr, float x) {
    if (!d.thr) return x;
    uint32_t o[4];
    Philox::gen((uint32_t)sr, q, d.tag, step, d.k0, d.k1, o);
    const uint32_t wd = (j4 >> 1) ? (half ? o[3] : o[1]) : (half ? o[2] : o[0]);
    const uint32_t u = (j4 & 1) ? (wd >> 16) : (wd & 0xffffu);
    return u >= d.thr ? x * d.scale : 0.f;
  };
  // The mask of the rows' first two sources, one Philox call per FOUR source rows (round 3). Lane = column, so the four lanes
  // of a 16-byte piece all need the same call (counter = (source row, piece, tag, step)) and use 16 bits of it each: done per
  // lane, as `dropped` does, the wave runs 2 * kHeadRows identical-per-quad calls — ~640 cycles each, 4 us of every SIMD's
  // time in this kernel. Instead lane j of a quad draws for source j of a batch of four and the two words a piece needs
  // travel inside the quad (DPP quad_perm broadcasts, full rate). Same counters, same bits.
  bool keep0[kHeadRows], keep1[kHeadRows];
#pragma unroll
  for (int it = 0; it < kHeadRows; ++it) keep0[it] = keep1[it] = true;
  if (d.thr) {
    constexpr int NS = 2 * kHeadRows;
    auto source = [&](int i) { return i < NS ? ((i & 1) ? s1[i >> 1] : s0[i >> 1]) : -1; };
    auto decide = [&](uint32_t b0, uint32_t b1) {
      const uint32_t wd = (j4 >> 1) ? b1 : b0;
      const uint32_t u = (j4 & 1) ? (wd >> 16) : (wd & 0xffffu);
      return u >= d.thr;
    };
#pragma unroll
    for (int b = 0; b < NS; b += 4) {
      const int32_t c0 = source(b), c1 = source(b + 1), c2 = source(b + 2), c3 = source(b + 3);
      const int32_t mine = j4 == 0 ? c0 : (j4 == 1 ? c1 : (j4 == 2 ? c2 : c3));
      uint32_t o[4];
      Philox::gen((uint32_t)mine, q, d.tag, step, d.k0, d.k1, o);
      const uint32_t w0 = half ? o[2] : o[0], w1 = half ? o[3] : o[1];
      const bool k0 = decide(quad_bcast<0>(w0), quad_bcast<0>(w1));
      const bool k1 = decide(quad_bcast<1>(w0), quad_bcast<1>(w1));
      const bool k2 = decide(quad_bcast<2>(w0), quad_bcast<2>(w1));
      const bool k3 = decide(quad_bcast<3>(w0), quad_bcast<3>(w1));
      if (b < NS) keep0[b >> 1] = k0;
      if (b + 1 < NS) keep1[b >> 1] = k1;
      if (b + 2 < NS) keep0[(b + 2) >> 1] = k2;
      if (b + 3 < NS) keep1[(b + 2) >> 1] = k3;
    }
  }
#pragma unroll
  for (int it = 0; it < kHeadRows; ++it) {
    float a = 0.f;
    if (is_k) {
      if (s0[it] >= 0) a += d.thr ? (keep0[it] ? x0[it] * d.scale : 0.f) : x0[it];
      if (s1[it] >= 0) a += d.thr ? (keep1[it] ? x1[it] * d.scale : 0.f) : x1[it];
      for (int32_t e = beg[it] + 2; e < end[it]; ++e) {     // fan-out > 2: the rest of the row, one by one
        const int32_t sr = PG_IDX(src[e], bnd, 0, PG_K_HEAD, 3);
        a += dropped(sr, h[(int64_t)sr * h_stride + lane]);
      }
      if (reduce == PG_REDUCE_MEAN && end[it] > beg[it]) a /= (float)(end[it] - beg[it]);
    } else if (is_s) {
      const int64_t v = wave_g * kHeadRows + it;
      if (v < n_dst) a = self.h[v * self.stride + (lane - K)];      // the destination's own row: no mask, no reduce
    }
    s_rows[w][it][lane] = a;
    if (lane == 0) {
      const int64_t l = lab[it];
      s_lab[w][it] = (l != ignore_index && l >= 0 && l < C) ? (int)l : -1;
      s_deg[w][it] = (dagg_per_edge && reduce == PG_REDUCE_MEAN && end[it] > beg[it]) ? (float)(end[it] - beg[it]) : 0.f;
    }
  }
  __syncthreads();
  float accb = 0.f, lsum = 0.f;
  // Two passes over the wave's rows with ONE barrier between them (round 3; rounds 1-2 ran both halves per row with two
  // workgroup barriers each — eight per block): pass A (lane = class) computes z, the log-sum-exp, the loss and dZ and
  // parks dZ in LDS; pass B (lane = input column) reads it back for dAgg and accumulates the dW row. Neither pass rewrites
  // anything the other reads, so the rows inside a pass are independent and the compiler may overlap them.
#pragma unroll 2
  for (int it = 0; it < kHeadRows; ++it) {
    const int64_t v = wave_g * kHeadRows + it;
    const bool live = v < n_dst;
    const float* arow = s_rows[w][it];
    // z[class = lane]
    float z = bz;
#pragma unroll
    for (int k = 0; k < kHeadMax; k += 4) {
      const float4 g = *reinterpret_cast<const float4*>(arow + k);
      z += wrow[k] * g.x + wrow[k + 1] * g.y + wrow[k + 2] * g.z + wrow[k + 3] * g.w;
    }
    if (live && is_c && logits) logits[v * C + lane] = z;
    const float m = hw_max(is_c ? z : -INFINITY);
    const float ssum = hw_sum(is_c ? expf(z - m) : 0.f);
    const float lse = m + logf(ssum);
    const int lb = s_lab[w][it];
    const bool counted = live && lb >= 0;
    const float zl = __shfl(z, counted ? lb : 0, kWave);
    float dl = 0.f;
    if (counted && is_c) dl = (expf(z - lse) - (lane == lb ? 1.f : 0.f)) * (inv * grad_scale);
    if (counted && lane == 0) lsum += lse - zl;
    s_dl[w][it][lane] = dl;
    // the dW row of class `lane` (dl is this lane's own value: no LDS round trip)
#pragma unroll
    for (int k = 0; k < kHeadMax; k += 4) {
      const float4 g = *reinterpret_cast<const float4*>(arow + k);
      accw[k] += dl * g.x; accw[k + 1] += dl * g.y; accw[k + 2] += dl * g.z; accw[k + 3] += dl * g.w;
    }
    accb += dl;
  }
  __syncthreads();
#pragma unroll 2
  for (int it = 0; it < kHeadRows; ++it) {
    const int64_t v = wave_g * kHeadRows + it;
    const bool live = v < n_dst;
    // dAgg[input = lane]
    float gk = 0.f;
#pragma unroll
    for (int c = 0; c < kHeadMax; c += 4) {
      const float4 g = *reinterpret_cast<const float4*>(&s_dl[w][it][c]);
      const float* wc = s_w + c * (kHeadMax + 1) + lane;          // W[c .. c+3][lane]: consecutive banks
      gk += wc[0] * g.x + wc[kHeadMax + 1] * g.y + wc[2 * (kHeadMax + 1)] * g.z + wc[3 * (kHeadMax + 1)] * g.w;
    }
    // per edge: what every in-edge of v carries back under the mean — the backward aggregation then runs as a plain
    // sum and never loads the destinations' degrees (same division, same operands: bit-identical gradients)
    const float dg = s_deg[w][it];
    if (live && is_k) dagg[v * K + lane] = dg > 0.f ? gk / dg : gk;
    if (live && is_s) self.dself[v * Ks + (lane - K)] = gk;
  }
  __syncthreads();        // (the partial sums below reuse nothing of the above, but s_big aliases nothing: kept for clarity of phases)
  // block partial = (wave 0 + wave 2) + (wave 1 + wave 3), a fixed tree (deterministic): two LDS hand-offs instead of four
  // read-modify-write passes, and wave 0 stores the block's row of the partial layout straight from its registers —
  // lane = class owns W's row `lane`, K contiguous floats — instead of a transposing copy with a div/mod per element
  float* regA = s_big;                                   // [k][class]: the 64 lanes of a wave hit 64 banks
  float* regB = s_big + kHeadMax * kHeadMax;
  float* s_b = s_big + 2 * kHeadMax * kHeadMax;          // [2][class] bias sums, then [2] loss
  // no counted row at all: the mean over zero rows is nan, as torch's CrossEntropyLoss returns
  float lpart = nv > 0 ? lsum * inv : NAN;               // (lane 0's value is the wave's)
  if (w >= 2) {
    float* r = w == 2 ? regA : regB;
#pragma unroll
    for (int k = 0; k < kHeadMax; ++k) r[k * kHeadMax + lane] = accw[k];
    s_b[(w - 2) * kHeadMax + lane] = accb;
    if (lane == 0) s_b[2 * kHeadMax + (w - 2)] = lpart;
  }
  __syncthreads();
  if (w < 2) {
    const float* r = w == 0 ? regA : regB;
#pragma unroll
    for (int k = 0; k < kHeadMax; ++k) accw[k] += r[k * kHeadMax + lane];
    accb += s_b[w * kHeadMax + lane];
    lpart += s_b[2 * kHeadMax + w];
  }
  __syncthreads();
  if (w == 1) {
#pragma unroll
    for (int k = 0; k < kHeadMax; ++k) regA[k * kHeadMax + lane] = accw[k];
    s_b[lane] = accb;
    if (lane == 0) s_b[2 * kHeadMax] = lpart;
  }
  __syncthreads();
  if (w == 0) {
    float* mine = part + (int64_t)blockIdx.x * row_len;      // row_len % 4 == 0 (pg_gcn_head_row_len)
#pragma unroll
    for (int k = 0; k < kHeadMax; ++k) accw[k] += regA[k * kHeadMax + lane];
    accb += s_b[lane];
    lpart += s_b[2 * kHeadMax];
    if (is_c) {
      // partial layout of a block: [C x K] dW | [C x Ks] dW_self | [C] db | loss — each weight gradient contiguous, so that
      // the optimiser's launch can add the blocks' rows up per parameter
      float* row = mine + (int64_t)lane * K;
      float* srow = mine + (int64_t)C * K + (int64_t)lane * Ks;
      if ((K & 3) == 0 && (Ks & 3) == 0 && (row_len & 3) == 0 && (reinterpret_cast<uintptr_t>(part) & 15) == 0) {
#pragma unroll
        for (int k = 0; k < kHeadMax; k += 4) {
          if (k < K) *reinterpret_cast<float4*>(row + k) = make_float4(accw[k], accw[k + 1], accw[k + 2], accw[k + 3]);
          else if (k < Kt) *reinterpret_cast<float4*>(srow + (k - K)) = make_float4(accw[k], accw[k + 1], accw[k + 2], accw[k + 3]);
        }
      } else {
#pragma unroll
        for (int k = 0; k < kHeadMax; ++k) {
          if (k < K) row[k] = accw[k];
          else if (k < Kt) srow[k - K] = accw[k];
        }
      }
      mine[C * Kt + lane] = accb;
    }
    if (lane == 0) mine[C * Kt + C] = lpart;
  }
}

}  // namespace pg

using namespace pg;

static int head_rows(int64_t n_dst) {
  for (int r : {1, 2, 4}) if (ceil_div<int64_t>(n_dst, 4 * r) <= 512) return r;   // 2 blocks per CU resident (~200 VGPRs)
  return 8;
}

extern "C" {

/* floats per block of the partial layout: [C * K] dW, [C] db, [1] loss, padded to whole 16-byte pieces (the rows leave the
 * kernel as float4 stores) */
int32_t pg_gcn_head_row_len(int32_t K, int32_t C) {
  if (K <= 0 || C <= 0) return 0;
  return (C * K + C + 1 + 3) & ~3;
}

int64_t pg_gcn_head_scratch(int64_t n_dst, int32_t K, int32_t C) {
  if (n_dst <= 0 || K <= 0 || C <= 0) return 0;
  return ceil_div<int64_t>(n_dst, 4 * head_rows(n_dst)) * (int64_t)pg_gcn_head_row_len(K, C);
}

static int head_impl(const int32_t* indptr, const int32_t* src, const float* h, int32_t h_stride, int32_t K,
                     const float* W, const float* bias, int32_t C, const int64_t* labels, int64_t ignore_index,
                     const int32_t* n_valid_dev, const float* grad_scale_dev, const pg_dropout_t* drop, int reduce,
                     int64_t n_dst, float* logits, float* dagg, float* partials, float* dW, float* db_loss,
                     int32_t flags, const HeadSelf& self, pg_stream_t stream) {
  const int32_t sum_partials = flags & PG_HEAD_SUM_PARTIALS;
  const int32_t Kt = K + self.Ks;
  if (flags & ~(PG_HEAD_SUM_PARTIALS | PG_HEAD_DAGG_PER_EDGE)) return PG_ERR_INVALID;
  if (n_dst <= 0 || K <= 0 || C <= 0 || h_stride < K || self.Ks < 0) return PG_ERR_INVALID;
  if (self.Ks > 0 && (!self.h || !self.W || !self.dself || self.stride < self.Ks)) return PG_ERR_INVALID;
  if (Kt > kHeadMax || C > kHeadMax) return PG_ERR_UNSUPPORTED;
  if (reduce != PG_REDUCE_MEAN && reduce != PG_REDUCE_SUM) return PG_ERR_INVALID;
  if (!indptr || !h || !W || !labels || !n_valid_dev || !dagg || !partials || !dW || !db_loss) return PG_ERR_INVALID;
  if (drop && drop->threshold > 65535u) return PG_ERR_INVALID;
  HeadDrop d{};
  if (drop && drop->threshold) {
    d.thr = drop->threshold; d.tag = drop->tag;
    d.k0 = (uint32_t)drop->seed; d.k1 = (uint32_t)(drop->seed >> 32);
    d.step = drop->step;
    d.scale = 65536.f / (float)(65536u - drop->threshold);
  }
  const int rpw = head_rows(n_dst);
  const int64_t blocks = ceil_div<int64_t>(n_dst, 4 * rpw);
  hipStream_t st = as_stream(stream);
#define PG_HEAD(R)                                                                                                   \
  hipLaunchKernelGGL(k_gcn_head<R>, dim3((unsigned)blocks), dim3(256), 0, st, indptr, src, h, h_stride, K, W, bias, C, \
                     labels, ignore_index, n_valid_dev, grad_scale_dev, d, reduce, n_dst, logits, dagg, partials,          \
                     (flags & PG_HEAD_DAGG_PER_EDGE) ? 1 : 0, (int)pg_gcn_head_row_len(Kt, C), succ, self,                \
                     bnd(bounds_elems(h, 4) / h_stride))
  const ProfSucc succ = take_prof_succ();
  if (rpw == 1) PG_HEAD(1);
  else if (rpw == 2) PG_HEAD(2);
  else if (rpw == 4) PG_HEAD(4);
  else PG_HEAD(8);
#undef PG_HEAD
  PG_LAUNCH_CHECK();
  if (!sum_partials) return PG_OK;   // pg_adam_step adds the blocks' partials up
  // dW [C*K] (then dW_self [C*Ks]: the caller's dW buffer holds both, contiguous), then db [C] and the loss (db_loss[C])
  // contiguous behind them in the partial layout
  return pg_sum_partials_strided(partials, (int32_t)blocks, (int64_t)C * Kt, C + 1, pg_gcn_head_row_len(Kt, C), dW, db_loss,
                                 stream);
}

// ONE entry point for the output head (round 6: pg_gcn_head, pg_gcn_head_ex and pg_sage_head — 21 to 28 positional arguments —
// are gone): the descriptor's self_* fields select GraphSAGE's two-operand output layer (Ks > 0) or the GCN head (Ks == 0).
int pg_head(const pg_head_desc_t* d, pg_stream_t stream) {
  if (!d) return PG_ERR_INVALID;
  if (d->Ks < 0 || (d->Ks == 0 && (d->h_self || d->W_self || d->dself))) return PG_ERR_INVALID;
  HeadSelf self{};
  if (d->Ks > 0) self = HeadSelf{d->h_self, d->W_self, d->bias_self, d->dself, d->hs_stride, d->Ks};
  return head_impl(d->indptr, d->src, d->h, d->h_stride, d->K, d->W, d->bias, d->C, d->labels, d->ignore_index, d->n_valid_dev,
                   d->grad_scale_dev, d->has_drop ? &d->drop : nullptr, d->reduce, d->n_dst, d->logits, d->dagg, d->partials,
                   d->dW, d->db_loss, d->flags, self, stream);
}

}  // extern "C"
