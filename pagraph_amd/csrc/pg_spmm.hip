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

#include <cstdlib>
#include <cstring>

#include "pg_common.h"
#include "pg_rows_w.h"

namespace pg {

template <>
struct SV<2> {
  using type = float2;
  __device__ static inline float2 zero() { return make_float2(0.f, 0.f); }
  __device__ static inline void add(float2& a, const float2& b) { a.x += b.x; a.y += b.y; }
  __device__ static inline void div(float2& a, float d) { a.x /= d; a.y /= d; }
  __device__ static inline float2 fill(float f) { return make_float2(f, f); }
  __device__ static inline void mx(float2& a, const float2& b) { a.x = fmaxf(a.x, b.x); a.y = fmaxf(a.y, b.y); }
  __device__ static inline float2 where_eq(const float2& x, const float2& o, const float2& b) {
    return make_float2(x.x == o.x ? b.x : 0.f, x.y == o.y ? b.y : 0.f);
  }
};
template <>
struct SV<1> {
  using type = float;
  __device__ static inline float zero() { return 0.f; }
  __device__ static inline void add(float& a, const float& b) { a += b; }
  __device__ static inline void div(float& a, float d) { a /= d; }
  __device__ static inline float fill(float f) { return f; }
  __device__ static inline void mx(float& a, const float& b) { a = fmaxf(a, b); }
  __device__ static inline float where_eq(const float& x, const float& o, const float& b) { return x == o ? b : 0.f; }
};

constexpr int kMaxAcc = 4;

// MAXR (PG_REDUCE_MAX, graphsage_nssc.py:106-110 'pool'): out[v] = element-wise maximum of v's in-edge messages in place of
// their sum; a destination without in-edges gets zeros like the other reducers.

template <int VEC, bool MAXR>
__global__ __launch_bounds__(256) void k_spmm_fwd(const int32_t* __restrict__ indptr,
                                                  const int32_t* __restrict__ src,
                                                  const float* __restrict__ h, int32_t h_stride, int64_t n_dst,
                                                  int32_t dim, int reduce, float* __restrict__ out,
                                                  int32_t out_stride, int lpr_log2, const Bnd bnd) {
  // PG_BOUNDS: [0] rows of h (a block edge is followed into the source layer)
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
    for (int m = 0; m < kMaxAcc; ++m) acc[m] = (MAXR && end > beg) ? S::fill(kNegInf) : S::zero();
    for (int32_t e = beg; e < end; ++e) {
      const V* hrow = reinterpret_cast<const V*>(h + (int64_t)PG_IDX(src[e], bnd, 0, PG_K_SPMM_FWD, 1) * h_stride);
#pragma unroll
      for (int m = 0; m < kMaxAcc; ++m) {
        const int c = c0 + m * lpr + gl;
        if (c < pieces) {
          if constexpr (MAXR) S::mx(acc[m], hrow[c]);
          else S::add(acc[m], hrow[c]);
        }
      }
    }
    const float d = (float)(end - beg);
#pragma unroll
    for (int m = 0; m < kMaxAcc; ++m) {
      const int c = c0 + m * lpr + gl;
      if (c < pieces) {
        if (!MAXR && reduce == PG_REDUCE_MEAN && end > beg) S::div(acc[m], d);
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

// ---- aggregation with the model's dropout folded in -------------------------------------------------
// GCNSampling / GraphSageSampling apply nn.Dropout to a layer's input right before aggregating it
// (gcn_nssc.py:66-69, graphsage_nssc.py:86-89).  As its own kernel that is a read + write of the whole
// [42K, 600] layer-0 frame plus a byte mask (25 us per step) for an input that needs no gradient, so the
// mask is generated where the rows are consumed instead.  Mask spec (restated in oracle/oracle.py
// dropout_mask): element (r, col) of the aggregated tensor, piece = col / 4,
//   q = (piece / 128) * 64 + piece % 64,  half = (piece / 64) % 2,  j = col % 4
//   w[0..3] = Philox4x32-10(counter = (r, q, tag, step), key = (seed_lo, seed_hi))
//   u16     = 16 bits of w[2 * half + j / 2], low half-word for even j, high for odd j
//   keep iff u16 >= threshold;  kept values are multiplied by scale = 65536 / (65536 - threshold).
// (pieces p and p + 64 share one Philox call: with 64 lanes on a row, lane l owns pieces l, l + 64, ...)
template <bool MAXR>
__global__ __launch_bounds__(256) void k_spmm_fwd_drop(const int32_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ src,
                                                       const float* __restrict__ h, int32_t h_stride, int64_t n_dst,
                                                       int32_t dim, int reduce, float* __restrict__ out,
                                                       int32_t out_stride, int lpr_log2, DropArgs d, const Bnd bnd) {
  using S = SV<4>;
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & (lpr - 1);
  const int rows_per_wave = kWave >> lpr_log2;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const int64_t v = wave * rows_per_wave + (lane >> lpr_log2);
  if (v >= n_dst) return;
  const uint32_t step = drop_step_of(d);
  const int pieces = dim / 4;
  const int32_t beg = indptr[v], end = indptr[v + 1];
  float4* orow = reinterpret_cast<float4*>(out + v * out_stride);
  for (int c0 = 0; c0 < pieces; c0 += lpr * kMaxAcc) {
    float4 acc[kMaxAcc];
#pragma unroll
    for (int m = 0; m < kMaxAcc; ++m) acc[m] = (MAXR && end > beg) ? S::fill(kNegInf) : S::zero();
    for (int32_t e = beg; e < end; ++e) {
      const int32_t sr = PG_IDX(src[e], bnd, 0, PG_K_SPMM_FWD, 2);
      const float4* hrow = reinterpret_cast<const float4*>(h + (int64_t)sr * h_stride);
      uint32_t o[4] = {0, 0, 0, 0};
      int have_q = -1;
#pragma unroll
      for (int m = 0; m < kMaxAcc; ++m) {
        const int c = c0 + m * lpr + gl;
        if (c < pieces) {
          const float4 x = hrow[c];
          const int q = ((c >> 7) << 6) | (c & 63);
          if (q != have_q) {
            Philox::gen((uint32_t)sr, (uint32_t)q, d.tag, step, d.k0, d.k1, o);
            have_q = q;
          }
          if constexpr (MAXR) S::mx(acc[m], drop_apply(x, o, (c >> 6) & 1, d.thr, d.scale));
          else S::add(acc[m], drop_apply(x, o, (c >> 6) & 1, d.thr, d.scale));
        }
      }
    }
    const float dg = (float)(end - beg);
#pragma unroll
    for (int m = 0; m < kMaxAcc; ++m) {
      const int c = c0 + m * lpr + gl;
      if (c < pieces) {
        if (!MAXR && reduce == PG_REDUCE_MEAN && end > beg) S::div(acc[m], dg);
        orow[c] = acc[m];
      }
    }
  }
}

// ---- aggregation straight from the feature cache (layer 0 is never materialised) ---------------
// out[v] = reduce over v's in-edges of dropout(row(src[e])), where row(p) is read where it lies:
//   slots[p] >= 0  : cache + slots[p] * cache_stride          (a hit: the HBM feature cache, storage.py:191-193)
//   slots[p] <= -3 : staged + (-slots[p] - 3) * staged_stride (a miss: row j of the block the miss path copied
//                                                              to the device, storage.py:196-200 without the scatter)
// Same per-destination summation order and the same dropout counters (row index = position p in the source
// layer) as pg_gather_rows + pg_spmm_fwd_drop, so the result is bit-identical to the unfused pair — minus one
// write and one read of the [|L0|, dim] frame (2 x 45 MB per step at the benchmark's shape).
// One wave per destination (dim >= 256): lane e of the wave looks up edge e's position and slot, the two
// dependent index loads of ALL of the destination's edges are in flight together, then the rows are streamed.
// prof (optional): the kernel's own ring of time stamps, entry (*drop.step or 0) % prof_ring (layout: pg_common.h,
// include/pagraph_hip.h) — a kernel inside a replayed hipGraph cannot carry HIP events.
constexpr int kRowsBatch = 4;   // source rows whose loads are in flight together per wave

// TAIL: dim % 4 != 0 (Reddit's 602). The rows are still read and written as 16-byte pieces — every row of the fused
// cache / of `out` is padded to a multiple of 4 floats (the host side checks the strides) — and the last piece's
// columns >= dim (the next field of the fused cache row, or padding) are forced to zero before they are summed, so
// `out`'s padding columns hold zeros.
template <bool DROP, bool TAIL, bool MAXR>
__global__ __launch_bounds__(256) void k_spmm_fwd_rows(const int32_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ src,
                                                       const int32_t* __restrict__ slots,
                                                       const int32_t* __restrict__ edge_slots,
                                                       const float* __restrict__ cache, int32_t cache_stride,
                                                       const float* __restrict__ staged, int32_t staged_stride,
                                                       int64_t n_dst, int32_t dim, int reduce,
                                                       float* __restrict__ out, int32_t out_stride, DropArgs d,
                                                       unsigned long long* __restrict__ prof, int prof_ring, Bnd bnd) {
  using S = SV<4>;
  const uint32_t step = drop_step_of(d);
  unsigned long long* pslot = prof_begin(prof, prof_ring, step, prof ? (unsigned long long)indptr[n_dst] : 0ull);
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t v = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  if (v < n_dst) {
    const int pieces = (dim + 3) / 4;
    const int tail = dim & 3;                        // valid columns of the last piece (TAIL only)
    const int32_t beg = indptr[v], end = indptr[v + 1];
    float4* orow = reinterpret_cast<float4*>(out + v * out_stride);
    for (int c0 = 0; c0 < pieces; c0 += kWave * kMaxAcc) {
      float4 acc[kMaxAcc];
      bool any = false;          // MAXR: a row was taken (padding / unresolved rows contribute nothing)
#pragma unroll
      for (int m = 0; m < kMaxAcc; ++m) acc[m] = MAXR ? S::fill(kNegInf) : S::zero();
      for (int32_t eb = beg; eb < end; eb += kWave) {
        const int ne = end - eb < kWave ? end - eb : kWave;
        // lane e holds edge e's source position and slot: the index loads of all of the destination's edges are in
        // flight together (edge_slots: slots[src[e]] composed per edge beforehand — one dependent load less)
        int32_t my_p = 0, my_s = -2;
        if (lane < ne) {
          my_p = PG_IDX(src[eb + lane], bnd, 0, PG_K_FWD_ROWS, 4);
          my_s = edge_slots ? edge_slots[eb + lane] : slots[my_p];
          my_s = bnd_slot(my_s, bnd, PG_K_FWD_ROWS, 5);
        }
        for (int e0 = 0; e0 < ne; e0 += kRowsBatch) {
          // the loads of up to kRowsBatch source rows are issued before the first is consumed; the accumulation
          // below still runs in edge order (bit-identical to one row at a time)
          float4 x[kRowsBatch][kMaxAcc];
          int32_t srp[kRowsBatch];
          bool ok[kRowsBatch];
#pragma unroll
          for (int j = 0; j < kRowsBatch; ++j) {
            const int e = e0 + j < ne ? e0 + j : ne - 1;
            const int32_t sl = __shfl(my_s, e);
            srp[j] = __shfl(my_p, e);
            ok[j] = e0 + j < ne && sl != -1 && sl != -2;      // padding / an unresolved miss contributes nothing
            const float4* hrow = sl >= 0 ? reinterpret_cast<const float4*>(cache + (int64_t)sl * cache_stride)
                                         : reinterpret_cast<const float4*>(staged + (int64_t)(-sl - 3) * staged_stride);
#pragma unroll
            for (int m = 0; m < kMaxAcc; ++m) {
              const int c = c0 + m * kWave + lane;
              x[j][m] = (ok[j] && c < pieces) ? hrow[c] : S::zero();
            }
          }
#pragma unroll
          for (int j = 0; j < kRowsBatch; ++j) {
            if (!ok[j]) continue;
            any = true;
            uint32_t o[4] = {0, 0, 0, 0};
            int have_q = -1;
#pragma unroll
            for (int m = 0; m < kMaxAcc; ++m) {
              const int c = c0 + m * kWave + lane;
              if (c < pieces) {
                float4 xv = x[j][m];
                if constexpr (DROP) {
                  const int q = ((c >> 7) << 6) | (c & 63);
                  if (q != have_q) {
                    Philox::gen((uint32_t)srp[j], (uint32_t)q, d.tag, step, d.k0, d.k1, o);
                    have_q = q;
                  }
                  xv = drop_apply(xv, o, (c >> 6) & 1, d.thr, d.scale);
                }
                if constexpr (TAIL) {
                  if (c == pieces - 1) {
                    if (tail < 2) xv.y = 0.f;
                    if (tail < 3) xv.z = 0.f;
                    xv.w = 0.f;
                  }
                }
                if constexpr (MAXR) S::mx(acc[m], xv);
                else S::add(acc[m], xv);
              }
            }
          }
        }
      }
      const float dg = (float)(end - beg);
#pragma unroll
      for (int m = 0; m < kMaxAcc; ++m) {
        const int c = c0 + m * kWave + lane;
        if (c < pieces) {
          if (!MAXR && reduce == PG_REDUCE_MEAN && end > beg) S::div(acc[m], dg);
          if (MAXR && !any) acc[m] = S::zero();
          orow[c] = acc[m];
        }
      }
    }
  }
  prof_end(pslot);
}

// k_spmm_fwd_rows_w<DROP, TAIL, M>: the same arithmetic (edge order, dropout counters: bit-identical) for rows of at
// most 64 * M pieces, laid out for the CU instead of for generality:
//  * the destination is wave-uniform (readfirstlane), so indptr[v], the loop bounds, every source row's slot / position
//    (readlane of the lane that looked them up) and hence the row's base address live in SGPRs: row loads are
//    `global_load_dwordx4 v, v_lane_offset, s[base]` with the piece offset as an immediate, and "is this row a hit, a
//    staged miss, padding" is a scalar branch;
//  * two source rows (the sampler's fan-out) x M pieces in flight per wave and M accumulators — no dead register slots
//    for pieces the row does not have — which is what lets 8 waves share a SIMD (the generic kernel: 126 VGPRs, 4 waves);
//  * a row's Philox draws are issued between the loads and their first use.
template <bool DROP, bool TAIL, int M, bool MAXR>
__global__ __launch_bounds__(256) void k_spmm_fwd_rows_w(const int32_t* __restrict__ indptr,
                                                         const int32_t* __restrict__ src,
                                                         const int32_t* __restrict__ slots,
                                                         const int32_t* __restrict__ edge_slots,
                                                         const float* __restrict__ cache, int32_t cache_stride,
                                                         const float* __restrict__ staged, int32_t staged_stride,
                                                         int64_t n_dst, int32_t dim, int reduce,
                                                         float* __restrict__ out, int32_t out_stride, DropArgs d,
                                                         unsigned long long* __restrict__ prof, int prof_ring,
                                                         int store_mode, Bnd bnd) {
  using S = SV<4>;
  const uint32_t step = drop_step_of(d);
  unsigned long long* pslot = prof_begin(prof, prof_ring, step, prof ? (unsigned long long)indptr[n_dst] : 0ull);
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t v = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  if (v < n_dst) {
    const int pieces = (dim + 3) / 4;
    const int tail = dim & 3;                        // valid columns of the last piece (TAIL only)
    const int32_t beg = indptr[v], end = indptr[v + 1];
    float4 acc[M];
    bool any = false;            // MAXR: a row was taken
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m] = MAXR ? S::fill(kNegInf) : S::zero();
    const RowsW rw{src, slots, edge_slots, cache, staged, cache_stride, staged_stride, bnd};
    rows_w_accumulate<DROP, TAIL, M, MAXR, false>(rw, beg, end, lane, pieces, tail, step, d, 0, 0, acc, any);
    const float dg = (float)(end - beg);
    float4* orow = reinterpret_cast<float4*>(out + v * out_stride);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int c = m * kWave + lane;
      if (c < pieces) {
        if (!MAXR && reduce == PG_REDUCE_MEAN && end > beg) S::div(acc[m], dg);
        if (MAXR && !any) acc[m] = S::zero();
        store_row_piece(orow + c, acc[m], store_mode);
      }
    }
  }
  prof_end(pslot);
}

// edge_slots[e] = slots[src[e]] for the block's edges (entries whose source position is out of range: -2)
__global__ __launch_bounds__(256) void k_compose_edge_slots(const int32_t* __restrict__ src, int64_t n_edges,
                                                            const int32_t* __restrict__ slots, int64_t n_src,
                                                            int32_t* __restrict__ edge_slots) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += (int64_t)gridDim.x * blockDim.x) {
    const int32_t p = src[e];
    edge_slots[e] = (p >= 0 && p < n_src) ? slots[p] : -2;
  }
}

// grad_h[src[e], c] += grad_out[v, c] * scale(v) * mask(src[e], c)
__global__ __launch_bounds__(256) void k_spmm_bwd_drop(const int32_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ src,
                                                       const float* __restrict__ go, int32_t go_stride, int64_t n_dst,
                                                       int32_t dim, int reduce, float* __restrict__ gh,
                                                       int32_t gh_stride, int lpr_log2, DropArgs d) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & (lpr - 1);
  const int rows_per_wave = kWave >> lpr_log2;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const int64_t v = wave * rows_per_wave + (lane >> lpr_log2);
  if (v >= n_dst) return;
  const int32_t beg = indptr[v], end = indptr[v + 1];
  if (end == beg) return;
  const uint32_t step = drop_step_of(d);
  const float dg = (float)(end - beg);
  const float* grow = go + v * go_stride;
  for (int c = gl; c < dim; c += lpr) {
    float g = grow[c];
    if (reduce == PG_REDUCE_MEAN) g /= dg;
    g *= d.scale;
    const int piece = c >> 2, j = c & 3;
    const int q = ((piece >> 7) << 6) | (piece & 63), half = (piece >> 6) & 1;
    for (int32_t e = beg; e < end; ++e) {
      const int32_t sr = src[e];
      uint32_t o[4];
      Philox::gen((uint32_t)sr, (uint32_t)q, d.tag, step, d.k0, d.k1, o);
      const uint32_t w = (j >> 1) ? (half ? o[3] : o[1]) : (half ? o[2] : o[0]);
      const uint32_t u = (j & 1) ? (w >> 16) : (w & 0xffffu);
      if (u >= d.thr) unsafeAtomicAdd(gh + (int64_t)sr * gh_stride + c, g);
    }
  }
}

// Backward aggregation in gather form over the block's source-major copy (pg_nodeflow_desc_t.blk_tptr /
// blk_tdst, built by the sampler off the critical path):
//   grad_h[s, :] = mask(s, :) * scale * sum_{t in [tptr[s], tptr[s+1])} grad_out[tdst[t], :] / deg(tdst[t])
// Every row of grad_h is written (no zero fill), nothing is atomic (the scatter form manages ~30 G fp32
// atomics/s: 26 us for the 12K-edge output block) and the sum runs in ascending destination order.
// Optional second output of the backward aggregation: when grad_h is the gradient of a NodeUpdate's skip-concat output
// y = [z | relu(z)] (gcn_nssc.py:20-21, 2N columns), dZ[s, j] = grad_h[s, j] + (y[s, j] > 0 ? grad_h[s, N + j] : 0) is what
// the layer's weight gradient needs (k_dz of pg_dense.hip: one launch of the replayed step). The lane that holds piece c of a
// row gets piece c + N/4 from its neighbour in the row's lane group and writes dZ next to grad_h.
struct DzOut {
  const float* y;   // the activation output the aggregation consumed (its forward input h)
  float* dz;        // [n_src, N]
  int32_t y_stride, N;
};

__device__ __forceinline__ float4 dz_piece(float4 lo, float4 hi, float4 y) {
  lo.x += y.x > 0.f ? hi.x : 0.f;
  lo.y += y.y > 0.f ? hi.y : 0.f;
  lo.z += y.z > 0.f ? hi.z : 0.f;
  lo.w += y.w > 0.f ? hi.w : 0.f;
  return lo;
}

// PG_REDUCE_MAX backward (pg_spmm_bwd (max, gather form)): the forward's input and output
struct MaxIn {
  const float* h;     // [n_src, dim] the aggregation's input (before dropout)
  const float* out;   // [n_dst, dim] its output
  int32_t h_stride, out_stride;
};

// piece c of row sr through the dropout mask: x * scale where kept, 0 where dropped (the forward's drop_apply for a
// float4 piece; for the scalar layout c is the column)
template <int VEC>
__device__ __forceinline__ typename SV<VEC>::type drop_piece(typename SV<VEC>::type x, uint32_t sr, int c, const DropArgs& d,
                                                             uint32_t step) {
  if constexpr (VEC == 4) {
    uint32_t o[4];
    Philox::gen(sr, (uint32_t)(((c >> 7) << 6) | (c & 63)), d.tag, step, d.k0, d.k1, o);
    return drop_apply(x, o, (c >> 6) & 1, d.thr, d.scale);
  } else {
    const int piece = c >> 2, j = c & 3, half = (piece >> 6) & 1;
    uint32_t o[4];
    Philox::gen(sr, (uint32_t)(((piece >> 7) << 6) | (piece & 63)), d.tag, step, d.k0, d.k1, o);
    const uint32_t w = (j >> 1) ? (half ? o[3] : o[1]) : (half ? o[2] : o[0]);
    const uint32_t u = (j & 1) ? (w >> 16) : (w & 0xffffu);
    return u >= d.thr ? x * d.scale : 0.f;
  }
}

template <int VEC, bool DROP, int T, bool MAXR>
__device__ __forceinline__ void heavy_rows(const int32_t* __restrict__ heavy, int32_t heavy_cap,
                                           const int32_t* __restrict__ tptr, const int32_t* __restrict__ tdst,
                                           const int32_t* __restrict__ indptr, const float* __restrict__ go,
                                           int32_t go_stride, int32_t dim, int reduce, float* __restrict__ gh,
                                           int32_t gh_stride, int lpr_log2, DropArgs d, int first, int stride, DzOut z,
                                           MaxIn mi, int parts, const Bnd& bnd);

constexpr int kBwdBatch = 4;

// (blocks >= n_row_blocks of the launch are the hub blocks: heavy_rows below)
template <int VEC, bool DROP, bool MAXR>
__global__ __launch_bounds__(256) void k_spmm_bwd_gather(const int32_t* __restrict__ tptr,
                                                         const int32_t* __restrict__ tdst,
                                                         const int32_t* __restrict__ indptr,
                                                         const float* __restrict__ go, int32_t go_stride,
                                                         int64_t n_src, int32_t dim, int reduce,
                                                         float* __restrict__ gh, int32_t gh_stride, int lpr_log2,
                                                         int skip_heavy, DropArgs d, const int32_t* __restrict__ heavy,
                                                         int32_t heavy_cap, int32_t n_row_blocks, DzOut z, MaxIn mi,
                                                         int hub_parts, const Bnd bnd) {
  // PG_BOUNDS: [0] destination rows of grad_out (an entry of the source-major copy is followed into them), [1] source rows
  using S = SV<VEC>;
  using V = typename S::type;
  if ((int)blockIdx.x >= n_row_blocks) {
    heavy_rows<VEC, DROP, 256, MAXR>(heavy, heavy_cap, tptr, tdst, indptr, go, go_stride, dim, reduce, gh, gh_stride,
                                     lpr_log2, d, (int)blockIdx.x - n_row_blocks, (int)gridDim.x - n_row_blocks, z, mi,
                                     hub_parts, bnd);
    return;
  }
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & (lpr - 1);
  const int rows_per_wave = kWave >> lpr_log2;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const int64_t sr = wave * rows_per_wave + (lane >> lpr_log2);
  if (sr >= n_src) return;
  const uint32_t step = DROP ? drop_step_of(d) : 0u;
  const int pieces = dim / VEC;
  const int32_t beg = tptr[sr], end = tptr[sr + 1];
  if (skip_heavy && end - beg > PG_HEAVY_ROW) return;   // a hub: k_spmm_bwd_heavy gives it a whole block
  V* grow = reinterpret_cast<V*>(gh + sr * gh_stride);
  V last = S::zero();       // this lane's piece of the row (the dZ epilogue needs it; pieces <= lpr there)
  // the dZ epilogue's own operand (the layer's saved output) depends on nothing but the row: its load is issued here, ahead
  // of the tptr -> tdst -> gradient-rows chain, instead of as one more dependent round trip behind it
  float4 yv_early = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (VEC == 4) {
    if (z.dz && gl < z.N / 4) yv_early = reinterpret_cast<const float4*>(z.y + sr * z.y_stride)[gl];
  }
  for (int c = gl; c < pieces; c += lpr) {
    V acc = S::zero();
    V xd = S::zero();          // MAXR: this source's message as the forward saw it
    if constexpr (MAXR) {
      if (end > beg) {
        xd = reinterpret_cast<const V*>(mi.h + sr * mi.h_stride)[c];
        if constexpr (DROP) xd = drop_piece<VEC>(xd, (uint32_t)sr, c, d, step);
      }
    }
    // kBwdBatch edges' loads in flight together (destination ids, then their rows and degrees), added in ascending
    // edge order as a plain loop would: a 30-edge row costs 8 x 2 memory round trips, not 30 x 2
    for (int32_t t = beg; t < end; t += kBwdBatch) {
      int32_t v[kBwdBatch];
#pragma unroll
      for (int u = 0; u < kBwdBatch; ++u) v[u] = t + u < end ? PG_IDX(tdst[t + u], bnd, 0, PG_K_BWD_GATHER, 1) : -1;
      V g[kBwdBatch];
      V ov[MAXR ? kBwdBatch : 1];
      float dg[kBwdBatch];
#pragma unroll
      for (int u = 0; u < kBwdBatch; ++u) {
        g[u] = v[u] >= 0 ? reinterpret_cast<const V*>(go + (int64_t)v[u] * go_stride)[c] : S::zero();
        if constexpr (MAXR) {
          ov[u] = v[u] >= 0 ? reinterpret_cast<const V*>(mi.out + (int64_t)v[u] * mi.out_stride)[c] : S::zero();
          dg[u] = 1.f;
        } else {
          dg[u] = (reduce == PG_REDUCE_MEAN && v[u] >= 0) ? (float)(indptr[v[u] + 1] - indptr[v[u]]) : 1.f;
        }
      }
#pragma unroll
      for (int u = 0; u < kBwdBatch; ++u) {
        if (v[u] >= 0) {
          if constexpr (MAXR) {
            S::add(acc, S::where_eq(xd, ov[u], g[u]));
          } else {
            if (reduce == PG_REDUCE_MEAN) S::div(g[u], dg[u]);
            S::add(acc, g[u]);
          }
        }
      }
    }
    if constexpr (DROP) {
      if (end > beg) acc = drop_piece<VEC>(acc, (uint32_t)sr, c, d, step);
    }
    grow[c] = acc;
    last = acc;
  }
  if constexpr (VEC == 4) {
    if (z.dz) {   // host side guarantees pieces <= lpr, dim == 2 N: piece c pairs with piece c + N/4 of the same row
      const int hp = z.N / 4;
      float4 hi;
      hi.x = __shfl_down(last.x, hp, lpr);
      hi.y = __shfl_down(last.y, hp, lpr);
      hi.z = __shfl_down(last.z, hp, lpr);
      hi.w = __shfl_down(last.w, hp, lpr);
      if (gl < hp) reinterpret_cast<float4*>(z.dz + sr * z.N)[gl] = dz_piece(last, hi, yv_early);
    }
  }
}

// hubs (sources with more than PG_HEAVY_ROW edges, listed by the sampler): a whole block per hub.
// Per chunk of T edges: (destination, degree) staged in LDS with coalesced loads, then T / lpr edge lanes accumulate
// strided edges — independent loads, 16 in flight per lane — and the partial sums are combined through LDS in lane
// order (deterministic). Runs as EXTRA BLOCKS of the k_spmm_bwd_gather launch (T = 256): two hub rows used to cost a
// launch of their own, 13-14 us of dependent latencies on the replayed step's critical path.
template <int VEC, bool DROP, int T, bool MAXR>
__device__ __forceinline__ void heavy_rows(const int32_t* __restrict__ heavy, int32_t heavy_cap,
                                           const int32_t* __restrict__ tptr, const int32_t* __restrict__ tdst,
                                           const int32_t* __restrict__ indptr, const float* __restrict__ go,
                                           int32_t go_stride, int32_t dim, int reduce, float* __restrict__ gh,
                                           int32_t gh_stride, int lpr_log2, DropArgs d, int first, int stride, DzOut z,
                                           MaxIn mi, int parts, const Bnd& bnd) {
  constexpr int kHeavyThreads = T;

  using S = SV<VEC>;
  using V = typename S::type;
  constexpr int kStage = 1024;              // edges staged per round, whatever the block size
  __shared__ int32_t s_v[kStage];
  __shared__ float s_w[kStage];
  __shared__ V red[kHeavyThreads];
  __shared__ float4 zrow[64];               // the finished row, for the dZ epilogue
  // Round 3: a hub's COLUMNS are split over `parts` blocks (extra block id = hub slot * parts + part). One block per hub
  // had 256 / lpr edge lanes with 16 loads in flight each: a hub of ~870 edges (the vertex every isolated seed aliases to)
  // took four dependent rounds of loads, and the two hub blocks set the duration of the whole launch (13-16 us; the
  // ~9.5 K regular rows need about half). With the row's pieces dealt to `parts` blocks each block has parts times the
  // edge lanes — one round for up to 1024 edges — and nothing has to be combined across blocks. Part p takes pieces
  // [p q, (p + 1) q) and their partners hp + the same, q = hp / parts, hp = pieces / 2, so that the skip-concat's dZ
  // (which pairs piece c with c + hp) stays inside a block. The mapping does not depend on whether dZ is asked for.
  const int part = first % parts;
  first /= parts;
  stride /= parts;
  int n_heavy = heavy[0];
  int32_t sr_next = first < heavy_cap ? PG_IDX(heavy[1 + first], bnd, 1, PG_K_BWD_GATHER, 3) : 0;   // fetched with the count, not after it
  if (n_heavy > heavy_cap) n_heavy = heavy_cap;
  int lp_log2 = lpr_log2;                         // log2 of the lanes across THIS block's pieces
  for (int p = parts; p > 1; p >>= 1) --lp_log2;
  const int lpr = 1 << lp_log2;
  const int el = threadIdx.x >> lp_log2, n_el = kHeavyThreads >> lp_log2, gl = threadIdx.x & (lpr - 1);
  const int pieces = dim / VEC;
  const int hp = pieces / 2, qn = parts > 1 ? hp / parts : 0;
  const uint32_t step = DROP ? drop_step_of(d) : 0u;
  for (int hi = first; hi < n_heavy; hi += stride) {
    const int sr = sr_next;
    if (hi + stride < n_heavy) sr_next = PG_IDX(heavy[1 + hi + stride], bnd, 1, PG_K_BWD_GATHER, 4);
    const int32_t beg = tptr[sr], end = tptr[sr + 1];
    for (int c0 = 0; c0 < (parts > 1 ? 1 : pieces); c0 += lpr) {
      // (parts > 1: one pass, this lane's piece comes from the part's two runs of qn pieces)
      const int c = parts > 1 ? (gl < qn ? part * qn + gl : hp + part * qn + (gl - qn)) : c0 + gl;
      V acc = S::zero();
      V xd = S::zero();          // MAXR: this source's message as the forward saw it
      if constexpr (MAXR) {
        if (c < pieces) {
          xd = reinterpret_cast<const V*>(mi.h + (int64_t)sr * mi.h_stride)[c];
          if constexpr (DROP) xd = drop_piece<VEC>(xd, (uint32_t)sr, c, d, step);
        }
      }
      for (int32_t base = beg; base < end; base += kStage) {
        const int n = end - base < kStage ? end - base : kStage;
        __syncthreads();
        for (int t = threadIdx.x; t < n; t += kHeavyThreads) {     // independent per t: all in flight together
          const int32_t v = PG_IDX(tdst[base + t], bnd, 0, PG_K_BWD_GATHER, 2);
          s_v[t] = v;
          s_w[t] = (!MAXR && reduce == PG_REDUCE_MEAN) ? (float)(indptr[v + 1] - indptr[v]) : 1.f;
        }
        __syncthreads();
        if (c < pieces) {
          // 16 rows' loads in flight per lane, then added in the same (ascending k) order as a plain loop: a hub's
          // few hundred edges cost one or two memory round trips per lane instead of one per four edges
          for (int k0 = el; k0 < n; k0 += 16 * n_el) {
            V g[16];
            V ov[MAXR ? 16 : 1];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int k = k0 + u * n_el;
              g[u] = k < n ? reinterpret_cast<const V*>(go + (int64_t)s_v[k] * go_stride)[c] : S::zero();
              if constexpr (MAXR) ov[u] = k < n ? reinterpret_cast<const V*>(mi.out + (int64_t)s_v[k] * mi.out_stride)[c] : S::zero();
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
              const int k = k0 + u * n_el;
              if (k < n) {
                if constexpr (MAXR) {
                  S::add(acc, S::where_eq(xd, ov[u], g[u]));
                } else {
                  if (reduce == PG_REDUCE_MEAN) S::div(g[u], s_w[k]);
                  S::add(acc, g[u]);
                }
              }
            }
          }
        }
      }
      red[threadIdx.x] = acc;
      __syncthreads();
      if (el == 0 && c < pieces) {
        V tot = S::zero();
        for (int k = 0; k < n_el; ++k) S::add(tot, red[(k << lp_log2) + gl]);
        if constexpr (DROP) tot = drop_piece<VEC>(tot, (uint32_t)sr, c, d, step);
        reinterpret_cast<V*>(gh + (int64_t)sr * gh_stride)[c] = tot;
        if constexpr (VEC == 4) {
          if (z.dz) zrow[c] = tot;
        }
      }
      if constexpr (VEC == 4) {
        if (z.dz) {        // one c0 pass: piece c pairs with piece c + N/4, both in this block
          __syncthreads();
          const int zh = z.N / 4;
          if (el == 0 && c < zh) {
            const float4 lo = zrow[c], hi2 = zrow[c + zh];
            const float4 yv = reinterpret_cast<const float4*>(z.y + (int64_t)sr * z.y_stride)[c];
            reinterpret_cast<float4*>(z.dz + (int64_t)sr * z.N)[c] = dz_piece(lo, hi2, yv);
          }
        }
      }
    }
  }
}

// scatter form of the max reducer's backward: lane group per destination row, one fp32 atomic per (edge, column) that
// attained the maximum
template <bool DROP>
__global__ __launch_bounds__(256) void k_spmm_bwd_max(const int32_t* __restrict__ indptr, const int32_t* __restrict__ src,
                                                      const float* __restrict__ go, int32_t go_stride, int64_t n_dst,
                                                      int32_t dim, MaxIn mi, float* __restrict__ gh, int32_t gh_stride,
                                                      int lpr_log2, DropArgs d) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & (lpr - 1);
  const int rows_per_wave = kWave >> lpr_log2;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const int64_t v = wave * rows_per_wave + (lane >> lpr_log2);
  if (v >= n_dst) return;
  const int32_t beg = indptr[v], end = indptr[v + 1];
  if (end == beg) return;
  const uint32_t step = DROP ? drop_step_of(d) : 0u;
  for (int c = gl; c < dim; c += lpr) {
    const float g = go[v * go_stride + c];
    const float o = mi.out[v * mi.out_stride + c];
    for (int32_t e = beg; e < end; ++e) {
      const int32_t sr = src[e];
      float x = mi.h[(int64_t)sr * mi.h_stride + c];
      float gg = g;
      if constexpr (DROP) {
        x = drop_piece<1>(x, (uint32_t)sr, c, d, step);
        gg = drop_piece<1>(g, (uint32_t)sr, c, d, step);
      }
      if (x == o && gg != 0.f) unsafeAtomicAdd(gh + (int64_t)sr * gh_stride + c, gg);
    }
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
  if (reduce != PG_REDUCE_MEAN && reduce != PG_REDUCE_SUM && reduce != PG_REDUCE_MAX) return PG_ERR_INVALID;
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
#define PG_FWD(VEC, MAXR) \
  hipLaunchKernelGGL((k_spmm_fwd<VEC, MAXR>), dim3(grid), dim3(256), 0, st, indptr, src, h, h_stride, n_dst, dim, reduce, out, out_stride, l2, bnd(bounds_elems(h, 4) / h_stride))
  const bool mx = reduce == PG_REDUCE_MAX;
  if (vec == 4) { if (mx) PG_FWD(4, true); else PG_FWD(4, false); }
  else if (vec == 2) { if (mx) PG_FWD(2, true); else PG_FWD(2, false); }
  else { if (mx) PG_FWD(1, true); else PG_FWD(1, false); }
#undef PG_FWD
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_spmm_fwd_drop(const int32_t* indptr, const int32_t* src, const float* h, int32_t h_stride, int64_t n_dst,
                     int32_t dim, int reduce, float* out, int32_t out_stride, const pg_dropout_t* drop,
                     pg_stream_t stream) {
  DropArgs d;
  if (drop && drop->threshold > 65535u) return PG_ERR_INVALID;
  if (!drop_args(drop, &d)) return pg_spmm_fwd(indptr, src, h, h_stride, n_dst, dim, reduce, out, out_stride, stream);
  if (n_dst < 0 || dim <= 0 || h_stride < dim || out_stride < dim) return PG_ERR_INVALID;
  if (reduce != PG_REDUCE_MEAN && reduce != PG_REDUCE_SUM && reduce != PG_REDUCE_MAX) return PG_ERR_INVALID;
  if (!(dim % 4 == 0 && h_stride % 4 == 0 && out_stride % 4 == 0 && al(h, 16) && al(out, 16))) return PG_ERR_UNSUPPORTED;
  if (n_dst == 0) return PG_OK;
  if (!indptr || !out) return PG_ERR_INVALID;
  const int pieces = dim / 4;
  int l2 = log2_ceil_pow2(pieces < 64 ? pieces : 64);
  const int rows_per_block = 4 * (64 >> l2);
  const dim3 grid((unsigned)ceil_div<int64_t>(n_dst, rows_per_block));
  if (reduce == PG_REDUCE_MAX)
    hipLaunchKernelGGL(k_spmm_fwd_drop<true>, grid, dim3(256), 0, as_stream(stream), indptr, src, h, h_stride, n_dst, dim,
                       reduce, out, out_stride, l2, d, bnd(bounds_elems(h, 4) / h_stride));
  else
    hipLaunchKernelGGL(k_spmm_fwd_drop<false>, grid, dim3(256), 0, as_stream(stream), indptr, src, h, h_stride, n_dst, dim,
                       reduce, out, out_stride, l2, d, bnd(bounds_elems(h, 4) / h_stride));
  PG_LAUNCH_CHECK();
  return PG_OK;
}

// One-thread marker kernel: word [3] of the profiling ring's entry (*step) % ring_len = device wall clock (100 MHz). Launched
// right behind a profiled kernel it cannot start before that dispatch has completed, whatever runs next (profiling aid for
// tools/join_stamps_trace.py: a one-thread kernel's stamp is taken within a fraction of a microsecond of its dispatch start,
// which ties the stamps' clock to rocprofv3's; bench.py PG_BENCH_STAMP_SUCCESSOR=1). It also takes the armed successor
// stamp, like any other dependent launch would.
__global__ void k_prof_stamp(unsigned long long* __restrict__ ring, int ring_len, const uint64_t* __restrict__ step,
                             ProfSucc succ) {
  prof_succ_stamp(succ);
  if (threadIdx.x == 0)
    ring[(size_t)((uint32_t)(step ? *step : 0) % (uint32_t)ring_len) * PG_PROF_WORDS + 3] = wall_clock64();
}

int pg_prof_stamp(uint64_t* ring, int32_t ring_len, const uint64_t* step, pg_stream_t stream) {
  if (!ring || ring_len <= 0) return PG_ERR_INVALID;
  hipLaunchKernelGGL(k_prof_stamp, dim3(1), dim3(64), 0, as_stream(stream), reinterpret_cast<unsigned long long*>(ring),
                     (int)ring_len, step, take_prof_succ());
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int pg_spmm_fwd_rows(const int32_t* indptr, const int32_t* src, const pg_row_source_t* rows, int64_t n_dst,
                     int32_t dim, int reduce, float* out, int32_t out_stride, const pg_dropout_t* drop,
                     uint64_t* prof, int32_t prof_ring, pg_stream_t stream) {
  if (!rows || n_dst < 0 || dim <= 0 || out_stride < dim) return PG_ERR_INVALID;
  if (reduce != PG_REDUCE_MEAN && reduce != PG_REDUCE_SUM && reduce != PG_REDUCE_MAX) return PG_ERR_INVALID;
  if (drop && drop->threshold > 65535u) return PG_ERR_INVALID;
  if (prof && prof_ring <= 0) return PG_ERR_INVALID;
  // one wave per destination, 16-byte pieces: the wide feature rows this path exists for. dim % 4 != 0 (602): the
  // last piece is read / written whole, so every row must be padded to a multiple of 4 floats
  const int32_t dim4 = (dim + 3) & ~3;
  if (!(dim >= 256 && out_stride >= dim4 && out_stride % 4 == 0 && al(out, 16))) return PG_ERR_UNSUPPORTED;
  if (rows->cache && !(rows->cache_stride >= dim4 && rows->cache_stride % 4 == 0 && al(rows->cache, 16))) return PG_ERR_UNSUPPORTED;
  if (rows->staged && !(rows->staged_stride >= dim4 && rows->staged_stride % 4 == 0 && al(rows->staged, 16))) return PG_ERR_UNSUPPORTED;
  if (n_dst == 0) return PG_OK;
  if (!indptr || !src || !out || !rows->slots) return PG_ERR_INVALID;
  DropArgs d{};
  const bool has_drop = drop_args(drop, &d);
  if (!has_drop && drop) {                        // the profiling ring is indexed by the caller's step counter
    d.step = drop->step;
    d.step_imm = (uint32_t)drop->step_value;
  }
  const unsigned grid = (unsigned)ceil_div<int64_t>(n_dst, 4);
  unsigned long long* pr = reinterpret_cast<unsigned long long*>(prof);
  // (debug build: rows of the source layer / of the cache / of the staged block, from the registered extents of the buffers)
  const Bnd rb = bnd(bounds_elems(rows->slots, 4), rows->cache ? bounds_elems(rows->cache, 4) / rows->cache_stride : 0,
                     rows->staged ? bounds_elems(rows->staged, 4) / rows->staged_stride : 0);
  (void)rb;
#define PG_FWD_ROWS(DROP, TAIL, MAXR)                                                                                   \
  hipLaunchKernelGGL((k_spmm_fwd_rows<DROP, TAIL, MAXR>), dim3(grid), dim3(256), 0, as_stream(stream), indptr, src,       \
                     rows->slots, rows->edge_slots, rows->cache, rows->cache_stride, rows->staged, rows->staged_stride,   \
                     n_dst, dim, reduce, out, out_stride, d, pr, (int)prof_ring, rb)
#define PG_FWD_ROWS_W(DROP, TAIL, M, MAXR)                                                                              \
  hipLaunchKernelGGL((k_spmm_fwd_rows_w<DROP, TAIL, M, MAXR>), dim3(grid), dim3(256), 0, as_stream(stream), indptr, src, \
                     rows->slots, rows->edge_slots, rows->cache, rows->cache_stride, rows->staged, rows->staged_stride, \
                     n_dst, dim, reduce, out, out_stride, d, pr, (int)prof_ring, store_mode, rb)
#define PG_FWD_ROWS_R(DROP, TAIL, MAXR)                               \
  do {                                                                \
    if (generic || dim4 > 1024) PG_FWD_ROWS(DROP, TAIL, MAXR);        \
    else if (dim4 <= 512) PG_FWD_ROWS_W(DROP, TAIL, 2, MAXR);         \
    else if (dim4 <= 768) PG_FWD_ROWS_W(DROP, TAIL, 3, MAXR);         \
    else PG_FWD_ROWS_W(DROP, TAIL, 4, MAXR);                          \
  } while (0)
#define PG_FWD_ROWS_ANY(DROP, TAIL)                                   \
  do {                                                                \
    if (reduce == PG_REDUCE_MAX) PG_FWD_ROWS_R(DROP, TAIL, true);     \
    else PG_FWD_ROWS_R(DROP, TAIL, false);                            \
  } while (0)
  // rows of up to 1024 floats take the wave-uniform kernel (PG_FWD_ROWS_GENERIC=1: the generic one, for A/B runs)
  static const bool generic = getenv("PG_FWD_ROWS_GENERIC") != nullptr;
  static const int store_mode = fwd_rows_store_mode();
  if (dim % 4 == 0) {
    if (has_drop) PG_FWD_ROWS_ANY(true, false);
    else PG_FWD_ROWS_ANY(false, false);
  } else {
    if (has_drop) PG_FWD_ROWS_ANY(true, true);
    else PG_FWD_ROWS_ANY(false, true);
  }
#undef PG_FWD_ROWS_ANY
#undef PG_FWD_ROWS_R
#undef PG_FWD_ROWS_W
#undef PG_FWD_ROWS
  PG_LAUNCH_CHECK();
  if (pr && d.step) {           // the next dense / head launch of this thread stamps this entry's word [1] (a launch with an
    g_prof_succ.ring = pr;      // immediate step value has no dependent successor on its own stream: body time only)
    g_prof_succ.ring_len = prof_ring;
    g_prof_succ.step = d.step;
  }
  return PG_OK;
}

int pg_compose_edge_slots(const int32_t* src, int64_t n_edges, const int32_t* slots, int64_t n_src,
                          int32_t* edge_slots, pg_stream_t stream) {
  if (n_edges < 0 || n_src < 0) return PG_ERR_INVALID;
  if (n_edges == 0) return PG_OK;
  if (!src || !slots || !edge_slots) return PG_ERR_INVALID;
  int64_t g = ceil_div<int64_t>(n_edges, 256);
  hipLaunchKernelGGL(k_compose_edge_slots, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, as_stream(stream), src,
                     n_edges, slots, n_src, edge_slots);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

static int spmm_bwd_plain_(const int32_t* indptr, const int32_t* src, const float* grad_out, int32_t go_stride, int64_t n_dst,
                           int32_t dim, int reduce, float* grad_h, int32_t gh_stride, pg_stream_t stream);
static int spmm_bwd_drop_(const int32_t* indptr, const int32_t* src, const float* grad_out, int32_t go_stride,
                     int64_t n_dst, int32_t dim, int reduce, float* grad_h, int32_t gh_stride,
                     const pg_dropout_t* drop, pg_stream_t stream) {
  DropArgs d;
  if (drop && drop->threshold > 65535u) return PG_ERR_INVALID;
  if (!drop_args(drop, &d))
    return spmm_bwd_plain_(indptr, src, grad_out, go_stride, n_dst, dim, reduce, grad_h, gh_stride, stream);
  if (n_dst < 0 || dim <= 0 || go_stride < dim || gh_stride < dim || dim % 4) return PG_ERR_INVALID;
  if (reduce != PG_REDUCE_MEAN && reduce != PG_REDUCE_SUM) return PG_ERR_INVALID;
  if (n_dst == 0) return PG_OK;
  if (!indptr || !grad_out) return PG_ERR_INVALID;
  int l2 = log2_ceil_pow2(dim < 64 ? dim : 64);
  const int rows_per_block = 4 * (64 >> l2);
  hipLaunchKernelGGL(k_spmm_bwd_drop, dim3((unsigned)ceil_div<int64_t>(n_dst, rows_per_block)), dim3(256), 0,
                     as_stream(stream), indptr, src, grad_out, go_stride, n_dst, dim, reduce, grad_h, gh_stride, l2, d);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

static int bwd_gather_impl(const int32_t* tptr, const int32_t* tdst, const int32_t* indptr, const float* grad_out,
                           int32_t go_stride, int64_t n_src, int32_t dim, int reduce, float* grad_h, int32_t gh_stride,
                           const int32_t* heavy, int32_t heavy_cap, const pg_dropout_t* drop, const float* act_out,
                           int32_t act_stride, float* dz, const MaxIn* mx, pg_stream_t stream) {
  if (n_src < 0 || dim <= 0 || go_stride < dim || gh_stride < dim) return PG_ERR_INVALID;
  if (!mx && reduce != PG_REDUCE_MEAN && reduce != PG_REDUCE_SUM) return PG_ERR_INVALID;
  if (drop && drop->threshold > 65535u) return PG_ERR_INVALID;
  if (n_src == 0) return PG_OK;
  if (!tptr || (!mx && !indptr) || !grad_h) return PG_ERR_INVALID;  // tdst / grad_out may be NULL only for an edgeless block
  DropArgs d{};
  const bool dr = drop_args(drop, &d);
  if (dr && dim % 4) return PG_ERR_INVALID;
  bool v4 = dim % 4 == 0 && go_stride % 4 == 0 && gh_stride % 4 == 0 && al(grad_out, 16) && al(grad_h, 16);
  MaxIn mi{};
  if (mx) {
    mi = *mx;
    if (!mi.h || !mi.out || mi.h_stride < dim || mi.out_stride < dim) return PG_ERR_INVALID;
    v4 = v4 && mi.h_stride % 4 == 0 && mi.out_stride % 4 == 0 && al(mi.h, 16) && al(mi.out, 16);
    if (dr && !v4) return PG_ERR_UNSUPPORTED;     // the scalar layout's dropout indexing needs nothing else, but keep one rule
  }
  const int pieces = v4 ? dim / 4 : dim;
  int l2 = log2_ceil_pow2(pieces < 64 ? pieces : 64);
  const int rows_per_block = 4 * (64 >> l2);
  DzOut z{};
  if (dz) {
    // dZ of a skip-concat NodeUpdate (2 N columns): 16-byte pieces, the whole row inside one lane group
    if (!act_out || !v4 || dim % 8 || pieces > (1 << l2) || act_stride < dim || act_stride % 4 || !al(act_out, 16) || !al(dz, 16))
      return PG_ERR_UNSUPPORTED;
    z.y = act_out; z.dz = dz; z.y_stride = act_stride; z.N = dim / 2;
  }
  const int64_t row_blocks = ceil_div<int64_t>(n_src, rows_per_block);
  const bool hubs = heavy && heavy_cap > 0;
  // hub rows: up to 16 extra blocks of the same launch (hub i goes to extra block i % 16)
  // ... each hub's columns dealt to `hub_parts` blocks when the row is one pass of an even number of 16-byte pieces
  // (dim 64: 16 pieces, 4 parts of 2 + 2) — see heavy_rows
  constexpr int parts_cfg = 4;
  int hub_parts = 1;
  if (hubs && v4 && pieces <= (1 << l2) && pieces == (1 << l2))
    for (int p = parts_cfg; p > 1; p >>= 1)
      if ((p & (p - 1)) == 0 && pieces % (2 * p) == 0 && (pieces / p) >= 2) { hub_parts = p; break; }
  const int hub_blocks = hubs ? (heavy_cap < 16 ? heavy_cap : 16) * hub_parts : 0;
  const dim3 grid((unsigned)(row_blocks + hub_blocks));
  hipStream_t st = as_stream(stream);
#define PG_BWD_GATHER(VEC, DROP, MAXR)                                                                                 \
  hipLaunchKernelGGL((k_spmm_bwd_gather<VEC, DROP, MAXR>), grid, dim3(256), 0, st, tptr, tdst, indptr, grad_out, go_stride, \
                     n_src, dim, reduce, grad_h, gh_stride, l2, hubs ? 1 : 0, d, heavy, heavy_cap, (int32_t)row_blocks, z, mi, \
                     hub_parts, bnd(bounds_elems(grad_out, 4) / go_stride, n_src))
  if (mx) {
    if (v4 && dr) PG_BWD_GATHER(4, true, true);
    else if (v4) PG_BWD_GATHER(4, false, true);
    else PG_BWD_GATHER(1, false, true);
  } else {
    if (v4 && dr) PG_BWD_GATHER(4, true, false);
    else if (v4) PG_BWD_GATHER(4, false, false);
    else if (dr) PG_BWD_GATHER(1, true, false);
    else PG_BWD_GATHER(1, false, false);
  }
#undef PG_BWD_GATHER
  PG_LAUNCH_CHECK();
  return PG_OK;
}

static int spmm_bwd_gather_dz_(const int32_t* tptr, const int32_t* tdst, const int32_t* indptr, const float* grad_out,
                          int32_t go_stride, int64_t n_src, int32_t dim, int reduce, float* grad_h, int32_t gh_stride,
                          const int32_t* heavy, int32_t heavy_cap, const pg_dropout_t* drop, const float* act_out,
                          int32_t act_stride, float* dz, pg_stream_t stream) {
  return bwd_gather_impl(tptr, tdst, indptr, grad_out, go_stride, n_src, dim, reduce, grad_h, gh_stride, heavy, heavy_cap,
                         drop, act_out, act_stride, dz, nullptr, stream);
}

static int spmm_bwd_gather_max_(const int32_t* tptr, const int32_t* tdst, const float* grad_out, int32_t go_stride,
                           int64_t n_src, int32_t dim, const float* h, int32_t h_stride, const float* out,
                           int32_t out_stride, float* grad_h, int32_t gh_stride, const int32_t* heavy,
                           int32_t heavy_cap, const pg_dropout_t* drop, float* dz, pg_stream_t stream) {
  const MaxIn mi{h, out, h_stride, out_stride};
  return bwd_gather_impl(tptr, tdst, nullptr, grad_out, go_stride, n_src, dim, PG_REDUCE_MAX, grad_h, gh_stride, heavy,
                         heavy_cap, drop, h, h_stride, dz, &mi, stream);
}

static int spmm_bwd_max_(const int32_t* indptr, const int32_t* src, const float* grad_out, int32_t go_stride, int64_t n_dst,
                    int32_t dim, const float* h, int32_t h_stride, const float* out, int32_t out_stride, float* grad_h,
                    int32_t gh_stride, const pg_dropout_t* drop, pg_stream_t stream) {
  if (n_dst < 0 || dim <= 0 || go_stride < dim || gh_stride < dim || h_stride < dim || out_stride < dim) return PG_ERR_INVALID;
  if (drop && drop->threshold > 65535u) return PG_ERR_INVALID;
  if (n_dst == 0) return PG_OK;
  if (!indptr || !grad_out || !out) return PG_ERR_INVALID;  // src / h / grad_h may be NULL only for an edgeless block
  DropArgs d{};
  const bool dr = drop_args(drop, &d);
  if (dr && dim % 4) return PG_ERR_INVALID;
  const MaxIn mi{h, out, h_stride, out_stride};
  int l2 = log2_ceil_pow2(dim < 64 ? dim : 64);
  const int rows_per_block = 4 * (64 >> l2);
  const dim3 grid((unsigned)ceil_div<int64_t>(n_dst, rows_per_block));
  if (dr)
    hipLaunchKernelGGL(k_spmm_bwd_max<true>, grid, dim3(256), 0, as_stream(stream), indptr, src, grad_out, go_stride, n_dst,
                       dim, mi, grad_h, gh_stride, l2, d);
  else
    hipLaunchKernelGGL(k_spmm_bwd_max<false>, grid, dim3(256), 0, as_stream(stream), indptr, src, grad_out, go_stride, n_dst,
                       dim, mi, grad_h, gh_stride, l2, d);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

static int spmm_bwd_plain_(const int32_t* indptr, const int32_t* src, const float* grad_out, int32_t go_stride, int64_t n_dst,
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

// ONE entry point for the backward of a block aggregation (round 6: pg_spmm_bwd, _drop, _max, _gather, _gather_dz and
// _gather_max — 10 to 17 positional arguments — are gone): gather form over the block's source-major copy when the descriptor
// has one (tptr), else the scatter form; PG_REDUCE_MAX takes the forward's input and output.
int pg_spmm_bwd(const pg_spmm_bwd_desc_t* d, pg_stream_t stream) {
  if (!d) return PG_ERR_INVALID;
  const pg_dropout_t* drop = d->has_drop ? &d->drop : nullptr;
  if (d->tptr) {
    if (d->reduce == PG_REDUCE_MAX)
      return spmm_bwd_gather_max_(d->tptr, d->tdst, d->grad_out, d->go_stride, d->n_src, d->dim, d->h, d->h_stride, d->out,
                                  d->out_stride, d->grad_h, d->gh_stride, d->heavy, d->heavy_cap, drop, d->dz, stream);
    return spmm_bwd_gather_dz_(d->tptr, d->tdst, d->indptr, d->grad_out, d->go_stride, d->n_src, d->dim, d->reduce, d->grad_h,
                               d->gh_stride, d->heavy, d->heavy_cap, drop, d->act_out, d->act_stride, d->dz, stream);
  }
  if (d->dz || d->act_out || d->heavy) return PG_ERR_INVALID;          // those belong to the gather form
  if (d->reduce == PG_REDUCE_MAX)
    return spmm_bwd_max_(d->indptr, d->src, d->grad_out, d->go_stride, d->n_dst, d->dim, d->h, d->h_stride, d->out, d->out_stride,
                         d->grad_h, d->gh_stride, drop, stream);
  return spmm_bwd_drop_(d->indptr, d->src, d->grad_out, d->go_stride, d->n_dst, d->dim, d->reduce, d->grad_h, d->gh_stride, drop,
                        stream);
}

}  // extern "C"
