// The per-destination body of k_spmm_fwd_rows_w (pg_spmm.hip), shared with the kernel that also runs layer 0's dense step on
// the aggregated rows (k_agg_dense_fwd, pg_dense.hip): out[v] = reduce over v's in-edges of dropout(row(src[e])), row(p) read
// where it lies (pg_row_source_t: the HBM feature cache or the miss path's staged block; storage.py:176-204 fused into
// gcn_nssc.py:66-74). One wave per destination; the destination is wave-uniform, so the loop bounds, every source row's
// slot / position (readlane of the lane that looked them up) and the row's base address live in SGPRs.
#pragma once
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
  __device__ static inline float4 fill(float f) { return make_float4(f, f, f, f); }
  __device__ static inline void mx(float4& a, const float4& b) { a.x = fmaxf(a.x, b.x); a.y = fmaxf(a.y, b.y); a.z = fmaxf(a.z, b.z); a.w = fmaxf(a.w, b.w); }
  // b where x == o (component-wise), else 0: the max reducer's backward routes a destination's gradient to every in-edge
  // whose message equals the maximum
  __device__ static inline float4 where_eq(const float4& x, const float4& o, const float4& b) {
    return make_float4(x.x == o.x ? b.x : 0.f, x.y == o.y ? b.y : 0.f, x.z == o.z ? b.z : 0.f, x.w == o.w ? b.w : 0.f);
  }
};

constexpr float kNegInf = -__builtin_huge_valf();

// drop_apply in two steps: the 8 keep-bits of a draw (bit i: column i of the first piece, bit 4 + i: of the second) ...
__device__ __forceinline__ uint32_t keep_bits(const uint32_t (&o)[4], uint32_t thr) {
  uint32_t k = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    k |= ((o[i] & 0xffffu) >= thr ? 1u : 0u) << (2 * i);
    k |= ((o[i] >> 16) >= thr ? 1u : 0u) << (2 * i + 1);
  }
  return k;
}
// ... and their application: the same products and zeros as drop_apply(x, o, half, thr, scale) with bits >> 4 * half
__device__ __forceinline__ float4 keep_apply(float4 x, uint32_t bits, float scale) {
  x.x = (bits & 1u) ? x.x * scale : 0.f;
  x.y = (bits & 2u) ? x.y * scale : 0.f;
  x.z = (bits & 4u) ? x.z * scale : 0.f;
  x.w = (bits & 8u) ? x.w * scale : 0.f;
  return x;
}

constexpr int kRowsPair = 2;    // source rows whose loads are in flight together per wave (the sampler's fan-out)

struct RowsW {                  // where the source rows live (pg_row_source_t, resolved by the launcher)
  const int32_t* src;
  const int32_t* slots;
  const int32_t* edge_slots;    // optional: slots[src[e]] per edge
  const float* cache;
  const float* staged;
  int32_t cache_stride, staged_stride;
  Bnd bnd;                      // PG_BOUNDS: [0] rows of the source layer (slots' length), [1] rows of the cache, [2] of the staged block
};

// acc[m] = piece (m * 64 + lane) of the reduction over edges [beg, end) — NOT yet divided by the degree.
// PRE: lane e's (position, slot) of edge beg + e have been loaded by the caller (pre_p, pre_s) for the first 64 edges.
template <bool DROP, bool TAIL, int M, bool MAXR, bool PRE>
__device__ __forceinline__ void rows_w_accumulate(const RowsW& r, int32_t beg, int32_t end, int lane, int pieces, int tail,
                                                  uint32_t step, const DropArgs& d, int32_t pre_p, int32_t pre_s,
                                                  float4 (&acc)[M], bool& any) {
  using S = SV<4>;
  for (int32_t eb = beg; eb < end; eb += kWave) {
    const int ne = end - eb < kWave ? end - eb : kWave;
    int32_t my_p = 0, my_s = -2;
    if (PRE && eb == beg) {
      my_p = pre_p;
      my_s = pre_s;
    } else if (lane < ne) {
      my_p = PG_IDX(r.src[eb + lane], r.bnd, 0, PG_K_FWD_ROWS, 1);
      my_s = r.edge_slots ? r.edge_slots[eb + lane] : r.slots[my_p];
      my_s = bnd_slot(my_s, r.bnd, PG_K_FWD_ROWS, 2);
    }
    for (int e0 = 0; e0 < ne; e0 += kRowsPair) {
      float4 x[kRowsPair][M];
      int32_t srp[kRowsPair], sl[kRowsPair];
      bool ok[kRowsPair];
      // all of the pair's lane reads first: they wait for the index loads, and must not wait for a row load
#pragma unroll
      for (int j = 0; j < kRowsPair; ++j) {
        const int e = e0 + j < ne ? e0 + j : ne - 1;
        sl[j] = __builtin_amdgcn_readlane(my_s, e);
        srp[j] = __builtin_amdgcn_readlane(my_p, e);
        ok[j] = e0 + j < ne && sl[j] != -1 && sl[j] != -2;   // padding / an unresolved miss contributes nothing
      }
#pragma unroll
      for (int j = 0; j < kRowsPair; ++j) {
        if (ok[j]) {
          const float4* hrow = sl[j] >= 0
                                   ? reinterpret_cast<const float4*>(r.cache + (int64_t)sl[j] * r.cache_stride)
                                   : reinterpret_cast<const float4*>(r.staged + (int64_t)(-sl[j] - 3) * r.staged_stride);
#pragma unroll
          for (int m = 0; m < M; ++m) {
            const int c = m * kWave + lane;
            if (m < M - 1 || c < pieces) x[j][m] = hrow[c];   // the launcher picks M = ceil(pieces / 64)
          }
        }
      }
      asm volatile("" ::: "memory");   // the row loads are issued HERE, not sunk to their use behind the draws
      // every draw of both rows happens between the loads' issue and their first use; what is kept of a draw is its
      // 8 keep-bits (two pieces x 4 columns): pieces lane + 64 mm and lane + 64 (mm + 1) share one draw (its two
      // halves), q = (c >> 7) << 6 | (c & 63) — the counters of pg_spmm_fwd_drop
      uint32_t keep[kRowsPair][(M + 1) / 2];
      if constexpr (DROP) {
        // An odd M's last column group is at most half a draw wide; when it is also at most 32 pieces wide (K = 600: pieces
        // 128..149) only lanes 0..31 hold a piece of it, so ONE draw serves both rows of the pair: lanes 0..31 draw for row 0,
        // lanes 32..63 — with the counters of lanes 0..31 — for row 1, whose bits then move down 32 lanes. Same counters, same
        // bits; three draws per pair of rows instead of four (the draws are a quarter-rate multiply chain: the kernel's whole
        // ALU bill).
        const bool packed = (M & 1) && kRowsPair == 2 && pieces - (M - 1) * kWave <= 32;
#pragma unroll
        for (int j = 0; j < kRowsPair; ++j) {
#pragma unroll
          for (int mm = 0; mm < M; mm += 2) {
            keep[j][mm >> 1] = 0;
            if (packed && mm == M - 1) continue;
            if (ok[j] && mm * kWave < pieces) {
              uint32_t o[4];
              Philox::gen((uint32_t)srp[j], (uint32_t)((mm >> 1) * kWave + lane), d.tag, step, d.k0, d.k1, o);
              keep[j][mm >> 1] = keep_bits(o, d.thr);
            }
          }
        }
        if constexpr ((M & 1) != 0) {
          if (packed && (ok[0] || ok[kRowsPair - 1])) {
            const uint32_t row = lane < 32 ? (uint32_t)srp[0] : (uint32_t)srp[kRowsPair - 1];
            uint32_t o[4];
            Philox::gen(row, (uint32_t)(((M - 1) >> 1) * kWave + (lane & 31)), d.tag, step, d.k0, d.k1, o);
            const uint32_t kb = keep_bits(o, d.thr);
            const uint32_t kb_hi = (uint32_t)__shfl_down((int)kb, 32, kWave);
            keep[0][(M - 1) >> 1] = ok[0] ? kb : 0u;
            keep[kRowsPair - 1][(M - 1) >> 1] = ok[kRowsPair - 1] ? kb_hi : 0u;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kRowsPair; ++j) {
        if (!ok[j]) continue;
        if constexpr (MAXR) any = true;
#pragma unroll
        for (int m = 0; m < M; ++m) {
          const int c = m * kWave + lane;
          if (m < M - 1 || c < pieces) {
            float4 xv = x[j][m];
            if constexpr (DROP) xv = keep_apply(xv, keep[j][m >> 1] >> (4 * (m & 1)), d.scale);
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
}

}  // namespace pg
