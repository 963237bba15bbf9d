// Skinny dense step of the GNN layers: Y[n, N] = X[n, K] · Wᵀ + b with N <= 64
// (hidden 32 for GCN, 16 for GraphSAGE, 41-60 classes; pa_gcn.py:130, pa_gs.py:134) and K = feature size —
// NodeUpdate.forward's nn.Linear (PaGraph/model/gcn_nssc.py:18, graphsage_nssc.py:24).
//
// This is the one place the path uses the matrix cores (north star: "MFMA only for the dense
// feat×W step"): exact-fp32 v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, same numerics class
// as the library GEMM it replaces). hipBLASLt spends 86 us (forward) + 52 us (weight gradient)
// per minibatch on this 12 000 x 600 x 32 shape — its macro-tiles are built for square problems —
// which made it the largest item on the compute stream once the gather was fast. The shape is
// HBM/L2-streaming bound: X (29 MB) is read once forward and once backward.
//
// forward   block = NW waves (4 / 8 / 16 by K) on ONE 32-row tile, K split NW ways (K = 600: 16 waves x <= 5 octets of k,
//           4 MFMAs per octet: A = one float4 of its X row per lane, B = one float4 of its weight
//           row per lane — W is read as stored, [N][K], no transpose pass); partial 32x32 tiles reduced
//           through LDS; epilogue fuses bias and NodeUpdate's activation / skip-concat
//           (gcn_nssc.py:18-23).  375 blocks for n = 12 000.
// backward  dW[N, K] = dYᵀ·X, db[N] = Σ dY (any N): block = one (32-feature, 32-column) tile of dW for
//           a chunk of 64-256 rows split over its 4 waves (A = dY rows, B = X rows: both coalesced), LDS
//           reduction, the chunk's partial tile written to scratch; k_sum_partials adds the chunks in order
//           (deterministic). fp32 atomics into dW were tried first: ~70 chunks x 32 floats hit each 128-byte
//           line of the small dW, the memory-side atomic unit serialises them (~12 ns each) and the kernel
//           sat at 25-50 us whatever the tiling. Also used for the output layer (N = 60, K = 64): the library GEMM has two
//           output tiles and a 6000-long reduction there and takes 51 us.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdlib>

#include "pg_common.h"

namespace pg {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float df4 __attribute__((ext_vector_type(4)));

constexpr int kTile = 32;

// act: 0 = none, 1 = relu, 2 = concat(z, relu(z)) -> Y has 2N columns (NodeUpdate's skip connection,
// gcn_nssc.py:20-21). W is the nn.Linear weight as stored: [N, K] row-major.
// Optional second operand pair (X2, W2, bias2): Z = X W^T + X2 W2^T + bias + bias2 — GraphSAGE's NodeUpdate
// `fc_self(h) + fc_neigh(neigh)` (graphsage_nssc.py:24) in one pass, the K range being the concatenation.
// blockIdx.y selects the 32-column tile of N (N <= 64 needs two).
// WV = widest aligned access to a row of W: 4 (K % 4 == 0), 2 (K even: Reddit's 602), 1. K need not be a multiple
// of 8: the last octet is zero-filled past K (X's rows are padded to a multiple of 4 floats, so its 16-byte load stays
// inside the row; whatever lies in the padding is masked).
typedef float df2 __attribute__((ext_vector_type(2)));

// NW = waves per 32-row tile, K split NW ways. A tile's waves are its only memory-level parallelism (12 000 rows are
// 375 tiles — 1.5 per CU): with 4 waves and K = 600 each wave walks 19 octets in five dependent rounds of loads.
// LDSX: a wave's share of X goes through LDS. Straight from global, lane (row, half) reads 16 bytes of ITS row per
// octet — one instruction touches 32 rows = 32 cache lines and uses 32 bytes of each, and the kernel sits at 1.6 TB/s
// whatever the wave count. With LDSX a wave fetches its [32 rows x 32 floats] slab (four octets) coalesced — 8 rows x
// 128 bytes per instruction — parks it in its own LDS region (row stride 36 floats) and reads the MFMA operand layout
// back from there. W (77 KB, re-read by every 32-row tile: as much cache traffic as X itself) takes the same route when
// its rows are 16-byte aligned (K % 4 == 0). 12 000 x 600 x 32: 18.1 us direct, 14.7 with the X slab, 13.7 with both
// (2.1 TB/s of X); the two-operand 12 000 x (600 + 600) x 16: 28.4 -> 21.1 us.
constexpr int kXsStride = 36;

// ROWS (round 3): the first operand's rows are never materialised — row r lives at X + slots[r] * x_stride (the fused
// feature cache) when slots[r] >= 0, at staged - (slots[r] + 3) * staged_stride (the miss queue's staged block) when
// slots[r] <= -3, and is all zeros for -2 / -1 (the padding of a fixed-shape layer): pg_row_source_t, the same
// addressing k_spmm_fwd_rows uses. A lane resolves the (at most four) rows it ever fetches once, before the K loop; the
// bytes multiplied are the ones a gather would have copied, so the result is bit-identical to gather + k_linear_fwd.
struct RowsArg {
  const int32_t* slots;
  const float* staged;
  int32_t staged_stride;
  Bnd bnd;                  // PG_BOUNDS: [1] rows of the cache, [2] rows of the staged block
};

__device__ __forceinline__ const float* row_of(const float* cache, int32_t cstride, const RowsArg& ra, int64_t r) {
  const int32_t sl = bnd_slot(ra.slots[r], ra.bnd, PG_K_LINEAR_ROWS, 1);
  if (sl >= 0) return cache + (int64_t)sl * cstride;
  if (sl <= -3) return ra.staged + (int64_t)(-(sl + 3)) * ra.staged_stride;
  return nullptr;
}

template <int WV, int NW, bool LDSX, bool ROWS = false>
__global__ __launch_bounds__(NW * 64) void k_linear_fwd(const float* __restrict__ X, int32_t x_stride,
                                                    const float* __restrict__ W /* [N][K] */,
                                                    const float* __restrict__ bias /* [N] or null */,
                                                    const float* __restrict__ X2, int32_t x2_stride,
                                                    const float* __restrict__ W2 /* [N][K2] */,
                                                    const float* __restrict__ bias2, int32_t K2,
                                                    float* __restrict__ Y, int32_t y_stride, int64_t n, int32_t K,
                                                    int32_t N, int32_t act, const RowsArg ra = RowsArg{},
                                                    const ProfSucc succ = ProfSucc{}) {
  prof_succ_stamp(succ);
  // one LDS buffer: during the K loop wave w's X slab, afterwards wave w's partial output tile (same region, same wave)
  // (+ with WV == 4 a second slab per wave for its share of W)
  __shared__ __attribute__((aligned(16))) float smem[NW * kTile * kXsStride * ((LDSX && WV == 4) ? 2 : 1)];
  float (*red)[kTile][kXsStride] = reinterpret_cast<float (*)[kTile][kXsStride]>(smem);
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int64_t r0 = (int64_t)blockIdx.x * kTile;
  const int n0 = (int)blockIdx.y * kTile;             // first output column of this block
  const int64_t row = r0 + (lane & 31);
  const int half = lane >> 5;
  const int oct1 = (K + 7) / 8, octets = oct1 + (K2 + 7) / 8;
  const int o_beg = (octets * w) / NW, o_end = (octets * (w + 1)) / NW;
  const bool row_ok = row < n;
  const int col = n0 + (lane & 31);
  const bool col_ok = col < N;
  const float* xr = X + (row_ok ? row : 0) * x_stride + 4 * half;
  bool x1_ok = true;                                  // ROWS: this lane's own row exists (not padding)
  if constexpr (ROWS) {
    const float* rp = row_ok ? row_of(X, x_stride, ra, row) : nullptr;
    x1_ok = rp != nullptr;
    xr = (x1_ok ? rp : X) + 4 * half;
  }
  // ROWS: the four rows this lane fetches for the wave's slab (row (lane >> 3) + 8 i of the tile), resolved once
  const float* slab_row[4] = {nullptr, nullptr, nullptr, nullptr};
  if constexpr (ROWS && LDSX) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t r = r0 + (lane >> 3) + 8 * i;
      slab_row[i] = r < n ? row_of(X, x_stride, ra, r) : nullptr;
    }
  }
  // B operand: lane (col, half) needs W[col][kk + 4*half + j], j = 0..3 -> one 16-byte load per octet
  const float* wr = W + (int64_t)(col_ok ? col : 0) * K + 4 * half;
  const float* xr2 = X2 ? X2 + (row_ok ? row : 0) * x2_stride + 4 * half : nullptr;
  const float* wr2 = W2 ? W2 + (int64_t)(col_ok ? col : 0) * K2 + 4 * half : nullptr;
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  auto load1 = [&](const float* xp, const float* wp, int o, int Kx, df4& a, df4& b) {
    const int left = Kx - (o * 8 + 4 * half);          // columns of this lane's quad that exist
    a = df4{0.f, 0.f, 0.f, 0.f};
    b = df4{0.f, 0.f, 0.f, 0.f};
    if (left <= 0) return;
    a = *reinterpret_cast<const df4*>(xp + o * 8);
    if (WV == 4) {
      b = *reinterpret_cast<const df4*>(wp + o * 8);
    } else if (WV == 2) {
      const df2 lo = *reinterpret_cast<const df2*>(wp + o * 8);
      b.x = lo.x; b.y = lo.y;
      if (left > 2) {
        const df2 hi = *reinterpret_cast<const df2*>(wp + o * 8 + 2);
        b.z = hi.x; b.w = hi.y;
      }
    } else {
      b.x = wp[o * 8];
      if (left > 1) b.y = wp[o * 8 + 1];
      if (left > 2) b.z = wp[o * 8 + 2];
      if (left > 3) b.w = wp[o * 8 + 3];
    }
    if (left < 4) {                                    // the ragged end of K: nothing from beyond it is multiplied
      if (left < 2) { a.y = 0.f; b.y = 0.f; }
      if (left < 3) { a.z = 0.f; b.z = 0.f; }
      a.w = 0.f; b.w = 0.f;
    }
  };
  auto load = [&](int o, df4& a, df4& b) {
    if (o < oct1) {
      load1(xr, wr, o, K, a, b);
      if (ROWS && !x1_ok) a = df4{0.f, 0.f, 0.f, 0.f};
    } else {
      load1(xr2, wr2, o - oct1, K2, a, b);
    }
    if (!row_ok) a = df4{0.f, 0.f, 0.f, 0.f};
    if (!col_ok) b = df4{0.f, 0.f, 0.f, 0.f};
  };
  auto mma = [&](const df4& a, const df4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
  };
  // four octets (8 x 16-byte loads per lane) in flight per iteration: the wave is otherwise one load
  // round trip per 4 MFMAs
  int o = o_beg;
  if constexpr (LDSX) {
    float* xs = smem + (size_t)w * kTile * kXsStride;
    for (; o + 3 < o_end; o += 4) {
      const bool first = o < oct1;
      if (first && o + 3 >= oct1) break;              // a slab never straddles the two operands: the tail loop takes it
      const float* xb = first ? X : X2;
      const int64_t xst = first ? x_stride : x2_stride;
      const int Kx = first ? K : K2;
      const int kbase = (first ? o : o - oct1) * 8;
      // coalesced fetch: chunk c = lane + 64 i -> row c / 8, 16-byte segment c % 8 of the slab
      df4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i, r = c >> 3, kc = kbase + (c & 7) * 4;
        v[i] = df4{0.f, 0.f, 0.f, 0.f};
        const float* rowp = (ROWS && first) ? slab_row[i] : xb + (r0 + r) * xst;
        if (r0 + r < n && kc < Kx && rowp) {
          v[i] = *reinterpret_cast<const df4*>(rowp + kc);   // rows are padded to 4 floats
          const int left = Kx - kc;
          if (left < 4) {
            if (left < 2) v[i].y = 0.f;
            if (left < 3) v[i].z = 0.f;
            v[i].w = 0.f;
          }
        }
      }
      df4 b0, b1, b2, b3, dummy;
      if constexpr (WV == 4) {
        // W's [32 columns x 32 floats] slab the same way (rows of W are 16-byte aligned when K % 4 == 0): every tile
        // re-reads all of W, so its access pattern matters as much as X's
        const float* wb = first ? W : W2;
        float* ws = smem + (size_t)(NW + w) * kTile * kXsStride;
        df4 u[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = lane + 64 * i, cc = n0 + (c >> 3), kc = kbase + (c & 7) * 4;
          u[i] = df4{0.f, 0.f, 0.f, 0.f};
          if (cc < N && kc < Kx) u[i] = *reinterpret_cast<const df4*>(wb + (int64_t)cc * Kx + kc);   // K % 4 == 0: whole quads
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int c = lane + 64 * i;
          *reinterpret_cast<df4*>(ws + (c >> 3) * kXsStride + (c & 7) * 4) = u[i];
        }
      } else {
        load(o, dummy, b0);          // (the direct A loads of these calls are dead code: only b is used)
        load(o + 1, dummy, b1);
        load(o + 2, dummy, b2);
        load(o + 3, dummy, b3);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = lane + 64 * i;
        *reinterpret_cast<df4*>(xs + (c >> 3) * kXsStride + (c & 7) * 4) = v[i];
      }
      // the slab is private to this wave: its lanes run in lock step, so all that is needed is that the writes have
      // landed in LDS before the reads are issued
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if constexpr (WV == 4) {
        const float* wa = smem + (size_t)(NW + w) * kTile * kXsStride + (lane & 31) * kXsStride + 4 * half;
        b0 = *reinterpret_cast<const df4*>(wa);
        b1 = *reinterpret_cast<const df4*>(wa + 8);
        b2 = *reinterpret_cast<const df4*>(wa + 16);
        b3 = *reinterpret_cast<const df4*>(wa + 24);
      }
      const float* xa = xs + (lane & 31) * kXsStride + 4 * half;
      const df4 a0 = *reinterpret_cast<const df4*>(xa);
      const df4 a1 = *reinterpret_cast<const df4*>(xa + 8);
      const df4 a2 = *reinterpret_cast<const df4*>(xa + 16);
      const df4 a3 = *reinterpret_cast<const df4*>(xa + 24);
      mma(a0, b0);
      mma(a1, b1);
      mma(a2, b2);
      mma(a3, b3);
      __builtin_amdgcn_wave_barrier();               // the next slab's writes stay behind these reads
    }
  }
  for (; o + 3 < o_end; o += 4) {
    df4 a0, b0, a1, b1, a2, b2, a3, b3;
    load(o, a0, b0);
    load(o + 1, a1, b1);
    load(o + 2, a2, b2);
    load(o + 3, a3, b3);
    mma(a0, b0);
    mma(a1, b1);
    mma(a2, b2);
    mma(a3, b3);
  }
  for (; o < o_end; ++o) {
    df4 a, b;
    load(o, a, b);
    mma(a, b);
  }
  // C layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) red[w][(r & 3) + 8 * (r >> 2) + 4 * half][lane & 31] = acc[r];
  __syncthreads();
  // 256 threads x 4 outputs: thread t -> row t / 8, cols 4 * (t % 8) .. +3; the waves' partial tiles are added in wave order
  const int orow = threadIdx.x >> 3, oc = (threadIdx.x & 7) * 4;
  if (threadIdx.x < 256 && r0 + orow < n) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = red[0][orow][oc + j] + red[1][orow][oc + j] + red[2][orow][oc + j] + red[3][orow][oc + j];
#pragma unroll
      for (int ww = 4; ww < NW; ++ww) v[j] += red[ww][orow][oc + j];
      if (bias && n0 + oc + j < N) v[j] += bias[n0 + oc + j];
      if (bias2 && n0 + oc + j < N) v[j] += bias2[n0 + oc + j];
    }
    float* yr = Y + (r0 + orow) * y_stride + n0 + oc;
    const bool vec_ok = (N & 3) == 0 && (y_stride & 3) == 0 && n0 + oc + 3 < N;
    if (act == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    if (vec_ok) {
      *reinterpret_cast<df4*>(yr) = df4{v[0], v[1], v[2], v[3]};
      if (act == 2)
        *reinterpret_cast<df4*>(yr + N) = df4{v[0] > 0.f ? v[0] : 0.f, v[1] > 0.f ? v[1] : 0.f,
                                              v[2] > 0.f ? v[2] : 0.f, v[3] > 0.f ? v[3] : 0.f};
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (n0 + oc + j < N) {
          yr[j] = v[j];
          if (act == 2) yr[N + j] = v[j] > 0.f ? v[j] : 0.f;
        }
    }
  }
}

// gradient of the pre-activation z w.r.t. the loss, from the gradient G of the (activated) output and the
// saved output Yout: act 0: G; act 1 (relu): G * (Yout > 0); act 2 (concat): G[:, :N] + G[:, N:] * (Yout[:, :N] > 0)
__device__ __forceinline__ float dz_at(const float* __restrict__ G, int32_t g_stride, const float* __restrict__ Yout,
                                       int32_t yo_stride, int64_t r, int i, int N, int act) {
  float g = G[r * g_stride + i];
  if (act == 1) g = Yout[r * yo_stride + i] > 0.f ? g : 0.f;
  else if (act == 2) g += Yout[r * yo_stride + i] > 0.f ? G[r * g_stride + N + i] : 0.f;
  return g;
}

// materialise dZ [n, N] once (every column-slice wave of k_linear_bwd_w re-reads its A operand; deriving
// it on the fly there re-did this 19 times and tripled the kernel's time)
__global__ __launch_bounds__(256) void k_dz(const float* __restrict__ G, int32_t g_stride,
                                            const float* __restrict__ Yout, int32_t yo_stride, int64_t n, int32_t N,
                                            int32_t act, float* __restrict__ dz) {
  const int64_t total = n * N;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / N;
    const int i = (int)(t - r * N);
    dz[t] = dz_at(G, g_stride, Yout, yo_stride, r, i, N, act);
  }
}

// block = 4 waves on ONE (32-feature, 32-column) tile of dW, each wave reducing its own `rpw` rows
// (16 rows = 8 MFMA steps with all 16 loads in flight per iteration); the four partial tiles are summed
// through LDS and leave as one set of atomics per block.
// ROWS: X's rows through a pg_row_source_t (see k_linear_fwd): lane l of a wave looks up the slot of the wave's l-th row
// once (rpw <= 64), a row's base address then comes from that lane by shuffle.
struct BwdWArgs {              // one weight-gradient launch (k_linear_bwd_w's former parameter list)
  const float* dY;
  const float* X;
  float* part;
  int64_t n;
  int32_t dy_stride, x_stride, K, N, with_bias, rpw, items, chunks;
  RowsArg ra;
};

template <bool ROWS>
__device__ __forceinline__ void linear_bwd_w_block(const BwdWArgs& g, unsigned bid, float (&red)[4][kTile][kTile + 1],
                                                   float (&bred)[4][kWave]) {
  const float* __restrict__ dY = g.dY;
  const float* __restrict__ X = g.X;
  float* __restrict__ part = g.part;
  const int32_t dy_stride = g.dy_stride, x_stride = g.x_stride, K = g.K, N = g.N, with_bias = g.with_bias, rpw = g.rpw,
                items = g.items, chunks = g.chunks;
  const int64_t n = g.n;
  const RowsArg& ra = g.ra;
  const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
  const int half = lane >> 5;
  // XCD-aware order: workgroups go round-robin over the 8 XCDs, so workgroup id -> (xcd, q); all `items`
  // tiles of one row chunk run back to back on ONE XCD. Their 128-byte column strips of X straddle cache
  // lines (row stride 2400 B), which that XCD's L2 then fetches once instead of once per neighbour.
  const int xcd = (int)(bid & 7), q = (int)(bid >> 3);
  const int item = q % items;
  const int chunk = (q / items) * 8 + xcd;
  if (chunk >= chunks) return;
  const int col_slices = (K + kTile - 1) / kTile;
  const int ft = item / col_slices;
  const int f0 = ft * kTile;
  const int c0 = (item - ft * col_slices) * kTile;
  const int i = f0 + (lane & 31);                    // output feature fed by this lane (A operand)
  const int64_t rb = ((int64_t)chunk * 4 + w) * rpw;
  const int64_t re = (rb + rpw < n) ? rb + rpw : n;
  const bool a_ok = i < N, b_ok = c0 + (lane & 31) < K;
  const bool do_bias = with_bias && c0 == 0;
  const float* ap = dY + (a_ok ? i : 0);
  const int bcol = c0 + (b_ok ? (lane & 31) : 0);
  const float* bp = X + bcol;
  // ROWS: lane l holds the slot of the wave's l-th row; a row pair's slots come back by v_readlane (the pair is
  // wave-uniform: lane half h works on row base + h)
  int32_t my_slot = -2;
  if constexpr (ROWS) {
    const int64_t r = rb + lane;
    if (lane < rpw && r < re) my_slot = bnd_slot(ra.slots[r], ra.bnd, PG_K_BWD_W_ROWS, 1);
  }
  auto xload = [&](int rel /* wave-uniform: even row of the pair, relative to rb */, int64_t r) -> float {
    if constexpr (ROWS) {
      const int32_t s_e = __builtin_amdgcn_readlane(my_slot, rel);
      const int32_t s_o = __builtin_amdgcn_readlane(my_slot, rel + 1 < kWave ? rel + 1 : kWave - 1);
      const int32_t sl = half ? s_o : s_e;
      if (sl >= 0) return X[(int64_t)sl * x_stride + bcol];
      if (sl <= -3) return ra.staged[(int64_t)(-(sl + 3)) * ra.staged_stride + bcol];
      return 0.f;
    } else {
      return bp[r * x_stride];
    }
  };
  f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float bsum = 0.f;
  // `base` is wave-uniform (MFMA needs the whole wave); lane half h works on row base + h
  int64_t base = rb;
  for (; base + 15 < re; base += 16) {
    const int64_t r = base + half;
    const int ub = __builtin_amdgcn_readfirstlane((int)(base - rb));
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = ap[(r + 2 * u) * dy_stride];
      b[u] = xload(ub + 2 * u, r + 2 * u);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float av = a_ok ? a[u] : 0.f, bv = b_ok ? b[u] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      bsum += av;
    }
  }
  for (; base < re; base += 2) {  // tail: the odd half may fall off the end
    const int64_t r = base + half;
    const bool ok = r < re;
    const int ub = __builtin_amdgcn_readfirstlane((int)(base - rb));
    const float a = (ok && a_ok) ? ap[r * dy_stride] : 0.f;
    const float bx = (ROWS || ok) ? xload(ub, ok ? r : rb) : 0.f;     // (ROWS: a row past the end has slot -2 -> 0)
    const float b = (ok && b_ok) ? bx : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    bsum += a;
  }
  // C[row = out feature][col = X column]
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) red[w][(rr & 3) + 8 * (rr >> 2) + 4 * half][lane & 31] = acc[rr];
  bred[w][lane] = bsum;
  __syncthreads();
  const int orow = threadIdx.x >> 3, oc = (threadIdx.x & 7) * 4;
  const int of = f0 + orow;
  float* mine = part + (int64_t)chunk * ((int64_t)N * K + N);      // this chunk's partial [N*K | N]
  if (of < N) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + oc + j < K)
        mine[(int64_t)of * K + c0 + oc + j] =
            red[0][orow][oc + j] + red[1][orow][oc + j] + red[2][orow][oc + j] + red[3][orow][oc + j];
  }
  if (do_bias && threadIdx.x < kTile && f0 + (int)threadIdx.x < N) {
    const int t = threadIdx.x;
    mine[(int64_t)N * K + f0 + t] = bred[0][t] + bred[0][t + 32] + bred[1][t] + bred[1][t + 32] + bred[2][t] +
                                    bred[2][t + 32] + bred[3][t] + bred[3][t + 32];
  }
}

template <bool ROWS>
__global__ __launch_bounds__(256) void k_linear_bwd_w(const BwdWArgs g) {
  __shared__ float red[4][kTile][kTile + 1];
  __shared__ float bred[4][kWave];
  linear_bwd_w_block<ROWS>(g, blockIdx.x, red, bred);
}

// GraphSAGE's NodeUpdate z = fc_self(h) + fc_neigh(neigh) (graphsage_nssc.py:24) has TWO weight gradients over the same
// dZ: dW_self = dZ^T h, dW_neigh = dZ^T neigh. One launch: workgroups [0, g1) are the first operand's tiles, the rest the
// second's — the same blocks doing the same arithmetic as two k_linear_bwd_w launches (bit-identical partial rows), minus
// one kernel boundary of the replayed step per NodeUpdate use (each of these launches is 5-15 us of mostly latency).
template <bool ROWS1>
__global__ __launch_bounds__(256) void k_linear_bwd_w_pair(const BwdWArgs a, const BwdWArgs b, unsigned g1) {
  __shared__ float red[4][kTile][kTile + 1];
  __shared__ float bred[4][kWave];
  if (blockIdx.x < g1) linear_bwd_w_block<ROWS1>(a, blockIdx.x, red, bred);
  else linear_bwd_w_block<false>(b, blockIdx.x - g1, red, bred);
}

// dW / db = sum over the row chunks' partials in a fixed order (deterministic). Block = 64 outputs x 4
// chunk groups, 8 independent loads in flight per thread (a one-thread-per-output loop over ~70 chunks is
// a chain of dependent-latency loads: 12 us for 5 MB).
__global__ __launch_bounds__(256) void k_sum_partials(const float* __restrict__ part, int32_t chunks, int64_t nk,
                                                      int32_t N, float* __restrict__ dW, float* __restrict__ db,
                                                      int64_t len /* floats between two partial rows, >= nk + N */) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + tx;
  const bool ok = j < (db ? nk + N : nk);
  const int per = (chunks + 3) / 4;
  const int cb = g * per, ce = (cb + per < chunks) ? cb + per : chunks;
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (ok) {
    int c = cb;
    for (; c + 7 < ce; c += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += part[(int64_t)(c + u) * len + j];
    }
    for (; c < ce; ++c) a[0] += part[(int64_t)c * len + j];
  }
  red[g][tx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (g == 0 && ok) {
    const float v = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    if (j < nk) dW[j] = v;
    else db[j - nk] = v;
  }
}

static inline int bwd_rows_per_wave(int64_t n, int32_t K, int32_t N) {
  // rows per wave: 64, shrunk (to >= 16) until the launch has ~1000 blocks
  const int64_t items = ceil_div<int64_t>(K, kTile) * ceil_div<int64_t>(N, kTile);
  int rpw = 64;
  while (rpw > 16 && items * ceil_div<int64_t>(n, 4 * rpw) < 1024) rpw >>= 1;
  return rpw;
}

}  // namespace pg

using namespace pg;

extern "C" {

// a row source's envelope for the dense kernels: 16-byte aligned rows in both homes
static int rows_arg(const pg_row_source_t* X, int32_t K, const float** base, int32_t* stride, RowsArg* ra) {
  if (!X || !X->slots) return PG_ERR_INVALID;
  if (!X->cache && !X->staged) return PG_ERR_INVALID;
  const int32_t k4 = (K + 3) & ~3;
  if (X->cache && (X->cache_stride < k4 || (X->cache_stride & 3) || (reinterpret_cast<uintptr_t>(X->cache) & 15)))
    return PG_ERR_UNSUPPORTED;
  if (X->staged && (X->staged_stride < k4 || (X->staged_stride & 3) || (reinterpret_cast<uintptr_t>(X->staged) & 15)))
    return PG_ERR_UNSUPPORTED;
  // a home that does not exist is never addressed (no slot points into it); its base only has to be a valid pointer
  *base = X->cache ? X->cache : X->staged;
  *stride = X->cache ? X->cache_stride : X->staged_stride;
  ra->slots = X->slots;
  ra->staged = X->staged ? X->staged : X->cache;
  ra->staged_stride = X->staged ? X->staged_stride : X->cache_stride;
  ra->bnd = bnd(0, X->cache ? bounds_elems(X->cache, 4) / X->cache_stride : 0,
                X->staged ? bounds_elems(X->staged, 4) / X->staged_stride : (X->cache ? bounds_elems(X->cache, 4) / X->cache_stride : 0));
  return PG_OK;
}

static int linear_fwd(const float* X, int32_t x_stride, const float* W, const float* bias, const float* X2,
                      int32_t x2_stride, const float* W2, const float* bias2, int32_t K2, float* Y, int32_t y_stride,
                      int64_t n, int32_t K, int32_t N, int32_t act, pg_stream_t stream, const RowsArg* rows = nullptr) {
  if (n < 0 || K <= 0 || N <= 0 || K2 < 0 || x_stride < K || act < 0 || act > 2 || y_stride < (act == 2 ? 2 * N : N))
    return PG_ERR_INVALID;
  if (K2 > 0 && (!X2 || !W2 || x2_stride < K2)) return PG_ERR_INVALID;
  // X rows: 16-byte quads (stride a multiple of 4 floats, so the last quad of a ragged K stays inside the row)
  if (N > 2 * kTile || (x_stride & 3) || (K2 > 0 && (x2_stride & 3))) return PG_ERR_UNSUPPORTED;
  if (n == 0) return PG_OK;
  if (!X || !W || !Y) return PG_ERR_INVALID;
  if (reinterpret_cast<uintptr_t>(X) & 15) return PG_ERR_UNSUPPORTED;
  if (K2 > 0 && (reinterpret_cast<uintptr_t>(X2) & 15)) return PG_ERR_UNSUPPORTED;
  auto wv_of = [](const float* w, int32_t k) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(w);
    return (k % 4 == 0 && !(a & 15)) ? 4 : ((k % 2 == 0 && !(a & 7)) ? 2 : 1);
  };
  int wv = wv_of(W, K);
  if (K2 > 0) {
    const int w2 = wv_of(W2, K2);
    wv = w2 < wv ? w2 : wv;
  }
  if (K2 == 0) X2 = W2 = bias2 = nullptr;
  const dim3 grid((unsigned)ceil_div<int64_t>(n, kTile), (unsigned)ceil_div<int>(N, kTile));
  // waves per tile: enough that a wave's share of K is a round or two of loads (each round = 4 octets in flight)
  const int octets = (K + 7) / 8 + (K2 + 7) / 8;
  // measured with both slabs in LDS: 8 waves win from K = 600 (13.7 us vs 14.2 / 15.3) to the two-operand K = 1200
  // (21.1 vs 23.5 / 24.7 at 12 000 rows; 13.2 vs 12.6 with 16 waves at 6 000)
  int nw = octets >= 32 ? 8 : 4;
  constexpr bool no_lds = false;              // (the straight-from-global variant of round 2: 18.1 us against 13.7)
  const ProfSucc succ = take_prof_succ();      // a profiled predecessor's "my successor started" stamp (pg_common.h)
#define PG_LIN_FWD(WV, NW)                                                                                          \
  do {                                                                                                              \
    if (rows)                                                                                                       \
      hipLaunchKernelGGL((k_linear_fwd<WV, NW, true, true>), grid, dim3(NW * 64), 0, as_stream(stream), X, x_stride, \
                         W, bias, X2, x2_stride, W2, bias2, K2, Y, y_stride, n, K, N, act, *rows, succ);            \
    else if (no_lds)                                                                                                \
      hipLaunchKernelGGL((k_linear_fwd<WV, NW, false>), grid, dim3(NW * 64), 0, as_stream(stream), X, x_stride, W,  \
                         bias, X2, x2_stride, W2, bias2, K2, Y, y_stride, n, K, N, act, RowsArg{}, succ);           \
    else                                                                                                            \
      hipLaunchKernelGGL((k_linear_fwd<WV, NW, true>), grid, dim3(NW * 64), 0, as_stream(stream), X, x_stride, W,   \
                         bias, X2, x2_stride, W2, bias2, K2, Y, y_stride, n, K, N, act, RowsArg{}, succ);           \
  } while (0)
#define PG_LIN_FWD_NW(WV)        \
  if (nw == 16) PG_LIN_FWD(WV, 16); \
  else if (nw == 8) PG_LIN_FWD(WV, 8); \
  else PG_LIN_FWD(WV, 4)
  if (wv == 4) { PG_LIN_FWD_NW(4); }
  else if (wv == 2) { PG_LIN_FWD_NW(2); }
  else { PG_LIN_FWD_NW(1); }
#undef PG_LIN_FWD_NW
#undef PG_LIN_FWD
  PG_LAUNCH_CHECK();
  return PG_OK;
}

// ONE forward entry point (round 6: pg_linear_fwd, pg_linear2_fwd and pg_linear2_fwd_rows — 11 to 16 positional arguments — are
// gone): the descriptor's first operand is dense (X1) or read in place (X1rows), the second one optional (K2 == 0: none).
int pg_linear_fwd(const pg_linear_fwd_desc_t* d, pg_stream_t stream) {
  if (!d) return PG_ERR_INVALID;
  if ((d->X1 != nullptr) == (d->X1rows != nullptr)) return PG_ERR_INVALID;       // exactly one form of the first operand
  if (d->K2 < 0) return PG_ERR_INVALID;
  if (d->X1rows) {
    const float* base = nullptr;
    int32_t stride = 0;
    RowsArg ra{};
    const int rc = rows_arg(d->X1rows, d->K1, &base, &stride, &ra);
    if (rc != PG_OK) return rc;
    return linear_fwd(base, stride, d->W1, d->bias1, d->X2, d->x2_stride, d->W2, d->bias2, d->K2, d->Y, d->y_stride, d->n, d->K1,
                      d->N, d->act, stream, &ra);
  }
  return linear_fwd(d->X1, d->x1_stride, d->W1, d->bias1, d->X2, d->x2_stride, d->W2, d->bias2, d->K2, d->Y, d->y_stride, d->n,
                    d->K1, d->N, d->act, stream);
}

/* out[j] = sum over chunks of part[c][j] (chunk order), j < nk -> dW[j], nk <= j < nk + N -> db[j - nk] */
int pg_sum_partials(const float* partials, int32_t chunks, int64_t nk, int32_t N, float* dW, float* db,
                    pg_stream_t stream) {
  return pg_sum_partials_strided(partials, chunks, nk, N, nk + N, dW, db, stream);
}

int pg_sum_partials_strided(const float* partials, int32_t chunks, int64_t nk, int32_t N, int64_t row_len, float* dW,
                            float* db, pg_stream_t stream) {
  if (!partials || chunks <= 0 || nk <= 0 || N < 0 || !dW || row_len < nk + N) return PG_ERR_INVALID;
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)ceil_div<int64_t>(nk + N, 64)), dim3(256), 0, as_stream(stream),
                     partials, chunks, nk, N, dW, db, row_len);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

int64_t pg_linear_bwd_w_scratch(int64_t n, int32_t K, int32_t N) {
  if (n <= 0 || K <= 0 || N <= 0) return 0;
  return ceil_div<int64_t>(n, 4 * bwd_rows_per_wave(n, K, N)) * ((int64_t)N * K + N);
}

static int linear_bwd_w(const float* dY, int32_t dy_stride, const float* X, int32_t x_stride, int64_t n, int32_t K,
                        int32_t N, float* dW, float* db, const float* Yout, int32_t yo_stride, int32_t act,
                        float* dz_scratch, float* partials, int32_t sum_partials, const RowsArg* rows, pg_stream_t stream);

// fills the launch record of one weight gradient; returns its grid (0: nothing to do) or a negative error
static int64_t bwd_w_args(BwdWArgs* a, const float* dY, int32_t dy_stride, const float* X, int32_t x_stride, int64_t n,
                          int32_t K, int32_t N, float* partials, bool with_bias, const RowsArg* rows, int64_t* chunks_out) {
  const int64_t items = ceil_div<int64_t>(K, kTile) * ceil_div<int64_t>(N, kTile);
  const int rpw = bwd_rows_per_wave(n, K, N);
  const int64_t chunks = ceil_div<int64_t>(n, 4 * rpw);
  const int64_t grid = items * 8 * ceil_div<int64_t>(chunks, 8);
  if (grid > 0x7fffffff) return PG_ERR_INVALID;
  a->dY = dY; a->X = X; a->part = partials; a->n = n;
  a->dy_stride = dy_stride; a->x_stride = x_stride; a->K = K; a->N = N;
  a->with_bias = with_bias ? 1 : 0; a->rpw = rpw; a->items = (int32_t)items; a->chunks = (int32_t)chunks;
  a->ra = rows ? *rows : RowsArg{};
  *chunks_out = chunks;
  return grid;
}

static int linear_bwd_w(const float* dY, int32_t dy_stride, const float* X, int32_t x_stride, int64_t n, int32_t K,
                        int32_t N, float* dW, float* db, const float* Yout, int32_t yo_stride, int32_t act,
                        float* dz_scratch, float* partials, int32_t sum_partials, const RowsArg* rows, pg_stream_t stream) {
  if (n < 0 || K <= 0 || N <= 0 || x_stride < K || act < 0 || act > 2 || dy_stride < (act == 2 ? 2 * N : N))
    return PG_ERR_INVALID;
  if (act != 0 && (!Yout || yo_stride < N || !dz_scratch)) return PG_ERR_INVALID;
  if (n == 0) return PG_OK;
  if (!dY || !X || !dW || !partials) return PG_ERR_INVALID;
  if (act != 0) {
    int64_t g = ceil_div<int64_t>(n * N, 256);
    hipLaunchKernelGGL(k_dz, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, as_stream(stream), dY, dy_stride, Yout,
                       yo_stride, n, N, act, dz_scratch);
    PG_LAUNCH_CHECK();
    dY = dz_scratch;
    dy_stride = N;
  }
  BwdWArgs a{};
  int64_t chunks = 0;
  const int64_t grid = bwd_w_args(&a, dY, dy_stride, X, x_stride, n, K, N, partials, db != nullptr, rows, &chunks);
  if (grid < 0) return (int)grid;
  if (rows) hipLaunchKernelGGL(k_linear_bwd_w<true>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
  else hipLaunchKernelGGL(k_linear_bwd_w<false>, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), a);
  PG_LAUNCH_CHECK();
  if (!sum_partials) return PG_OK;   // the consumer (pg_adam_step) adds the chunks up itself
  const int64_t nk = (int64_t)N * K;
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)ceil_div<int64_t>(nk + N, 64)), dim3(256), 0, as_stream(stream),
                     partials, (int32_t)chunks, nk, N, dW, db, nk + N);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

static int linear2_bwd_w(const float* dY, int32_t dy_stride, const float* X1, int32_t x1_stride, const pg_row_source_t* X1rows,
                         int32_t K1, const float* X2, int32_t x2_stride, int32_t K2, int64_t n, int32_t N, float* dW1, float* db1,
                         float* dW2, float* db2, const float* Yout, int32_t yo_stride, int32_t act, float* dz_scratch,
                         float* partials1, float* partials2, int32_t sum_partials, pg_stream_t stream) {
  if (n < 0 || K1 <= 0 || K2 <= 0 || N <= 0 || x2_stride < K2 || act < 0 || act > 2 || dy_stride < (act == 2 ? 2 * N : N))
    return PG_ERR_INVALID;
  if (act != 0 && (!Yout || yo_stride < N || !dz_scratch)) return PG_ERR_INVALID;
  if ((X1 != nullptr) == (X1rows != nullptr)) return PG_ERR_INVALID;          // exactly one form of the first operand
  const float* base = X1;
  int32_t stride = x1_stride;
  RowsArg ra{};
  if (X1rows) {
    const int rc = rows_arg(X1rows, K1, &base, &stride, &ra);
    if (rc != PG_OK) return rc;
  } else if (x1_stride < K1) {
    return PG_ERR_INVALID;
  }
  if (n == 0) return PG_OK;
  if (!dY || !X2 || !dW1 || !dW2 || !partials1 || !partials2) return PG_ERR_INVALID;
  if (act != 0) {
    int64_t g = ceil_div<int64_t>(n * N, 256);
    hipLaunchKernelGGL(k_dz, dim3((unsigned)(g > 2048 ? 2048 : g)), dim3(256), 0, as_stream(stream), dY, dy_stride, Yout,
                       yo_stride, n, N, act, dz_scratch);
    PG_LAUNCH_CHECK();
    dY = dz_scratch;
    dy_stride = N;
  }
  BwdWArgs a{}, b{};
  int64_t ch1 = 0, ch2 = 0;
  const int64_t g1 = bwd_w_args(&a, dY, dy_stride, base, stride, n, K1, N, partials1, db1 != nullptr, X1rows ? &ra : nullptr, &ch1);
  const int64_t g2 = bwd_w_args(&b, dY, dy_stride, X2, x2_stride, n, K2, N, partials2, db2 != nullptr, nullptr, &ch2);
  if (g1 < 0 || g2 < 0 || g1 + g2 > 0x7fffffff) return PG_ERR_INVALID;
  if (X1rows)
    hipLaunchKernelGGL(k_linear_bwd_w_pair<true>, dim3((unsigned)(g1 + g2)), dim3(256), 0, as_stream(stream), a, b, (unsigned)g1);
  else
    hipLaunchKernelGGL(k_linear_bwd_w_pair<false>, dim3((unsigned)(g1 + g2)), dim3(256), 0, as_stream(stream), a, b, (unsigned)g1);
  PG_LAUNCH_CHECK();
  if (!sum_partials) return PG_OK;
  const int64_t nk1 = (int64_t)N * K1, nk2 = (int64_t)N * K2;
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)ceil_div<int64_t>(nk1 + N, 64)), dim3(256), 0, as_stream(stream),
                     partials1, (int32_t)ch1, nk1, N, dW1, db1, nk1 + N);
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)ceil_div<int64_t>(nk2 + N, 64)), dim3(256), 0, as_stream(stream),
                     partials2, (int32_t)ch2, nk2, N, dW2, db2, nk2 + N);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

// ONE weight-gradient entry point (round 6: pg_linear_bwd_w, _ex, _rows and pg_linear2_bwd_w — 15 to 23 positional arguments —
// are gone): one operand (K2 == 0) or GraphSAGE's two over the same dZ in one launch; the first dense or read in place.
int pg_linear_bwd_w(const pg_linear_bwd_desc_t* d, pg_stream_t stream) {
  if (!d) return PG_ERR_INVALID;
  if ((d->X1 != nullptr) == (d->X1rows != nullptr)) return PG_ERR_INVALID;
  if (d->K2 > 0)
    return linear2_bwd_w(d->dY, d->dy_stride, d->X1, d->x1_stride, d->X1rows, d->K1, d->X2, d->x2_stride, d->K2, d->n, d->N, d->dW1,
                         d->db1, d->dW2, d->db2, d->Yout, d->yo_stride, d->act, d->dz_scratch, d->partials1, d->partials2,
                         d->sum_partials, stream);
  if (d->K2 < 0 || d->X2 || d->dW2 || d->db2 || d->partials2) return PG_ERR_INVALID;
  if (d->X1rows) {
    const float* base = nullptr;
    int32_t stride = 0;
    RowsArg ra{};
    const int rc = rows_arg(d->X1rows, d->K1, &base, &stride, &ra);
    if (rc != PG_OK) return rc;
    return linear_bwd_w(d->dY, d->dy_stride, base, stride, d->n, d->K1, d->N, d->dW1, d->db1, d->Yout, d->yo_stride, d->act,
                        d->dz_scratch, d->partials1, d->sum_partials, &ra, stream);
  }
  return linear_bwd_w(d->dY, d->dy_stride, d->X1, d->x1_stride, d->n, d->K1, d->N, d->dW1, d->db1, d->Yout, d->yo_stride, d->act,
                      d->dz_scratch, d->partials1, d->sum_partials, nullptr, stream);
}

}  // extern "C"
