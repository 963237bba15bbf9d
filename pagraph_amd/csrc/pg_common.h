// Shared bits of libpagraph_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../../include/pagraph_hip.h"

// a pair of HIP events (pg_timer_*): either recorded around a launch (pg_timer_start / _stop) or attached
// to the launch itself (hipExtLaunchKernelGGL: begin / end of that dispatch only)
struct pg_timer {
  hipEvent_t start, stop;
};

// internal (pg_dense.hip; not exported): ordered sum of per-chunk partial tiles
extern "C" __attribute__((visibility("hidden"))) int pg_sum_partials(const float* partials, int32_t chunks, int64_t nk, int32_t N, float* dW, float* db,
                               pg_stream_t stream);
// ... with `row_len` floats between two partial rows (pg_head pads its rows to whole 16-byte pieces)
extern "C" __attribute__((visibility("hidden"))) int pg_sum_partials_strided(const float* partials, int32_t chunks, int64_t nk, int32_t N, int64_t row_len,
                                       float* dW, float* db, pg_stream_t stream);

namespace pg {

// internal (pg_gather.hip): rows [n * start_num / 256, n) of a miss list read from the pinned host table by the device;
// pos == NULL writes row j of the list to row j of `out`, else to row pos[j] - pos_lo (rows below pos_lo skipped)
int scatter_host_tail(const float* table, int64_t table_stride, const int32_t* pos, int32_t pos_lo, const int64_t* fullid,
                      int64_t n_max, const int32_t* n_dev, int32_t start_num, int32_t dim, float* out, int32_t out_stride,
                      pg_stream_t stream);

constexpr int kWave = 64;

extern thread_local int g_last_hip_error;

// ---- self-timing of a kernel inside a replayed hipGraph (HIP events cannot bracket one kernel there) ----------------
// One ring entry = PG_PROF_WORDS uint64 (include/pagraph_hip.h): [0] first wave's start, [1] the SUCCESSOR's first wave's
// start, [2] a count the kernel reports (edges), [PG_PROF_END0 + PG_PROF_SHARD_STRIDE * s] the latest end-of-block stamp
// of shard s (block b stamps shard b % PG_PROF_SHARDS, so "kernel body end" = the maximum over the shards: EVERY block
// takes part — round 3 stamped only the last 256 block indices, which in a fixed-shape launch are padding blocks that
// retire before the last real ones; every shard has a 128-byte line of its own). 100 MHz device wall clock.
// The successor's stamp: a launcher that profiles arms this thread-local record; the next dependent launch of one of the
// dense / head kernels on the same host thread takes it (once) and its block 0 writes entry word [1]. "Start of this kernel's
// first wave -> start of its successor's first wave" is the time the kernel occupies its stream: body + drain + the
// end-of-kernel release + the next dispatch's launch latency — what rocprofv3's End - Start of the same dispatch shows.
struct ProfSucc {
  unsigned long long* ring = nullptr;
  int32_t ring_len = 0;
  const uint64_t* step = nullptr;
};
extern thread_local ProfSucc g_prof_succ;
inline ProfSucc take_prof_succ() {
  ProfSucc p = g_prof_succ;
  g_prof_succ = ProfSucc{};
  return p;
}
__device__ __forceinline__ void prof_succ_stamp(const ProfSucc& p) {
  if (p.ring && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    p.ring[(size_t)((uint32_t)(p.step ? *p.step : 0) % (uint32_t)p.ring_len) * PG_PROF_WORDS + 1] = wall_clock64();
}
// first thing in a profiled kernel (every thread calls it): returns this launch's ring entry or null
__device__ __forceinline__ unsigned long long* prof_begin(unsigned long long* prof, int ring_len, uint32_t step,
                                                          unsigned long long count) {
  if (!prof) return nullptr;
  unsigned long long* e = prof + (size_t)(step % (uint32_t)ring_len) * PG_PROF_WORDS;
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      e[0] = wall_clock64();
      e[2] = count;
    }
    // the NEXT launch's entry is cleared here: nobody writes it before this launch has completed
    if (ring_len > 1)
      for (int i = threadIdx.x; i < PG_PROF_WORDS; i += blockDim.x)
        prof[(size_t)((step + 1u) % (uint32_t)ring_len) * PG_PROF_WORDS + i] = 0ull;
  }
  return e;
}
// last thing in a profiled kernel (every thread of every block calls it; contains a block barrier)
__device__ __forceinline__ void prof_end(unsigned long long* e) {
  if (!e) return;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(e + PG_PROF_END0 + PG_PROF_SHARD_STRIDE * (blockIdx.x % PG_PROF_SHARDS), wall_clock64());
}

// ---- PG_BOUNDS: the debug build that bounds-checks ids ----------------------------------------------------------
// `make bounds` compiles every unit with -DPG_BOUNDS into libpagraph_hip_bounds.so (PG_BOUNDS=1 makes pagraph_amd/_lib.py load
// that one). Every index a kernel reads from a buffer SOMEBODY ELSE wrote and then follows — vertex ids into the graph / the
// slot map / the bitmap, block edges into a layer, cache slots and staged-row numbers into the feature rows — goes through
// PG_IDX(i, bnd, which, kernel, site): inside [0, bnd.n[which]) it is the index; outside, the FIRST offender of the process is
// recorded (kernel, site, value, bound, block; a count of all of them) and the access is redirected to element 0, so the launch
// completes and pg_bounds_report() can say which kernel followed which bad index — instead of a hipErrorIllegalAddress
// somewhere behind it (the reference has one assert, storage.py:149, and no bounds checks on ids).
// Bounds the C-ABI does not carry come from a registry of buffer extents (pg_bounds_region, filled by _lib.ptr() for every
// tensor handed to the library): bounds_elems(p, elem_size) = elements from p to the end of the registered buffer holding p,
// or "unknown" (then the check is skipped). The product build compiles all of this away: Bnd is empty, PG_IDX(i, ...) is i.
enum {
  PG_K_FWD_ROWS = 1, PG_K_COMPOSE, PG_K_SX_SAMPLE, PG_K_SX_RELABEL, PG_K_BM_RANK, PG_K_T_KEYS, PG_K_T_BLOCK, PG_K_SPLIT,
  PG_K_GATHER, PG_K_LABELS, PG_K_LINEAR_ROWS, PG_K_BWD_W_ROWS, PG_K_HEAD, PG_K_BWD_GATHER, PG_K_SPMM_FWD, PG_K_SPMM_BWD,
  PG_K_SCATTER, PG_K_COUNT_
};
#ifdef PG_BOUNDS
constexpr long long kBndUnknown = 0x7fffffffffffffffll;
struct Bnd {
  long long n[4] = {kBndUnknown, kBndUnknown, kBndUnknown, kBndUnknown};
};
struct BoundsRec {
  unsigned long long hit, kernel, site, value, bound, block, count, pad;
};
static __device__ BoundsRec g_bounds_rec;      // one per translation unit (no relocatable device code): see bounds_register
__device__ __forceinline__ long long bounds_idx(long long i, long long n, int kernel, int site) {
  if ((unsigned long long)i < (unsigned long long)n) return i;
  atomicAdd(&g_bounds_rec.count, 1ull);
  if (atomicCAS(&g_bounds_rec.hit, 0ull, 1ull) == 0ull) {
    g_bounds_rec.kernel = (unsigned long long)kernel;
    g_bounds_rec.site = (unsigned long long)site;
    g_bounds_rec.value = (unsigned long long)i;
    g_bounds_rec.bound = (unsigned long long)n;
    g_bounds_rec.block = (unsigned long long)blockIdx.x;
  }
  return 0;
}
#define PG_IDX(i, bnd, which, kernel, site) \
  ((std::decay_t<decltype(i)>)pg::bounds_idx((long long)(i), (bnd).n[which], kernel, site))
typedef int (*bounds_collect_fn)(BoundsRec* out, int reset);
void bounds_register(bounds_collect_fn fn, const char* unit);        // pg_api.hip
long long bounds_elems(const void* p, size_t elem_size);              // pg_api.hip: kBndUnknown when p is in no registered buffer
static int bounds_collect_unit(BoundsRec* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bounds_rec), sizeof(BoundsRec)) != hipSuccess) return -1;
  if (reset) {
    BoundsRec z{};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bounds_rec), &z, sizeof(BoundsRec)) != hipSuccess) return -1;
  }
  return 0;
}
static const int g_bounds_registered = (bounds_register(&bounds_collect_unit, __BASE_FILE__), 0);
inline Bnd bnd(long long a = kBndUnknown, long long b = kBndUnknown, long long c = kBndUnknown, long long d = kBndUnknown) {
  Bnd r;
  r.n[0] = a; r.n[1] = b; r.n[2] = c; r.n[3] = d;
  return r;
}
#else
struct Bnd {};
#define PG_IDX(i, bnd, which, kernel, site) (i)
inline long long bounds_elems(const void*, size_t) { return 0; }
inline Bnd bnd(long long = 0, long long = 0, long long = 0, long long = 0) { return Bnd{}; }
#endif

// a slot of a pg_row_source_t through the debug build's checks: >= 0 a cache row (bound [1]), <= -3 a staged row (bound [2])
__device__ __forceinline__ int32_t bnd_slot(int32_t sl, const Bnd& b, int kernel, int site) {
#ifdef PG_BOUNDS
  if (sl >= 0) return PG_IDX(sl, b, 1, kernel, site);
  if (sl <= -3) return -(int32_t)PG_IDX((long long)(-sl - 3), b, 2, kernel, site + 1) - 3;
#endif
  return sl;
}

inline int hip_fail(hipError_t e) {
  g_last_hip_error = (int)e;
  return PG_ERR_HIP;
}

#define PG_HIP(expr)                               \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) return pg::hip_fail(_e); \
  } while (0)

#define PG_LAUNCH_CHECK() PG_HIP(hipGetLastError())

inline hipStream_t as_stream(pg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T>
__host__ __device__ inline T ceil_div(T a, T b) {
  return (a + b - 1) / b;
}

// Philox4x32-10 (Salmon et al., SC'11): the counter-based RNG of the sampler
// spec. Same arithmetic as oracle/pgc_oracle.c (independent restatement).
struct Philox {
  static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  static constexpr uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  __host__ __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)M0 * c[0];
    const uint64_t p1 = (uint64_t)M1 * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
  }
  __host__ __device__ static inline void gen(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                             uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
    uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      round(c, k0, k1);
      k0 += W0;
      k1 += W1;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  }
};

// ---- the model's dropout folded into the kernels that consume its output (mask spec: pg_spmm.hip, include/pagraph_hip.h) ----
struct DropArgs {
  uint32_t thr, tag, k0, k1;
  const uint64_t* step;
  float scale;
  uint32_t step_imm;     // the step value when step == nullptr (pg_dropout_t.step_value)
};
__device__ __forceinline__ uint32_t drop_step_of(const DropArgs& d) { return d.step ? (uint32_t)*d.step : d.step_imm; }

__device__ __forceinline__ float4 drop_apply(float4 x, const uint32_t (&o)[4], int half, uint32_t thr, float scale) {
  const uint32_t w0 = half ? o[2] : o[0], w1 = half ? o[3] : o[1];
  x.x = (w0 & 0xffffu) >= thr ? x.x * scale : 0.f;
  x.y = (w0 >> 16) >= thr ? x.y * scale : 0.f;
  x.z = (w1 & 0xffffu) >= thr ? x.z * scale : 0.f;
  x.w = (w1 >> 16) >= thr ? x.w * scale : 0.f;
  return x;
}


inline bool drop_args(const pg_dropout_t* dp, DropArgs* d) {
  if (!dp || dp->threshold == 0 || dp->threshold > 65535u) return false;
  d->thr = dp->threshold;
  d->tag = dp->tag;
  d->k0 = (uint32_t)dp->seed;
  d->k1 = (uint32_t)(dp->seed >> 32);
  d->step = dp->step;
  d->step_imm = (uint32_t)dp->step_value;
  d->scale = 65536.f / (float)(65536u - dp->threshold);
  return true;
}


// How the aggregated row leaves the CU (store_mode is wave-uniform: a scalar branch around three stores).
//   PG_STORE_PLAIN: write-back — the 23 MB of `out` a launch produces sit dirty in the XCDs' L2s until the kernel's closing
//                   release writes them back, AFTER the last wave has retired: time the kernel's own stamps do not see but
//                   the dispatch (rocprofv3's End, and the successor's start) pays (profiles/r04).
//   PG_STORE_NT:    non-temporal hint (still write-back).
//   PG_STORE_WT:    `sc0 sc1` write-through: every piece goes to memory while the row loads of other waves stream, nothing is
//                   left dirty for the kernel's end; the consumer (another launch, usually on another XCD) reads from
//                   memory / MALL either way.
enum { PG_STORE_PLAIN = 0, PG_STORE_NT = 1, PG_STORE_WT = 2 };
typedef float pg_f4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_row_piece(float4* p, const float4& v, int mode) {
  if (mode == PG_STORE_WT) {
    pg_f4v t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(t) : "memory");
  } else if (mode == PG_STORE_NT) {
    pg_f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<pg_f4v*>(p));
  } else {
    *p = v;
  }
}


// how k_spmm_fwd_rows_w writes `out` (see store_row_piece): write-through (round 4: dispatch End - Start in the training loop
// 20.5 (plain) / 19.2 (wt) / 21.5 (nt) us, alone on cold rows 16.1 / 15.2 / 15.0, profiles/r04/fused_store_modes.txt)
inline int fwd_rows_store_mode() { return PG_STORE_WT; }


// uniform integer in [0, n) from a 64-bit draw: floor(r * n / 2^64)
__host__ __device__ inline uint64_t bounded(uint64_t r, uint64_t n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(r, n);
#else
  return (uint64_t)(((unsigned __int128)r * n) >> 64);
#endif
}

}  // namespace pg
