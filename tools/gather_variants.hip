// Standalone tuning harness for the cached-feature gather (not part of the library).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gather_variants.hip -o tools/gather_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int kWave = 64;
typedef float vf4 __attribute__((ext_vector_type(4)));

// ---------------- V0: current library kernel shape (wave owns RPW rows, U rows in flight) -------------
template <int RPW, int U, bool NT>
__global__ __launch_bounds__(256) void v0(const int64_t* __restrict__ ids, const int32_t* __restrict__ slot_map,
                                          const float* __restrict__ cache, const float* __restrict__ cnorm,
                                          float* __restrict__ out, float* __restrict__ onorm, int64_t n, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t row0 = wave * RPW;
  if (row0 >= n) return;
  const int64_t my = row0 + lane;
  const bool valid = lane < RPW && my < n;
  int32_t slot = -2;
  if (valid) { slot = slot_map[ids[my]]; }
  if (valid && slot >= 0) onorm[my] = cnorm[slot];
  const int rows = (int)((n - row0) < RPW ? (n - row0) : RPW);
  const int pieces = dim / 4;
  for (int j = 0; j < rows; j += U) {
    int32_t s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = (j + u < rows) ? __builtin_amdgcn_readlane(slot, j + u) : -1;
    for (int c = lane; c < pieces; c += 64) {
      vf4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) if (s[u] >= 0) v[u] = reinterpret_cast<const vf4*>(cache + (int64_t)s[u] * dim)[c];
#pragma unroll
      for (int u = 0; u < U; ++u) if (s[u] >= 0) {
        vf4* d = reinterpret_cast<vf4*>(out + (row0 + j + u) * dim) + c;
        if (NT) __builtin_nontemporal_store(v[u], d); else *d = v[u];
      }
    }
  }
}

// ---------------- V1: flat (row, piece) mapping: every lane busy, block tile of T rows ----------------
template <int T, int U, bool NT>
__global__ __launch_bounds__(256) void v1(const int64_t* __restrict__ ids, const int32_t* __restrict__ slot_map,
                                          const float* __restrict__ cache, const float* __restrict__ cnorm,
                                          float* __restrict__ out, float* __restrict__ onorm, int64_t n, int dim,
                                          uint32_t magic /* ceil(2^32 / pieces) */) {
  __shared__ int32_t s_slot[T];
  const int64_t row0 = (int64_t)blockIdx.x * T;
  const int rows = (int)((n - row0) < T ? (n - row0) : T);
  if (threadIdx.x < rows) {
    const int32_t s = slot_map[ids[row0 + threadIdx.x]];
    s_slot[threadIdx.x] = s;
    if (s >= 0) onorm[row0 + threadIdx.x] = cnorm[s];
  }
  __syncthreads();
  const int pieces = dim / 4;
  const int total = rows * pieces;
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * U) {
    vf4 v[U];
    int r[U], c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 256;
      r[u] = (int)(((uint64_t)i * magic) >> 32);
      c[u] = i - r[u] * pieces;
      if (i < total) {
        const int32_t s = s_slot[r[u]];
        if (s >= 0) v[u] = reinterpret_cast<const vf4*>(cache + (int64_t)s * dim)[c[u]];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 256;
      if (i < total && s_slot[r[u]] >= 0) {
        vf4* d = reinterpret_cast<vf4*>(out + (row0 + r[u]) * dim) + c[u];
        if (NT) __builtin_nontemporal_store(v[u], d); else *d = v[u];
      }
    }
  }
}

// ---------------- V2: one row per wave-iteration, 3 loads of the row in flight, grid-stride over rows --
template <bool NT>
__global__ __launch_bounds__(256) void v2(const int64_t* __restrict__ ids, const int32_t* __restrict__ slot_map,
                                          const float* __restrict__ cache, const float* __restrict__ cnorm,
                                          float* __restrict__ out, float* __restrict__ onorm, int64_t n, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  const int pieces = dim / 4;
  for (int64_t base = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; base < n; base += nw * 64) {
    const int64_t my = base + lane;
    int32_t slot = -2;
    if (my < n) { slot = slot_map[ids[my]]; if (slot >= 0) onorm[my] = cnorm[slot]; }
    const int rows = (int)((n - base) < 64 ? (n - base) : 64);
    for (int j = 0; j < rows; j += 2) {
      const int32_t s0 = __builtin_amdgcn_readlane(slot, j);
      const int32_t s1 = (j + 1 < rows) ? __builtin_amdgcn_readlane(slot, j + 1) : -1;
      const vf4* p0 = reinterpret_cast<const vf4*>(cache + (int64_t)(s0 < 0 ? 0 : s0) * dim);
      const vf4* p1 = reinterpret_cast<const vf4*>(cache + (int64_t)(s1 < 0 ? 0 : s1) * dim);
      vf4 a0, a1, a2, b0, b1, b2;
      const bool l2 = lane + 128 < pieces;
      if (s0 >= 0) { a0 = p0[lane]; a1 = p0[lane + 64]; if (l2) a2 = p0[lane + 128]; }
      if (s1 >= 0) { b0 = p1[lane]; b1 = p1[lane + 64]; if (l2) b2 = p1[lane + 128]; }
      vf4* d0 = reinterpret_cast<vf4*>(out + (base + j) * dim);
      vf4* d1 = reinterpret_cast<vf4*>(out + (base + j + 1) * dim);
      if (s0 >= 0) { d0[lane] = a0; d0[lane + 64] = a1; if (l2) d0[lane + 128] = a2; }
      if (s1 >= 0) { d1[lane] = b0; d1[lane + 64] = b1; if (l2) d1[lane + 128] = b2; }
    }
  }
}

// ---------------- V3: flat mapping, incremental (row,piece) arithmetic, NT load/store options ---------
template <int T, int U, bool NTS, bool NTL, int BS = 256>
__global__ __launch_bounds__(BS) void v3(const int64_t* __restrict__ ids, const int32_t* __restrict__ slot_map,
                                          const float* __restrict__ cache, const float* __restrict__ cnorm,
                                          float* __restrict__ out, float* __restrict__ onorm, int64_t n, int dim) {
  __shared__ int32_t s_slot[T];
  const int64_t row0 = (int64_t)blockIdx.x * T;
  const int rows = (int)((n - row0) < T ? (n - row0) : T);
  if (threadIdx.x < rows) {
    const int32_t s = slot_map[ids[row0 + threadIdx.x]];
    s_slot[threadIdx.x] = s;
    if (s >= 0) onorm[row0 + threadIdx.x] = cnorm[s];
  }
  __syncthreads();
  const int pieces = dim / 4;
  const int total = rows * pieces;
  const int step_r = BS / pieces, step_c = BS % pieces;   // advance of (r,c) per +256 flat positions
  int r = threadIdx.x / pieces, c = threadIdx.x % pieces;
  vf4* obase = reinterpret_cast<vf4*>(out + row0 * dim);
  for (int i0 = threadIdx.x; i0 < total; i0 += BS * U) {
    vf4 v[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * BS;
      ok[u] = false;
      if (i < total) {
        const int32_t s = s_slot[r];
        if (s >= 0) {
          const vf4* p = reinterpret_cast<const vf4*>(cache + (int64_t)s * dim) + c;
          v[u] = NTL ? __builtin_nontemporal_load(p) : *p;
          ok[u] = true;
        }
      }
      r += step_r; c += step_c;
      if (c >= pieces) { c -= pieces; ++r; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ok[u]) {
        vf4* d = obase + (i0 + u * BS);
        if (NTS) __builtin_nontemporal_store(v[u], d); else *d = v[u];
      }
    }
  }
}

// ---------------- V4: persistent blocks, next tile's slots prefetched while the current tile copies -----
template <int T, int U, bool NTS>
__global__ __launch_bounds__(256) void v4(const int64_t* __restrict__ ids, const int32_t* __restrict__ slot_map,
                                          const float* __restrict__ cache, const float* __restrict__ cnorm,
                                          float* __restrict__ out, float* __restrict__ onorm, int64_t n, int dim) {
  __shared__ int32_t s_slot[2][T];
  const int pieces = dim / 4;
  const int step_r = 256 / pieces, step_c = 256 % pieces;
  const int64_t ntiles = (n + T - 1) / T;
  int buf = 0;
  int64_t tile = blockIdx.x;
  if (tile < ntiles && threadIdx.x < T) {
    const int64_t row = tile * T + threadIdx.x;
    int32_t s = -1;
    if (row < n) { s = slot_map[ids[row]]; if (s >= 0) onorm[row] = cnorm[s]; }
    s_slot[0][threadIdx.x] = s;
  }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * T;
    const int rows = (int)((n - row0) < T ? (n - row0) : T);
    const int64_t nt = tile + gridDim.x;
    int32_t ns = -1;
    int64_t nrow = nt * T + threadIdx.x;
    const bool pf = nt < ntiles && threadIdx.x < T && nrow < n;
    if (pf) ns = slot_map[ids[nrow]];
    const int total = rows * pieces;
    int r = threadIdx.x / pieces, c = threadIdx.x % pieces;
    vf4* obase = reinterpret_cast<vf4*>(out + row0 * dim);
    for (int i0 = threadIdx.x; i0 < total; i0 += 256 * U) {
      vf4 v[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * 256;
        ok[u] = false;
        if (i < total) {
          const int32_t s = s_slot[buf][r];
          if (s >= 0) { v[u] = *(reinterpret_cast<const vf4*>(cache + (int64_t)s * dim) + c); ok[u] = true; }
        }
        r += step_r; c += step_c;
        if (c >= pieces) { c -= pieces; ++r; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) if (ok[u]) {
        vf4* d = obase + (i0 + u * 256);
        if (NTS) __builtin_nontemporal_store(v[u], d); else *d = v[u];
      }
    }
    if (threadIdx.x < T) {
      s_slot[buf ^ 1][threadIdx.x] = ns;
      if (pf && ns >= 0) onorm[nrow] = cnorm[ns];
    }
    buf ^= 1;
    __syncthreads();
  }
}

// better copy ceiling: 4 independent 16-B loads in flight per lane, optional nontemporal
template <bool NT>
__global__ __launch_bounds__(256) void copy4x4(const vf4* __restrict__ a, vf4* __restrict__ b, int64_t n4) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    vf4 v0 = NT ? __builtin_nontemporal_load(a + i) : a[i];
    vf4 v1 = NT ? __builtin_nontemporal_load(a + i + stride) : a[i + stride];
    vf4 v2 = NT ? __builtin_nontemporal_load(a + i + 2 * stride) : a[i + 2 * stride];
    vf4 v3 = NT ? __builtin_nontemporal_load(a + i + 3 * stride) : a[i + 3 * stride];
    if (NT) { __builtin_nontemporal_store(v0, b + i); __builtin_nontemporal_store(v1, b + i + stride);
              __builtin_nontemporal_store(v2, b + i + 2 * stride); __builtin_nontemporal_store(v3, b + i + 3 * stride); }
    else { b[i] = v0; b[i + stride] = v1; b[i + 2 * stride] = v2; b[i + 3 * stride] = v3; }
  }
  for (; i < n4; i += stride) b[i] = a[i];
}

// ---------------- reference: plain vf4 copy of the same number of bytes -----------------------------
__global__ __launch_bounds__(256) void copy4(const vf4* __restrict__ a, vf4* __restrict__ b, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) b[i] = a[i];
}

int main(int argc, char** argv) {
  const int dim = 600;
  const int64_t NC = 2560000;                      // cached rows (30% of 8.5M)
  std::vector<int64_t> Rs = {18700, 42000};
  float *cache, *cnorm, *out, *onorm;
  int32_t* slot_map; int64_t* ids;
  const int64_t Rmax = 1 << 20;
  CK(hipMalloc(&cache, NC * dim * 4)); CK(hipMalloc(&cnorm, NC * 4));
  CK(hipMalloc(&out, Rmax * dim * 4)); CK(hipMalloc(&onorm, Rmax * 4));
  CK(hipMalloc(&slot_map, NC * 4)); CK(hipMalloc(&ids, Rmax * 8));
  CK(hipMemset(cache, 1, NC * dim * 4)); CK(hipMemset(cnorm, 1, NC * 4));
  std::vector<int32_t> sm(NC); for (int64_t i = 0; i < NC; ++i) sm[i] = (int32_t)((i * 7919) % NC);
  CK(hipMemcpy(slot_map, sm.data(), NC * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::mt19937_64 rng(0);
  for (int dist = 0; dist < 2; ++dist) {
    std::vector<int64_t> h(Rmax);
    for (auto& x : h) {
      double u = (rng() >> 11) * (1.0 / 9007199254740992.0);
      x = dist == 0 ? (int64_t)(u * NC) : (int64_t)(u * u * u * u * NC);   // uniform / skewed to hot rows
      if (x >= NC) x = NC - 1;
    }
    CK(hipMemcpy(ids, h.data(), Rmax * 8, hipMemcpyHostToDevice));
    for (int64_t R : Rs) {
      const double bytes = (double)R * (8.0 * (dim + 1) + 17);
      auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        const int reps = 20;
        float best = 1e9, tot = 0;
        for (int i = 0; i < reps; ++i) {
          CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("dist=%s R=%8ld %-22s avg %8.1f us  %7.0f GB/s (%.1f%% of 8TB/s)  best %7.0f GB/s\n", dist ? "skew" : "unif",
               (long)R, name, tot / reps * 1e3, bytes / (tot / reps) / 1e6, bytes / (tot / reps) / 1e6 / 80.0, bytes / best / 1e6);
      };
      const int pieces = dim / 4;
      const uint32_t magic = (uint32_t)(((1ull << 32) + pieces - 1) / pieces);
      run("copy4(same bytes)", [&] { hipLaunchKernelGGL(copy4, dim3(4096), dim3(256), 0, 0, (const vf4*)cache, (vf4*)out, R * dim / 4); });
#define V0(RPW, U, NT) run("v0<" #RPW "," #U "," #NT ">", [&] { hipLaunchKernelGGL((v0<RPW, U, NT>), dim3((unsigned)((R + RPW * 4 - 1) / (RPW * 4))), dim3(256), 0, 0, ids, slot_map, cache, cnorm, out, onorm, R, dim); })
#define V3(T, U, NTS, NTL) run("v3<" #T "," #U "," #NTS "," #NTL ">", [&] { hipLaunchKernelGGL((v3<T, U, NTS, NTL>), dim3((unsigned)((R + T - 1) / T)), dim3(256), 0, 0, ids, slot_map, cache, cnorm, out, onorm, R, dim); })
#define V3B(T, U, BS) run("v3<" #T "," #U ",nts,BS=" #BS ">", [&] { hipLaunchKernelGGL((v3<T, U, true, false, BS>), dim3((unsigned)((R + T - 1) / T)), dim3(BS), 0, 0, ids, slot_map, cache, cnorm, out, onorm, R, dim); })
      V3(8, 4, true, false); V3(8, 5, true, false); V3(4, 3, true, false); V3(6, 4, true, false); V3(10, 6, true, false); V3(8, 3, true, false); V3(8, 2, true, false);
      V3(5, 3, true, false); V3(7, 4, true, false); V3(2, 2, true, false); V3(3, 2, true, false);
#define V4(T, U, G) run("v4<" #T "," #U ",grid=" #G ">", [&] { hipLaunchKernelGGL((v4<T, U, true>), dim3(G), dim3(256), 0, 0, ids, slot_map, cache, cnorm, out, onorm, R, dim); })
      V4(4, 3, 2048); V4(4, 3, 1024); V4(2, 2, 2048); V4(8, 4, 1024); V4(4, 3, 4096);
      V3B(4, 5, 128); V3B(8, 5, 128); V3B(4, 3, 128); V3B(16, 5, 512); V3B(8, 3, 512); V3B(16, 3, 512); V3B(3, 4, 128); V3B(2, 5, 64); V3B(1, 3, 64);
    }
  }
  return 0;
}
