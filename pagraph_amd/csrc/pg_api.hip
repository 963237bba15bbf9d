// Host-side entry points that are not kernels: version/errors, the server-side
// row gather of the miss path (storage.py:128), HIP-event timer for bench.py.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <fcntl.h>

#include "pg_common.h"

#include <map>
#include <mutex>

namespace pg {
thread_local int g_last_hip_error = 0;
thread_local ProfSucc g_prof_succ;

#ifdef PG_BOUNDS
namespace {
struct BoundsState {
  std::mutex m;
  std::vector<std::pair<bounds_collect_fn, const char*>> units;
  std::map<uintptr_t, size_t> regions;      // base -> bytes
};
BoundsState& bounds_state() {
  static BoundsState* s = new BoundsState;   // never destroyed: units register from static initialisers of other objects
  return *s;
}
}  // namespace
void bounds_register(bounds_collect_fn fn, const char* unit) {
  BoundsState& b = bounds_state();
  std::lock_guard<std::mutex> l(b.m);
  b.units.emplace_back(fn, unit);
}
long long bounds_elems(const void* p, size_t elem_size) {
  if (!p || !elem_size) return kBndUnknown;
  BoundsState& b = bounds_state();
  std::lock_guard<std::mutex> l(b.m);
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  auto it = b.regions.upper_bound(a);
  if (it == b.regions.begin()) return kBndUnknown;
  --it;
  if (a >= it->first + it->second) return kBndUnknown;
  return (long long)((it->first + it->second - a) / elem_size);
}
#endif
}

// PG_NATIVE_BACKTRACE=1: a native back-trace of the faulting thread on SIGSEGV / SIGBUS / SIGABRT / SIGFPE, written straight
// to stderr (async-signal-safe calls only), then the previous disposition runs (Python's faulthandler installs itself on
// top of this one and hands the signal down). For the one core dump of round 3 that left nothing but a banner.
namespace {
struct sigaction g_prev_sa[NSIG];
int g_bt_fd = -1;      // PG_NATIVE_BACKTRACE_FILE: a second copy that survives a test runner's capture of stderr
void fault_handler(int sig, siginfo_t* info, void* ctx) {
  static const char head[] = "\n[libpagraph_hip] fatal signal, native back-trace of the faulting thread:\n";
  (void)!write(2, head, sizeof(head) - 1);
  void* frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  if (g_bt_fd >= 0) {
    (void)!write(g_bt_fd, head, sizeof(head) - 1);
    backtrace_symbols_fd(frames, n, g_bt_fd);
  }
  const struct sigaction& prev = g_prev_sa[sig];
  if (prev.sa_flags & SA_SIGINFO) {
    if (prev.sa_sigaction) { prev.sa_sigaction(sig, info, ctx); return; }
  } else if (prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN) {
    prev.sa_handler(sig);
    return;
  }
  signal(sig, SIG_DFL);
  raise(sig);
}
struct FaultInit {
  FaultInit() {
    const char* e = getenv("PG_NATIVE_BACKTRACE");
    if (!e || !atoi(e)) return;
    void* warm[2];
    (void)backtrace(warm, 2);                       // loads libgcc now, not inside the handler
    if (const char* f = getenv("PG_NATIVE_BACKTRACE_FILE")) g_bt_fd = open(f, O_WRONLY | O_CREAT | O_APPEND, 0644);
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_sigaction = fault_handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    sigemptyset(&sa.sa_mask);
    for (int sig : {SIGSEGV, SIGBUS, SIGABRT, SIGFPE}) sigaction(sig, &sa, &g_prev_sa[sig]);
  }
} g_fault_init;
}  // namespace


extern "C" {

int pg_version(void) { return 100; /* 0.1.0 */ }

const char* pg_strerror(int code) {
  switch (code) {
    case PG_OK: return "ok";
    case PG_ERR_INVALID: return "invalid argument";
    case PG_ERR_HIP: return "HIP runtime error";
    case PG_ERR_NOMEM: return "out of memory";
    case PG_ERR_UNSUPPORTED: return "unsupported";
    case PG_ERR_OVERFLOW: return "capacity overflow";
    default: return "unknown error";
  }
}

int pg_last_hip_error(void) { return pg::g_last_hip_error; }

int pg_bounds_enabled(void) {
#ifdef PG_BOUNDS
  return 1;
#else
  return 0;
#endif
}

int pg_bounds_region(const void* base, int64_t bytes) {
#ifdef PG_BOUNDS
  if (!base) return PG_OK;
  auto& b = pg::bounds_state();
  std::lock_guard<std::mutex> l(b.m);
  const uintptr_t a = reinterpret_cast<uintptr_t>(base);
  if (bytes <= 0) {
    b.regions.erase(a);
    return PG_OK;
  }
  // a new registration replaces every older one it overlaps (the allocator handed the memory to somebody else since)
  auto it = b.regions.lower_bound(a);
  if (it != b.regions.begin()) {
    auto prev = std::prev(it);
    if (prev->first + prev->second > a) b.regions.erase(prev);
  }
  while (it != b.regions.end() && it->first < a + (uintptr_t)bytes) it = b.regions.erase(it);
  b.regions[a] = (size_t)bytes;
  return PG_OK;
#else
  (void)base; (void)bytes;
  return PG_OK;
#endif
}

int pg_bounds_report(uint64_t* rec, char* unit, int32_t unit_len, int32_t reset) {
#ifdef PG_BOUNDS
  if (!rec) return PG_ERR_INVALID;
  for (int i = 0; i < 8; ++i) rec[i] = 0;
  if (unit && unit_len > 0) unit[0] = 0;
  PG_HIP(hipDeviceSynchronize());
  auto& b = pg::bounds_state();
  std::lock_guard<std::mutex> l(b.m);
  for (auto& u : b.units) {
    pg::BoundsRec r{};
    if (u.first(&r, reset) != 0) return PG_ERR_HIP;
    rec[6] += r.count;
    if (r.hit && !rec[0]) {
      rec[0] = 1; rec[1] = r.kernel; rec[2] = r.site; rec[3] = r.value; rec[4] = r.bound; rec[5] = r.block;
      if (unit && unit_len > 0) {
        strncpy(unit, u.second ? u.second : "?", (size_t)unit_len - 1);
        unit[unit_len - 1] = 0;
      }
    }
  }
  return PG_OK;
#else
  (void)rec; (void)unit; (void)unit_len; (void)reset;
  return PG_ERR_UNSUPPORTED;
#endif
}

int pg_device_cu_count(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return -1;
  return p.multiProcessorCount;
}

// storage.py:128 `self.graph._node_frame._frame[name].data[nids]` — the CPU
// fancy-index over the host feature table. Rows are scattered in DRAM, so the
// copy is split over threads; `staged` is normally pinned so the following
// hipMemcpyAsync is a true async DMA (the reference's pageable copy is not).
int pg_host_gather_rows(const float* table, int64_t table_stride, int32_t dim, const int64_t* fullids,
                        int64_t n, float* staged, int n_threads) {
  if (n < 0 || dim <= 0 || table_stride < dim) return PG_ERR_INVALID;
  if (n == 0) return PG_OK;
  if (!table || !fullids || !staged) return PG_ERR_INVALID;
  const size_t row_bytes = (size_t)dim * sizeof(float);
  auto work = [&](int64_t lo, int64_t hi) {
    for (int64_t j = lo; j < hi; ++j)
      std::memcpy(staged + j * dim, table + fullids[j] * table_stride, row_bytes);
  };
  int t = std::max(1, n_threads);
  t = (int)std::min<int64_t>(t, (n + 255) / 256);
  if (t <= 1) {
    work(0, n);
    return PG_OK;
  }
  std::vector<std::thread> th;
  th.reserve(t - 1);
  const int64_t per = (n + t - 1) / t;
  for (int i = 1; i < t; ++i) th.emplace_back(work, std::min<int64_t>(n, i * per), std::min<int64_t>(n, (i + 1) * per));
  work(0, std::min<int64_t>(n, per));
  for (auto& x : th) x.join();
  return PG_OK;
}

int pg_timer_create(pg_timer_t** t) {
  if (!t) return PG_ERR_INVALID;
  pg_timer* p = new (std::nothrow) pg_timer;
  if (!p) return PG_ERR_NOMEM;
  if (hipEventCreate(&p->start) != hipSuccess || hipEventCreate(&p->stop) != hipSuccess) {
    delete p;
    return PG_ERR_HIP;
  }
  *t = p;
  return PG_OK;
}
int pg_timer_destroy(pg_timer_t* t) {
  if (!t) return PG_OK;
  (void)hipEventDestroy(t->start);
  (void)hipEventDestroy(t->stop);
  delete t;
  return PG_OK;
}
int pg_timer_start(pg_timer_t* t, pg_stream_t s) {
  if (!t) return PG_ERR_INVALID;
  PG_HIP(hipEventRecord(t->start, pg::as_stream(s)));
  return PG_OK;
}
int pg_timer_stop(pg_timer_t* t, pg_stream_t s) {
  if (!t) return PG_ERR_INVALID;
  PG_HIP(hipEventRecord(t->stop, pg::as_stream(s)));
  return PG_OK;
}
int pg_timer_elapsed_ms(pg_timer_t* t, float* ms) {
  if (!t || !ms) return PG_ERR_INVALID;
  PG_HIP(hipEventSynchronize(t->stop));
  PG_HIP(hipEventElapsedTime(ms, t->start, t->stop));
  return PG_OK;
}

}  // extern "C"
