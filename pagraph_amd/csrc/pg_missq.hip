// Asynchronous miss path: "cache misses fall through to a pinned-host hipMemcpyAsync on a side
// stream overlapped with compute" (north star) — the reference's get_feat_from_server
// (PaGraph/storage/storage.py:117-131: CPU fancy-index over the shared-memory table + H2D copy)
// moved off the trainer's critical path.
//
// Why not just read the host table from a kernel (pg_scatter_rows_from_host)?  It works and needs no
// host thread, but while a kernel has PCIe reads in flight every kernel boundary of the OTHER streams
// gets ~1.2 us slower (measured, tools/exp_overlap2.py: a 380 us zero-copy scatter next to a chain of
// 90 small dependent kernels costs 488 us instead of 380; the same low-occupancy kernel reading HBM
// overlaps perfectly). The copy engines (SDMA) do not have that side effect.
//
// Per slot (one per in-flight minibatch):
//   GPU, load stream : k_split/k_gather write the miss list (positions -> HBM, full ids -> pinned
//                      host) ; k_publish copies the miss count to pinned host and raises a flag.
//   worker thread    : waits for the flag, gathers table[fullid] rows into a pinned staging buffer
//                      with a small thread pool, enqueues hipMemcpyAsync(staging -> HBM) and the
//                      row-scatter kernel on its own copy stream, records an event.
//   consumer         : pg_missq_wait(slot, stream) makes the compute stream wait on that event.
// The trainer thread never blocks on the GPU; it blocks in pg_missq_wait only if the worker has not
// yet *enqueued* the copy of a batch that was submitted a whole step earlier.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <cerrno>
#include <unistd.h>

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/amd_hsa_signal.h>

#include <pthread.h>
#include <sched.h>

#include "pg_common.h"

namespace pg {

__global__ void k_publish(const int32_t* __restrict__ count_dev, int32_t* count_host, uint32_t* flag_host,
                          uint32_t seq) {
  *count_host = *count_dev;
  __threadfence_system();
  __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// k_publish for a launch whose split ran with index dedup: every dup entry's "earlier row" becomes that row's
// staged index (the split wrote -(j + 3) into the primary's slot), then the count is published as above
// (round 3) ... and the repeat's OWN slot entry takes the primary's value: a consumer that reads rows in place through
// the slot array (pg_spmm_fwd_rows, pg_linear_fwd (rows in place): no frame the repeat could be copied into) then finds the
// primary's staged row. A repeat is never a primary, so the entries read and the entries written are disjoint.
__global__ __launch_bounds__(256) void k_publish_dedup(const int32_t* __restrict__ count_dev, int32_t* count_host,
                                                       uint32_t* flag_host, uint32_t seq, int32_t* slots,
                                                       int32_t* __restrict__ dup_src,
                                                       const int32_t* __restrict__ dup_pos,
                                                       const int32_t* __restrict__ dup_count) {
  const int32_t nd = *dup_count;
  for (int32_t k = threadIdx.x; k < nd; k += blockDim.x) {
    const int32_t s = slots[dup_src[k]];
    dup_src[k] = s <= -3 ? -s - 3 : -1;
    if (s <= -3) slots[dup_pos[k]] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    *count_host = *count_dev;
    __threadfence_system();
    __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// copy stream -> consumer stream hand-off WITHOUT the host: the worker's copy stream raises `landed`
// after the scatter kernel; the consumer stream runs a one-wave kernel that sleeps on it. The trainer
// thread therefore never waits for the worker (a host-side wait made the whole pipeline settle in a
// slow "just in time" cycle: the host enqueued compute(k) — and with it everything downstream — only
// after batch k's copy had been enqueued; measured 1.1 ms/step instead of 0.4).
__global__ void k_signal(uint32_t* landed, uint32_t seq) {
  __hip_atomic_store(landed, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void k_wait_landed(const uint32_t* landed, uint32_t seq, uint32_t* timed_out) {
  const unsigned long long t0 = wall_clock64();   // 100 MHz
  while ((int32_t)(__hip_atomic_load(landed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > 300000000ull) {     // 3 s: the worker died; do not hang the GPU
      timed_out[2] = seq;                         // [2]: the sequence number the consumer gave up on
      timed_out[0] = 1;
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// The host->device copy of the miss rows is submitted straight to ONE chosen SDMA engine through ROCr
// (hsa_amd_memory_async_copy_on_engine) instead of hipMemcpyAsync: the HIP runtime re-picks "the lowest free
// engine" for a stream every ~170 commands, and when engine 0 happens to be busy at that moment (always, once the
// step is PCIe bound) it moves the stream to engine 1 for the next ~170 copies — which reads host memory at 23 GB/s
// instead of 53 GB/s on this part (measured per copy, AMD_LOG_MASK=0x300: copy_engine=0x1 -> 53-54 GB/s,
// copy_engine=0x2 -> 23.0 GB/s, strictly bimodal). That was the "slow mode" of the pipeline: a GraphSAGE step
// 0.40 -> 0.90 ms, a GCN step 0.20 -> 0.36 ms for a few hundred steps at a time. The copy's completion signal is
// an amd_signal_t in GPU-visible system memory; the copy stream waits for it with a one-wave kernel.
//
// The signal lives in HOST memory, so every poll is a PCIe read, and a kernel with PCIe reads in flight makes every
// kernel boundary of the other streams ~1 us slower (tools/exp_overlap2.py): polling back to back cost the GCN
// step 0.197 -> 0.204 ms. `poll_sleeps` x ~3.9 us between polls (default 3: the read is in flight ~10 % of the
// time, the copy's completion is seen ~6 us late on average — 1.5 % of a 20 MB copy).
__global__ void k_wait_hsa_signal(const volatile int64_t* value, uint32_t* timed_out, int poll_sleeps) {
  const unsigned long long t0 = wall_clock64();   // 100 MHz
  while (__hip_atomic_load(const_cast<const int64_t*>(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) > 0) {
    for (int i = 0; i < poll_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    if (wall_clock64() - t0 > 300000000ull) {     // 3 s
      timed_out[1] = 1;                           // [1]: the copy engine's completion signal never came
      timed_out[0] = 1;
      break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// Consumer-side wait for a job whose rows are all read IN PLACE from the staged block (no scatter: the fused
// gather+aggregate path of the GCN loop): nothing of such a job is put on the copy stream. The worker raises
// `issued` (pinned host word) once every field's copy has been handed to the SDMA engine — each copy's completion
// signal set to 1 beforehand — and the consumer's one-wave kernel waits for `issued` >= seq, then for the signals to
// drop. No kernel is parked on the copy stream (a wait-for-SDMA kernel there holds back whatever shares its hardware
// queue — when that was the load stream's k_publish, gather(k+1) could not overlap copy(k) and the whole run sat at
// 0.279 ms/step, ~3 % of runs), no k_signal, no event, and no packet that a barrier could overtake.
struct DirectSignals {
  const volatile int64_t* v[PG_MAX_FIELDS];
};

__global__ void k_wait_direct(const uint32_t* issued, uint32_t seq, const DirectSignals sg, uint32_t* timed_out,
                              int poll_sleeps) {
  const unsigned long long t0 = wall_clock64();   // 100 MHz
  bool ok = true;
  // (both words live in host memory: every poll is a PCIe read, and a kernel with PCIe reads in flight makes the kernel
  // boundaries of the other streams slower — poll sparsely; the copy that follows the issue takes >= 100 us anyway)
  while ((int32_t)(__hip_atomic_load(issued, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
    for (int i = 0; i < poll_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    if (wall_clock64() - t0 > 300000000ull) { ok = false; break; }
  }
#pragma unroll
  for (int f = 0; f < PG_MAX_FIELDS; ++f) {
    if (!sg.v[f]) continue;
    while (ok && __hip_atomic_load(const_cast<const int64_t*>(sg.v[f]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) > 0) {
      for (int i = 0; i < poll_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
      if (wall_clock64() - t0 > 300000000ull) ok = false;
    }
  }
  if (!ok) {
    timed_out[2] = seq;
    timed_out[0] = 1;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// One row of the host table into the pinned staging buffer. The staging buffer is only ever read by the copy
// engine, so the stores go around the cache (movntps): no read-for-ownership of the destination lines and
// no eviction of the gather threads' working set. Falls back to memcpy for rows that are not 16-byte multiples.
static inline void copy_row_stream(float* dst, const float* src, size_t bytes) {
  if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 15) == 0) {
    typedef float v4f __attribute__((vector_size(16)));
    const v4f* s = reinterpret_cast<const v4f*>(src);
    v4f* d = reinterpret_cast<v4f*>(dst);
    const size_t n = bytes / 16;
    for (size_t i = 0; i < n; ++i) __builtin_nontemporal_store(s[i], d + i);
  } else {
    std::memcpy(dst, src, bytes);
  }
}

// Persistent thread pool for the miss path's row gather: parallel_for over [0, n) in chunks of kChunk rows
// that the threads CLAIM from an atomic counter. A job arrives every ~0.15 ms and is ~50-100 us of work, so
// how fast the threads start matters as much as how fast they copy: after a job they spin on the generation
// counter for PG_MISSQ_SPIN_US microseconds (default 100) before they sleep on a condition variable. Keep the
// thread count well under the CPU quota of the process (the GPU boxes give 16 CPUs: 31 spinning threads
// starved the launch thread and the step got 2x slower; 8 are fine).
//
// Stragglers (round 3). A pool thread that has claimed a chunk and then loses its CPU used to hold the whole job —
// and with it the training step that needs the rows — until the scheduler ran it again: 2-7 ms stalls of the launch
// loop, once or twice per epoch on a busy host (DESIGN "Where a host stall sits"). A chunk is an idempotent copy, so
// the caller now RE-EXECUTES chunks that are claimed but not finished kOverdueUs after it ran out of chunks to claim
// itself; whoever finishes a chunk first counts it. What makes that safe:
//  * a job lives in a descriptor of its own (a small ring inside the pool), not on the caller's stack: a straggler
//    that wakes up after parallel_for has returned still finds its function object, bounds and counters intact;
//  * every thread brackets its time inside a job with the descriptor's `active` count; a descriptor is recycled only
//    when that count is zero, and parallel_for returns a ticket with which the caller asks "is anybody still
//    writing on behalf of that job?" (quiesced) before IT recycles what the job wrote to — the miss queue's
//    worker does so before it reuses a slot's staging buffer, ring-size jobs later.
class Pool {
 public:
  static constexpr int64_t kChunk = 32;
  static constexpr int kRing = 256;           // job descriptors: one is reused 256 jobs (tens of ms) after its job — a thread
                                              // that lost its CPU with a chunk in hand (8-9 ms at a time on the shared boxes)
                                              // has long left by then; with 16 the caller waited for it 2.4 ms later
  static constexpr int64_t kMaxChunks = 1 << 16;
  struct Job {
    std::function<void(int64_t, int64_t)> fn;
    int64_t total = 0, chunks = 0;
    std::atomic<int64_t> left{0}, done{0};
    std::atomic<int> active{0};               // threads currently inside work(job)
    std::atomic<uint64_t> ticket{0};          // generation this descriptor currently serves
    std::unique_ptr<std::atomic<uint8_t>[]> state;   // per chunk: 0 unclaimed, 1 claimed, 2 done
    int64_t state_cap = 0;
  };
  explicit Pool(int n) : n_(n < 1 ? 1 : n) {
    spin_us_ = 100;
    overdue_us_ = 40;
    // test hook: PG_MISSQ_TEST_STALL=<every>,<us> — a pool thread (never the caller) that claims every <every>-th
    // chunk sleeps <us> microseconds before it executes it, like a thread that lost its CPU with a chunk in hand
    if (const char* t = getenv("PG_MISSQ_TEST_STALL")) {
      stall_every_ = atoi(t);
      if (const char* c = strchr(t, ',')) stall_us_ = atoi(c + 1);
    }
    const int stride = 0;      // (round 5 tried pinning gather thread i to a CPU: no steadier on a shared box — profiles/r05/host_gather_sweep.txt)
    for (int i = 1; i < n_; ++i) th_.emplace_back([this, i] {
      if (stride > 0) {
        cpu_set_t all, one;
        CPU_ZERO(&all);
        if (sched_getaffinity(0, sizeof(all), &all) == 0) {
          const int n = CPU_COUNT(&all);
          int want = n > 0 ? (i * stride) % n : -1, seen = 0;
          for (int c = 0; c < CPU_SETSIZE && want >= 0; ++c)
            if (CPU_ISSET(c, &all) && seen++ == want) {
              CPU_ZERO(&one);
              CPU_SET(c, &one);
              (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
              break;
            }
        }
      }
      loop();
    });
  }
  ~Pool() {
    stop_.store(true, std::memory_order_release);
    gen_.fetch_add(1, std::memory_order_release);
    {
      std::lock_guard<std::mutex> l(m_);
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  // runs f over [0, n) in chunks; returns a ticket for quiesced(). One caller at a time.
  uint64_t parallel_for(int64_t n, std::function<void(int64_t, int64_t)> f) {
    if (n <= 0) return 0;
    const int64_t chunks = (n + kChunk - 1) / kChunk;
    if (chunks <= 1 || n_ == 1 || chunks > kMaxChunks) {
      f(0, n);
      return 0;
    }
    const uint64_t ticket = ++tickets_;
    Job& j = ring_[ticket % kRing];
    // the descriptor's previous job (kRing jobs ago) may still have a straggler inside: it must leave first
    while (j.active.load(std::memory_order_acquire) != 0) cpu_relax();
    if (j.state_cap < chunks) {
      j.state.reset(new std::atomic<uint8_t>[(size_t)chunks]);
      j.state_cap = chunks;
    }
    for (int64_t c = 0; c < chunks; ++c) j.state[c].store(0, std::memory_order_relaxed);
    j.fn = std::move(f);
    j.total = n;
    j.chunks = chunks;
    j.done.store(0, std::memory_order_relaxed);
    j.ticket.store(ticket, std::memory_order_relaxed);
    // publish order: descriptor, then its claim counter, then the pool's current job, then the generation (workers
    // acquire on left: a claim that succeeds sees the whole descriptor)
    j.left.store(chunks, std::memory_order_release);
    cur_.store(&j, std::memory_order_release);
    gen_.fetch_add(1, std::memory_order_release);
    if (sleepers_.load(std::memory_order_acquire) > 0) {
      {
        std::lock_guard<std::mutex> l(m_);
      }
      cv_.notify_all();
    }
    work(j, true);
    // nothing left to claim. Chunks still open are in the hands of other threads: normally a few microseconds from
    // done; a thread that lost its CPU in the middle of one can be away for a scheduler time slice
    const auto t0 = std::chrono::steady_clock::now();
    int polls = 0;
    while (j.done.load(std::memory_order_acquire) < chunks) {
      cpu_relax();
      if ((++polls & 31) == 0 &&
          std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= overdue_us_) {
        for (int64_t c = 0; c < chunks; ++c) {
          if (j.state[c].load(std::memory_order_acquire) == 2) continue;
          const int64_t lo = c * kChunk, hi = std::min<int64_t>(n, lo + kChunk);
          j.fn(lo, hi);
          if (j.state[c].exchange(2, std::memory_order_acq_rel) != 2) {
            j.done.fetch_add(1, std::memory_order_acq_rel);
            rescued_.fetch_add(1, std::memory_order_relaxed);
          }
        }
        break;
      }
    }
    while (j.done.load(std::memory_order_acquire) < chunks) cpu_relax();   // (every chunk is counted exactly once)
    return ticket;
  }
  // true when no thread is still executing on behalf of the job with this ticket (a re-executed chunk's original
  // owner may still be copying the same bytes): what the job wrote to may be reused
  bool quiesced(uint64_t ticket) const {
    if (ticket == 0) return true;
    const Job& j = ring_[ticket % kRing];
    if (j.ticket.load(std::memory_order_acquire) != ticket) return true;   // recycled since: it had drained then
    return j.active.load(std::memory_order_acquire) == 0;
  }
  void wait_quiesced(uint64_t ticket) const {
    while (!quiesced(ticket)) cpu_relax();
  }
  int64_t rescued() const { return rescued_.load(std::memory_order_relaxed); }

 private:
  static inline void cpu_relax() { __builtin_ia32_pause(); }
  void work(Job& j, bool caller = false) {
    j.active.fetch_add(1, std::memory_order_acq_rel);
    for (;;) {
      const int64_t c = j.left.fetch_sub(1, std::memory_order_acq_rel) - 1;
      if (c < 0) break;       // (the counter counts DOWN: a late claim is invalid by its sign alone)
      uint8_t expect = 0;
      if (!j.state[c].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) continue;   // rescued already
      if (stall_every_ > 0 && !caller && stall_count_.fetch_add(1, std::memory_order_relaxed) % stall_every_ == 0)
        std::this_thread::sleep_for(std::chrono::microseconds(stall_us_));
      const int64_t lo = c * kChunk, hi = std::min<int64_t>(j.total, lo + kChunk);
      j.fn(lo, hi);
      if (j.state[c].exchange(2, std::memory_order_acq_rel) != 2) j.done.fetch_add(1, std::memory_order_acq_rel);
    }
    j.active.fetch_sub(1, std::memory_order_acq_rel);
  }
  void loop() {
    uint64_t seen = gen_.load(std::memory_order_acquire);
    for (;;) {
      const auto t0 = std::chrono::steady_clock::now();
      uint64_t g;
      int polls = 0;
      while ((g = gen_.load(std::memory_order_acquire)) == seen) {
        cpu_relax();
        if ((++polls & 63) == 0 &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >=
                spin_us_) {
          std::unique_lock<std::mutex> l(m_);
          sleepers_.fetch_add(1, std::memory_order_acq_rel);
          cv_.wait(l, [&] { return gen_.load(std::memory_order_acquire) != seen; });
          sleepers_.fetch_sub(1, std::memory_order_acq_rel);
        }
      }
      seen = g;
      if (stop_.load(std::memory_order_acquire)) return;
      Job* j = cur_.load(std::memory_order_acquire);
      if (j) work(*j, false);
    }
  }
  int n_;
  int spin_us_ = 100, overdue_us_ = 40;
  int stall_every_ = 0, stall_us_ = 0;
  std::atomic<int64_t> stall_count_{1};
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  Job ring_[kRing];
  std::atomic<Job*> cur_{nullptr};
  uint64_t tickets_ = 0;
  std::atomic<int64_t> rescued_{0};
  std::atomic<uint64_t> gen_{0};
  std::atomic<int> sleepers_{0};
  std::atomic<bool> stop_{false};
};

}  // namespace pg

using namespace pg;

struct pg_missq_slot {
  int64_t* fullid_h = nullptr;   // pinned [max_rows]
  int32_t* count_h = nullptr;    // pinned
  uint32_t* flag_h = nullptr;    // pinned
  int32_t* pos_d = nullptr;      // device [max_rows]
  int32_t* count_d = nullptr;    // device
  int32_t* dup_pos_d = nullptr;  // device [max_rows]: rows that repeat an earlier missed id (index dedup) ...
  int32_t* dup_src_d = nullptr;  // device [max_rows]: ... and the earlier row, resolved to its staged row at publish
  int32_t* dup_count_d = nullptr;
  bool dedup = false;            // this submission carries a dup list
  uint32_t* issued_h = nullptr;  // pinned: last sequence number whose copies are all with the SDMA engine (direct jobs)
  bool direct = false;           // this submission: every wanted field is staged-only -> nothing goes on the copy stream
  uint32_t* landed_d = nullptr;  // device: last sequence number whose rows are in place (k_signal)
  float* staging_h[PG_MAX_FIELDS] = {nullptr};  // pinned [max_rows * dim]
  float* staged_d[PG_MAX_FIELDS] = {nullptr};   // device [max_rows * dim]
  hipEvent_t filled = nullptr;
  hipEvent_t cp0 = nullptr, cp1 = nullptr;   // PG_MISSQ_COPYLOG: brackets of the slot's last H2D copy
  hsa_signal_t sig[PG_MAX_FIELDS] = {};      // direct SDMA path: completion signal of field f's copy
  int cp_field = -1;
  int64_t cp_bytes = 0;
  int share = 256;         // cpu_share the latest submission was made under
  hipEvent_t tail_go = nullptr, tail_done = nullptr;   // pg_missq_device_tail on the copy stream: split done / tail rows landed
  bool tail_pending = false;
  uint64_t gather_ticket[PG_MAX_FIELDS] = {0};   // Pool ticket of the last CPU gather into staging_h[f]
  uint32_t submitted = 0;  // last sequence number handed to the worker (trainer thread)
  uint32_t done = 0;       // last sequence number whose copy has been enqueued (worker, under mutex)
  uint32_t done_pub = 0;   // the same, published with release order for lock-free polling (pg_missq_wait_idle)
  int32_t last_count = 0;
  float* out[PG_MAX_FIELDS] = {nullptr};
  int32_t out_stride[PG_MAX_FIELDS] = {0};   // -1 with out == NULL: copy to the staging block only
  int32_t pos_lo[PG_MAX_FIELDS] = {0};       // rows below it are not scattered
  std::chrono::steady_clock::time_point t_submit;
};

struct pg_missq {
  int device = 0, n_slots = 0, n_fields = 0;
  int64_t max_rows = 0;
  pg_missq_field_t fields[PG_MAX_FIELDS];
  int32_t sstride[PG_MAX_FIELDS] = {0};   // floats per staged / staging row: wide rows padded to whole 16-byte pieces
  std::vector<pg_missq_slot> slots;
  hipStream_t copy_stream = nullptr;
  uint32_t* timeout_d = nullptr;
  Pool* pool = nullptr;
  // staging buffers out of rotation: spares, and buffers a straggler of the gather pool may still write into (with the
  // ticket of that gather). Worker thread only.
  std::vector<std::pair<float*, uint64_t>> parked[PG_MAX_FIELDS];
  int64_t n_spared = 0;              // jobs that took a spare because a straggler still held their slot's buffer
  std::thread worker;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::deque<std::pair<int, uint32_t>> jobs;
  bool stop = false;
  int error = PG_OK;
  int64_t n_wait_event = 0, n_wait_spin = 0;   // how pg_missq_wait_device ordered the consumer (under m)
  bool wait_value = false;       // (round 3's hipStreamWaitValue variant of the wait: kept in the code, never selected)
  std::atomic<int> cpu_share{256};   // of 256: the leading share of every miss list that the CPU path moves
  // PG_MISSQ_DEBUG=1: accumulated worker phase times (us) printed at destroy
  double t_sync = 0, t_flag = 0, t_gather = 0, t_enqueue = 0, t_copy = 0, t_sub2flag = 0, t_sub2pop = 0;
  // the longest single occurrence of each phase since the last pg_missq_stats(reset_max): where a stall of the miss path sat
  double mx_flag = 0, mx_gather = 0, mx_enqueue = 0, mx_total = 0;
  double t_total = 0;
  bool copy_log = getenv("PG_MISSQ_COPYLOG") != nullptr;
  // direct SDMA path (see k_wait_hsa_signal)
  bool hsa_ok = false;
  bool hsa_inited = false;        // this queue holds a reference on the HSA runtime (hsa_init / hsa_shut_down)
  hsa_agent_t gpu_agent = {}, cpu_agent = {};
  uint32_t engine = 0;            // hsa_amd_sdma_engine_id_t bit
  double engine_GBps[16] = {0};   // calibration at creation: host->device rate of every engine that reported free
  uint64_t ts_freq = 0;
  std::vector<std::pair<int64_t, float>> copies;   // (bytes, ms) per job, resolved when the slot is reused
  int64_t n_jobs = 0, n_rows = 0;
};

// ---- direct SDMA path: agents, engine calibration, per-slot signals -------------------------------------------
static bool hsa_copy_sync(pg_missq* q, void* dst, const void* src, size_t bytes, uint32_t engine, hsa_signal_t sig,
                          double* seconds) {
  hsa_signal_store_relaxed(sig, 1);
  const auto t0 = std::chrono::steady_clock::now();
  if (hsa_amd_memory_async_copy_on_engine(dst, q->gpu_agent, src, q->cpu_agent, bytes, 0, nullptr, sig,
                                          (hsa_amd_sdma_engine_id_t)engine, true) != HSA_STATUS_SUCCESS)
    return false;
  // bounded wait (1 s): an engine that never completes must not hang creation
  if (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, 1000000000ull, HSA_WAIT_STATE_ACTIVE) > 0) return false;
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return true;
}

// The wait kernels poll amd_signal_t::value of a completion signal directly (ROCr's user-mode signal layout,
// amd_hsa_signal.h). Checked once instead of assumed: the signal must be a plain user signal and a store through the API
// must be visible at that address — anything else (an IPC / doorbell signal kind, a different layout) disables the direct
// path and the runtime's hipMemcpyAsync is used.
static bool signal_layout_ok(hsa_signal_t sig) {
  if (!sig.handle) return false;
  const amd_signal_t* a = reinterpret_cast<const amd_signal_t*>(sig.handle);
  if (a->kind != AMD_SIGNAL_KIND_USER) return false;
  hsa_signal_store_screlease(sig, 0x5a17);
  const bool ok = __atomic_load_n(&a->value, __ATOMIC_ACQUIRE) == 0x5a17 && hsa_signal_load_scacquire(sig) == 0x5a17;
  hsa_signal_store_screlease(sig, 0);
  return ok;
}

// Ranks of one node calibrate one at a time (an advisory lock on a file in /tmp, held for the ~10 ms per engine the timed
// copies take): eight ranks timing 16 MiB host->device copies at the same moment share the host's memory system, and a
// mis-read calibration is exactly what put round 2's run on a 29 GB/s engine. PG_MISSQ_CALIB_LOCK=0 turns it off.
struct CalibLock {
  int fd = -1;
  CalibLock() {
    const char* e = getenv("PG_MISSQ_CALIB_LOCK");
    if (e && atoi(e) == 0) return;
    // a per-user lock file (another user's ranks calibrate another job's GPUs; their file would not be ours to open),
    // never followed through a symlink, mode independent of the umask
    char path[96];
    snprintf(path, sizeof(path), "/tmp/.pagraph_sdma_calibration.%u.lock", (unsigned)geteuid());
    fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC | O_NOFOLLOW, 0600);
    if (fd < 0) {
      fprintf(stderr, "[pg_missq] calibration lock %s: %s — calibrating without it\n", path, strerror(errno));
      return;
    }
    (void)fchmod(fd, 0600);
    // bounded: a stopped or hung rank that holds the lock must not block every other rank's queue creation for ever. A
    // calibration takes ~10 ms per engine; after 5 s this rank goes ahead without it.
    const long budget_ms = 5000;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      if (flock(fd, LOCK_EX | LOCK_NB) == 0) return;
      const bool busy = errno == EWOULDBLOCK || errno == EINTR;
      const long waited = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (!busy || waited >= budget_ms) {
        fprintf(stderr, "[pg_missq] calibration lock %s not acquired after %ld ms (%s) — calibrating without it\n", path,
                waited, busy ? "held by another process" : strerror(errno));
        close(fd);
        fd = -1;
        return;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  }
  ~CalibLock() {
    if (fd >= 0) {
      (void)flock(fd, LOCK_UN);
      close(fd);
    }
  }
};

static void hsa_copy_init(pg_missq* q) {
  const char* e = getenv("PG_MISSQ_HSA_COPY");
  if (e && atoi(e) == 0) return;
  if (hsa_init() != HSA_STATUS_SUCCESS) return;   // reference counted; the HIP runtime holds its own; paired in missq_free
  q->hsa_inited = true;
  pg_missq_slot& s0 = q->slots[0];
  hsa_amd_pointer_info_t pi_d, pi_h;
  memset(&pi_d, 0, sizeof(pi_d)); memset(&pi_h, 0, sizeof(pi_h));
  pi_d.size = sizeof(pi_d); pi_h.size = sizeof(pi_h);
  if (hsa_amd_pointer_info(s0.staged_d[0], &pi_d, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS ||
      hsa_amd_pointer_info(s0.staging_h[0], &pi_h, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS)
    return;
  if (pi_d.type == HSA_EXT_POINTER_TYPE_UNKNOWN || pi_h.type == HSA_EXT_POINTER_TYPE_UNKNOWN) return;
  q->gpu_agent = pi_d.agentOwner;
  q->cpu_agent = pi_h.agentOwner;
  hsa_device_type_t tg, tc;
  if (hsa_agent_get_info(q->gpu_agent, HSA_AGENT_INFO_DEVICE, &tg) != HSA_STATUS_SUCCESS || tg != HSA_DEVICE_TYPE_GPU) return;
  if (hsa_agent_get_info(q->cpu_agent, HSA_AGENT_INFO_DEVICE, &tc) != HSA_STATUS_SUCCESS || tc != HSA_DEVICE_TYPE_CPU) return;
  for (auto& s : q->slots)
    for (int f = 0; f < q->n_fields; ++f)
      if (hsa_signal_create(0, 0, nullptr, &s.sig[f]) != HSA_STATUS_SUCCESS) return;
  for (auto& s : q->slots)
    for (int f = 0; f < q->n_fields; ++f)
      if (!signal_layout_ok(s.sig[f])) return;
  (void)hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &q->ts_freq);
  CalibLock calib_lock;
  uint32_t mask = 0;
  if (hsa_amd_memory_copy_engine_status(q->gpu_agent, q->cpu_agent, &mask) != HSA_STATUS_SUCCESS || mask == 0) return;
  // calibrate: the widest field's staging buffer of slot 0, up to 16 MiB, twice per engine (first = warm-up)
  int fw = 0;
  for (int f = 1; f < q->n_fields; ++f)
    if (q->fields[f].dim > q->fields[fw].dim) fw = f;
  const size_t bytes = std::min<size_t>((size_t)q->max_rows * q->sstride[fw] * sizeof(float), (size_t)16 << 20);
  double best = 0;
  for (int b = 0; b < 16; ++b) {
    if (!((mask >> b) & 1u)) continue;
    double sec = 0;
    if (!hsa_copy_sync(q, s0.staged_d[fw], s0.staging_h[fw], bytes, 1u << b, s0.sig[0], nullptr)) continue;
    if (!hsa_copy_sync(q, s0.staged_d[fw], s0.staging_h[fw], bytes, 1u << b, s0.sig[0], &sec) || sec <= 0) continue;
    q->engine_GBps[b] = bytes / sec / 1e9;
    if (q->engine_GBps[b] > best) best = q->engine_GBps[b];
  }
  // the LOWEST engine id within 10 % of the best rate. Engines 0-3 all calibrate at ~55 GB/s on their own, but only
  // engine 0 keeps that rate inside the training loop: a run whose calibration happened to read 53.3 for engine 0 and
  // 55.5 for engine 1 (a 3 % tie rule let engine 1 win) sat at 0.278 ms/step — gather + a 29 GB/s copy — from
  // start to end; that was the "metastable" mode seen in ~3 % of runs.
  for (int b = 0; b < 16 && best > 0; ++b)
    if (q->engine_GBps[b] >= 0.9 * best) {
      q->engine = 1u << b;
      break;
    }
  if (!q->engine) return;
  if (q->copy_log) (void)hsa_amd_profiling_async_copy_enable(true);
  for (auto& s : q->slots)
    for (int f = 0; f < q->n_fields; ++f) hsa_signal_store_relaxed(s.sig[f], 0);
  q->hsa_ok = true;
  if (getenv("PG_MISSQ_DEBUG")) {
    const char* rk = getenv("LOCAL_RANK");
    fprintf(stderr, "[missq] local rank %s device %d: direct SDMA copies on engine mask 0x%x; host->device GB/s per engine:",
            rk ? rk : "-", q->device, q->engine);
    for (int b = 0; b < 16; ++b)
      if (q->engine_GBps[b] > 0) fprintf(stderr, " %d:%.1f", b, q->engine_GBps[b]);
    fprintf(stderr, "\n");
  }
}

static void missq_worker(pg_missq* q) {
  if (hipSetDevice(q->device) != hipSuccess) {
    std::lock_guard<std::mutex> l(q->m);
    q->error = PG_ERR_HIP;
  }
  for (;;) {
    std::pair<int, uint32_t> job;
    {
      std::unique_lock<std::mutex> l(q->m);
      q->cv_job.wait(l, [q] { return q->stop || !q->jobs.empty(); });
      if (q->stop && q->jobs.empty()) return;
      job = q->jobs.front();
      q->jobs.pop_front();
    }
    pg_missq_slot& s = q->slots[job.first];
    int rc = PG_OK;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::micro>(b - a).count();
    };
    const auto t0 = now();
    // the previous copy out of this slot's staging buffers must have drained (4 steps ago: a formality)
    if (hipEventSynchronize(s.filled) != hipSuccess) rc = PG_ERR_HIP;
    if (q->hsa_ok)     // ... and a direct job's copies are in no stream: their completion signals tell
      for (int f = 0; f < q->n_fields; ++f)
        if (hsa_signal_wait_scacquire(s.sig[f], HSA_SIGNAL_CONDITION_LT, 1, 3000000000ull, HSA_WAIT_STATE_ACTIVE) > 0)
          rc = PG_ERR_HIP;
    if (q->copy_log && s.cp_bytes > 0 && q->hsa_ok && s.cp_field >= 0) {
      hsa_amd_profiling_async_copy_time_t t;
      if (hsa_amd_profiling_get_async_copy_time(s.sig[s.cp_field], &t) == HSA_STATUS_SUCCESS && q->ts_freq) {
        std::lock_guard<std::mutex> l(q->m);
        q->copies.emplace_back(s.cp_bytes, (float)((double)(t.end - t.start) * 1e3 / (double)q->ts_freq));
      }
      s.cp_bytes = 0;
    } else if (q->copy_log && s.cp_bytes > 0) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, s.cp0, s.cp1) == hipSuccess) {
        std::lock_guard<std::mutex> l(q->m);
        q->copies.emplace_back(s.cp_bytes, ms);
      }
      s.cp_bytes = 0;
    }
    const auto t1 = now();
    // wait for the GPU to publish the miss list of this submission
    // poll: the list arrives 50-150 us after the job was queued. sleep_for(5 us) really sleeps ~55 us (timer
    // slack), a whole extra stage of latency per minibatch, so spin first and only fall back to sleeping
    // when nothing has come for a millisecond (idle queue).
    for (int64_t spins = 0; __atomic_load_n(s.flag_h, __ATOMIC_ACQUIRE) != job.second; ++spins) {
      if ((spins & 1023) == 1023) {
        {
          std::lock_guard<std::mutex> l(q->m);
          if (q->stop) return;
        }
        if (us(t1, now()) > 1000.0) std::this_thread::sleep_for(std::chrono::microseconds(50));
      }
      __builtin_ia32_pause();
    }
    const int64_t m_all = *s.count_h;
    // rows [0, m) are this worker's; the tail is read by the device itself (pg_scatter_rows_from_host_tail)
    const int64_t m = m_all * s.share / 256;   // the share this SUBMISSION was made under (its device tail used the same)
    const auto t2 = now();
    q->t_sub2flag += us(s.t_submit, t2);
    q->t_sub2pop += us(s.t_submit, t0);
    double tg = 0, te = 0;
    if (m > 0 && rc == PG_OK) {
      for (int f = 0; f < q->n_fields && rc == PG_OK; ++f) {
        if (!s.out[f] && s.out_stride[f] != -1) continue;   // this launch did not ask for the field
        const pg_missq_field_t& fd = q->fields[f];
        const size_t row_bytes = (size_t)fd.dim * sizeof(float);
        const size_t srow = (size_t)q->sstride[f];                     // staged row stride (floats)
        const size_t copy_bytes = (size_t)m * srow * sizeof(float);    // what crosses PCIe (padding included: < 1 %)
        float* stg = s.staging_h[f];
        const int64_t* ids = s.fullid_h;
        // prefetch: 6 rows ahead, the whole row (up to 4 KB). Round 4: a
        // thread that only had the head (256 bytes) of the row after next on its way spent its time waiting for DRAM — with
        // six whole rows in flight per thread the same gather takes 80 us instead of 100-114 on 12 threads, 119-135 on 6
        // (the step stays on its PCIe floor with half the threads), 280 instead of 508 on 2 (profiles/r04/host_gather_sweep.txt).
        constexpr int pf_dist = 6;
        constexpr size_t pf_cfg = 4096;
        const size_t pf_bytes = std::min(pf_cfg, row_bytes);
        const auto ta = now();
        // a straggler of this slot's PREVIOUS job (a re-executed chunk's original owner) may still be copying into
        // this staging buffer: it must have left before the buffer is rewritten. Normally a formality (n_slots jobs ago);
        // a thread that lost its CPU stays away for 8-9 ms on the shared boxes — sixty jobs — and the job that waited
        // for it here WAS the stall (round 3: `longest_us.cpu_gather` 9 ms in 3 of 6 epochs). So the buffer is left to
        // the straggler and the job takes a spare one; the old buffer comes back once the straggler has gone.
        if (!q->pool->quiesced(s.gather_ticket[f])) {
          float* spare = nullptr;
          for (auto& pk : q->parked[f])
            if (pk.first && q->pool->quiesced(pk.second)) {
              spare = pk.first;
              pk = {s.staging_h[f], s.gather_ticket[f]};
              break;
            }
          if (spare) {
            s.staging_h[f] = stg = spare;
            s.gather_ticket[f] = 0;
            ++q->n_spared;
          } else {
            q->pool->wait_quiesced(s.gather_ticket[f]);     // every spare is held by a straggler too
          }
        }
        // everything captured BY VALUE: a straggler may run this after the call has returned
        const float* table = fd.table;
        const int64_t tstride = fd.table_stride;
        const int pfd = pf_dist;
        s.gather_ticket[f] = q->pool->parallel_for(m, [=](int64_t lo, int64_t hi) {   // storage.py:128 table[nids]
          for (int64_t j = lo; j < hi && j < lo + pfd; ++j) {     // the chunk's first rows: all on their way before the first copy
            const char* nx = reinterpret_cast<const char*>(table + ids[j] * tstride);
            for (size_t b = 0; b < pf_bytes; b += 64) __builtin_prefetch(nx + b, 0, 0);
          }
          for (int64_t j = lo; j < hi; ++j) {
            if (j + pfd < hi) {   // rows are random DRAM pages: start a later one while this one streams
              const char* nx = reinterpret_cast<const char*>(table + ids[j + pfd] * tstride);
              for (size_t b = 0; b < pf_bytes; b += 64) __builtin_prefetch(nx + b, 0, 0);
            }
            copy_row_stream(stg + j * srow, table + ids[j] * tstride, row_bytes);
          }
          __builtin_ia32_sfence();   // the streaming stores must be globally visible before the chunk counts as done
        });
        const auto tb = now();
        tg += us(ta, tb);
        const bool logc = q->copy_log && row_bytes >= 256;
        bool direct = false;
        if (q->hsa_ok) {
          // straight to the calibrated SDMA engine; the copy stream then waits for the completion signal on the device
          hsa_signal_store_relaxed(s.sig[f], 1);
          direct = hsa_amd_memory_async_copy_on_engine(s.staged_d[f], q->gpu_agent, stg, q->cpu_agent, copy_bytes,
                                                       0, nullptr, s.sig[f], (hsa_amd_sdma_engine_id_t)q->engine,
                                                       true) == HSA_STATUS_SUCCESS;
          if (direct && s.direct) {
            // staged-only job: the consumer's own wait kernel watches this signal (k_wait_direct)
            if (logc) {
              s.cp_bytes = (int64_t)copy_bytes;
              s.cp_field = f;
            }
          } else if (direct) {
            const volatile int64_t* val = &reinterpret_cast<amd_signal_t*>(s.sig[f].handle)->value;
            constexpr int poll_sleeps = 3;
            hipLaunchKernelGGL(k_wait_hsa_signal, dim3(1), dim3(1), 0, q->copy_stream, val, q->timeout_d, poll_sleeps);
            if (hipGetLastError() != hipSuccess) rc = PG_ERR_HIP;
            if (logc) {
              s.cp_bytes = (int64_t)copy_bytes;
              s.cp_field = f;
            }
          } else {
            hsa_signal_store_relaxed(s.sig[f], 0);
            q->hsa_ok = false;   // ROCr refused: back to the runtime's copy for good
          }
        }
        if (!direct) {
          if (logc) (void)hipEventRecord(s.cp0, q->copy_stream);
          if (hipMemcpyAsync(s.staged_d[f], stg, copy_bytes, hipMemcpyHostToDevice, q->copy_stream) !=
              hipSuccess)
            rc = PG_ERR_HIP;
          if (logc) {
            (void)hipEventRecord(s.cp1, q->copy_stream);
            s.cp_bytes = (int64_t)copy_bytes;
            s.cp_field = -1;
          }
        }
        // (a direct job's rows are scattered by its consumer, on the consumer's stream: scatter_on_consumer)
        if (rc == PG_OK && s.out[f] && !s.direct)
          rc = pg_scatter_rows_strided(s.staged_d[f], (int32_t)srow, s.pos_d, nullptr, m, nullptr, fd.dim, s.out[f],
                                       s.out_stride[f], s.pos_lo[f], 0, (pg_stream_t)q->copy_stream);   // storage.py:199-200
        if (rc == PG_OK && s.out[f] && s.dedup && !s.direct)    // repeats of a missed id: copied on the device, never over PCIe
          rc = pg_scatter_rows_strided(s.staged_d[f], (int32_t)srow, s.dup_pos_d, s.dup_src_d, q->max_rows, s.dup_count_d,
                                       fd.dim, s.out[f], s.out_stride[f], s.pos_lo[f], 64, (pg_stream_t)q->copy_stream);
        te += us(tb, now());
      }
    }
    if (s.direct) {
      // a field whose copy fell back to the runtime's hipMemcpyAsync (ROCr refused the engine) is waited for here,
      // on the host: the consumer only watches the SDMA signals
      if (!q->hsa_ok && hipStreamSynchronize(q->copy_stream) != hipSuccess) rc = PG_ERR_HIP;
      __atomic_store_n(s.issued_h, job.second, __ATOMIC_RELEASE);
    } else {
      if (q->wait_value) {
        if (hipStreamWriteValue32(q->copy_stream, s.landed_d, job.second, 0) != hipSuccess) rc = PG_ERR_HIP;
      } else
        hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, q->copy_stream, s.landed_d, job.second);
      if (hipGetLastError() != hipSuccess) rc = PG_ERR_HIP;
      if (hipEventRecord(s.filled, q->copy_stream) != hipSuccess) rc = PG_ERR_HIP;
    }
    static const bool dbg_copy = getenv("PG_MISSQ_DEBUG") && atoi(getenv("PG_MISSQ_DEBUG")) >= 2;
    if (dbg_copy) {
      const auto tc = now();
      (void)hipEventSynchronize(s.filled);
      q->t_copy += us(tc, now());
    }
    {
      std::lock_guard<std::mutex> l(q->m);
      s.done = job.second;
      __atomic_store_n(&s.done_pub, job.second, __ATOMIC_RELEASE);
      s.last_count = (int32_t)m;
      q->t_sync += us(t0, t1); q->t_flag += us(t1, t2); q->t_gather += tg; q->t_enqueue += te;
      q->n_jobs += 1; q->n_rows += m;
      const double tt = us(s.t_submit, now());
      q->t_total += tt;
      q->mx_flag = std::max(q->mx_flag, us(t1, t2)); q->mx_gather = std::max(q->mx_gather, tg);
      q->mx_enqueue = std::max(q->mx_enqueue, te); q->mx_total = std::max(q->mx_total, tt);
      if (rc != PG_OK) q->error = rc;
    }
    q->cv_done.notify_all();
  }
}

static void missq_free(pg_missq* q) {
  if (!q) return;
  if (getenv("PG_MISSQ_DEBUG") && q->n_jobs)
    fprintf(stderr, "[missq] jobs %ld rows/job %.0f | per job us: event-sync %.1f flag-wait %.1f cpu-gather %.1f enqueue %.1f copy-drain(dbg2) %.1f | submit->pop %.1f submit->flag %.1f\n",
            (long)q->n_jobs, (double)q->n_rows / q->n_jobs, q->t_sync / q->n_jobs, q->t_flag / q->n_jobs,
            q->t_gather / q->n_jobs, q->t_enqueue / q->n_jobs, q->t_copy / q->n_jobs, q->t_sub2pop / q->n_jobs, q->t_sub2flag / q->n_jobs);
  if (getenv("PG_MISSQ_DEBUG") && q->n_jobs)
    fprintf(stderr, "[missq] consumer waits: %ld by event (copy already enqueued), %ld by the spin kernel\n",
            (long)q->n_wait_event, (long)q->n_wait_spin);
  if (q->worker.joinable()) {
    {
      std::lock_guard<std::mutex> l(q->m);
      q->stop = true;
    }
    q->cv_job.notify_all();
    q->worker.join();
  }
  delete q->pool;        // joins the gather threads: nobody writes to the staging buffers any more
  q->pool = nullptr;
  for (auto& v : q->parked) {
    for (auto& pk : v) (void)hipHostFree(pk.first);
    v.clear();
  }
  // direct jobs' copies are in no HIP stream: let the engine finish before the buffers go (bounded: 1 s each). Every
  // signal that was created is waited for, also after a mid-run fall-back to hipMemcpyAsync (hsa_ok false by then):
  // copies handed to the engine before the fall-back may still be in flight
  for (auto& s : q->slots)
    for (int f = 0; f < PG_MAX_FIELDS; ++f)
      if (s.sig[f].handle)
        (void)hsa_signal_wait_scacquire(s.sig[f], HSA_SIGNAL_CONDITION_LT, 1, 1000000000ull, HSA_WAIT_STATE_BLOCKED);
  for (auto& s : q->slots) {
    (void)hipHostFree(s.issued_h);
    (void)hipHostFree(s.fullid_h);
    (void)hipHostFree(s.count_h);
    (void)hipHostFree(s.flag_h);
    (void)hipFree(s.pos_d);
    (void)hipFree(s.count_d);
    (void)hipFree(s.dup_pos_d);
    (void)hipFree(s.dup_src_d);
    (void)hipFree(s.dup_count_d);
    (void)hipFree(s.landed_d);
    for (int f = 0; f < PG_MAX_FIELDS; ++f) {
      (void)hipHostFree(s.staging_h[f]);
      (void)hipFree(s.staged_d[f]);
    }
    if (s.filled) (void)hipEventDestroy(s.filled);
    if (s.tail_go) (void)hipEventDestroy(s.tail_go);
    if (s.tail_done) (void)hipEventDestroy(s.tail_done);
    for (int f = 0; f < PG_MAX_FIELDS; ++f)
      if (s.sig[f].handle) (void)hsa_signal_destroy(s.sig[f]);
    if (s.cp0) (void)hipEventDestroy(s.cp0);
    if (s.cp1) (void)hipEventDestroy(s.cp1);
  }
  (void)hipFree(q->timeout_d);
  if (q->copy_stream) (void)hipStreamDestroy(q->copy_stream);
  if (q->hsa_inited) (void)hsa_shut_down();     // drops this queue's reference only (the HIP runtime keeps its own)
  delete q;
}

// the scatter of a direct job's miss rows (and of the repeats of its index dedup), enqueued by the consumer on ITS
// stream behind the wait for the copy (storage.py:199-200): the row count is read from the device (count_d)
static int scatter_on_consumer(pg_missq* q, pg_missq_slot& s, pg_stream_t stream) {
  for (int f = 0; f < q->n_fields; ++f) {
    if (!s.out[f]) continue;
    int rc = pg_scatter_rows_strided(s.staged_d[f], q->sstride[f], s.pos_d, nullptr, q->max_rows, s.count_d, q->fields[f].dim,
                                     s.out[f], s.out_stride[f], s.pos_lo[f], 0, stream);
    if (rc == PG_OK && s.dedup)
      rc = pg_scatter_rows_strided(s.staged_d[f], q->sstride[f], s.dup_pos_d, s.dup_src_d, q->max_rows, s.dup_count_d,
                                   q->fields[f].dim, s.out[f], s.out_stride[f], s.pos_lo[f], 64, stream);
    if (rc != PG_OK) return rc;
  }
  return PG_OK;
}

extern "C" {

int pg_missq_create(int device, int n_slots, int64_t max_rows, const pg_missq_field_t* fields, int n_fields,
                    int n_threads, pg_missq_t** out) {
  if (!out || n_slots <= 0 || n_slots > 64 || max_rows <= 0 || !fields || n_fields <= 0 || n_fields > PG_MAX_FIELDS)
    return PG_ERR_INVALID;
  for (int f = 0; f < n_fields; ++f)
    if (!fields[f].table || fields[f].dim <= 0 || fields[f].table_stride < fields[f].dim) return PG_ERR_INVALID;
  PG_HIP(hipSetDevice(device));
  pg_missq* q = new (std::nothrow) pg_missq;
  if (!q) return PG_ERR_NOMEM;
  q->device = device; q->n_slots = n_slots; q->n_fields = n_fields; q->max_rows = max_rows;
  for (int f = 0; f < n_fields; ++f) {
    q->fields[f] = fields[f];
    q->sstride[f] = fields[f].dim >= 16 ? ((fields[f].dim + 3) & ~3) : fields[f].dim;
  }
  q->slots.resize(n_slots);
  // The consumer (compute) stream may have a spin-wait kernel parked on it until this stream's k_signal has run. HIP
  // multiplexes streams of ONE priority class onto a few hardware queues, and a kernel behind a spinning kernel in the
  // same hardware queue never starts: with every stream at normal priority the pipeline deadlocked until the 3 s
  // time-out (measured: PG_PRIO_LOAD=0 PG_PRIO_SAMPLER=0). The copy stream therefore lives in the HIGH priority class,
  // whose hardware queues are separate from those of the normal-priority compute stream. (PG_PRIO_COPY overrides.)
  //
  // A class of its own is not enough, though: the sampler and load streams of the SAME pipeline (high priority as
  // well) wait — barrier packets — for events the compute stream records AFTER its spin kernel (ring slot free, frames
  // consumed). Once a process has made more high-priority streams than the class has hardware queues (a second
  // trainer: bench.py's reference-equivalent leg, the tail of a long pytest session) the copy stream shares a queue with
  // one of them, and if the worker enqueues copy(j) after such a barrier went in, copy(j) sits behind a barrier that
  // waits for the kernel that waits for copy(j): a 3 s stall per occurrence (measured: 90-1350 ms/step). The launch
  // thread therefore calls pg_missq_wait_idle(slot) BEFORE it enqueues any wait for an event recorded after the
  // consumer of that slot's previous job (GraphedTrainer.prepare, NeighborSampler._enqueue): by then copy(j) is in its
  // queue, ahead of the barrier — whatever the runtime multiplexes.
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  const char* pe = getenv("PG_PRIO_COPY");
  bool ok = hipStreamCreateWithPriority(&q->copy_stream, hipStreamNonBlocking, pe ? atoi(pe) : prio_hi) == hipSuccess;
  ok = ok && hipMalloc((void**)&q->timeout_d, 64) == hipSuccess && hipMemset(q->timeout_d, 0, 64) == hipSuccess;
  for (auto& s : q->slots) {
    ok = ok && hipHostMalloc((void**)&s.fullid_h, max_rows * 8, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&s.count_h, 64, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&s.flag_h, 64, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&s.issued_h, 64, hipHostMallocDefault) == hipSuccess;
    if (ok) *s.issued_h = 0;
    ok = ok && hipMalloc((void**)&s.pos_d, max_rows * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.count_d, 64) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.dup_pos_d, max_rows * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.dup_src_d, max_rows * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.dup_count_d, 64) == hipSuccess && hipMemset(s.dup_count_d, 0, 64) == hipSuccess;
    ok = ok && (q->wait_value ? hipExtMallocWithFlags((void**)&s.landed_d, 8, hipMallocSignalMemory)
                              : hipMalloc((void**)&s.landed_d, 64)) == hipSuccess;
    for (int f = 0; f < n_fields && ok; ++f) {
      const size_t bytes = (size_t)max_rows * q->sstride[f] * sizeof(float);
      ok = ok && hipHostMalloc((void**)&s.staging_h[f], bytes, hipHostMallocDefault) == hipSuccess;
      ok = ok && hipMalloc((void**)&s.staged_d[f], bytes) == hipSuccess;
      if (ok) (void)pg_bounds_region(s.staged_d[f], (int64_t)bytes);   // (debug build: the block's extent bounds staged-row numbers)
    }
    ok = ok && hipEventCreateWithFlags(&s.filled, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&s.tail_go, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&s.tail_done, hipEventDisableTiming) == hipSuccess;
    if (q->copy_log) ok = ok && hipEventCreate(&s.cp0) == hipSuccess && hipEventCreate(&s.cp1) == hipSuccess;
    if (!ok) break;
    *s.flag_h = 0;
    *s.count_h = 0;
    ok = ok && hipMemset(s.count_d, 0, 64) == hipSuccess;
    ok = ok && hipMemset(s.landed_d, 0, 8) == hipSuccess;
    ok = ok && hipEventRecord(s.filled, q->copy_stream) == hipSuccess;
  }
  if (!ok) {
    missq_free(q);
    return PG_ERR_NOMEM;
  }
  q->pool = new (std::nothrow) Pool(n_threads);
  if (!q->pool) {
    missq_free(q);
    return PG_ERR_NOMEM;
  }
  // two spare staging buffers per field (see the worker: a job whose slot's buffer is still held by a straggler of the
  // gather pool takes one instead of waiting for it)
  {
    const int spares = 2;
    for (int f = 0; f < n_fields; ++f)
      for (int i = 0; i < spares; ++i) {
        float* b = nullptr;
        if (hipHostMalloc((void**)&b, (size_t)max_rows * q->sstride[f] * sizeof(float), hipHostMallocDefault) != hipSuccess) {
          (void)hipGetLastError();      // no spare: the job waits for the straggler instead
          break;
        }
        q->parked[f].push_back({b, 0});
      }
  }
  hsa_copy_init(q);
  q->worker = std::thread(missq_worker, q);
  *out = q;
  return PG_OK;
}

int pg_missq_destroy(pg_missq_t* q) {
  missq_free(q);
  return PG_OK;
}

int pg_missq_slot_buffers(pg_missq_t* q, int slot, int32_t** miss_pos_dev, int64_t** miss_fullid_pinned,
                          int32_t** miss_count_dev) {
  if (!q || slot < 0 || slot >= q->n_slots) return PG_ERR_INVALID;
  pg_missq_slot& s = q->slots[slot];
  if (miss_pos_dev) *miss_pos_dev = s.pos_d;
  if (miss_fullid_pinned) *miss_fullid_pinned = s.fullid_h;
  if (miss_count_dev) *miss_count_dev = s.count_d;
  return PG_OK;
}

int pg_missq_slot_staged(pg_missq_t* q, int slot, int field, float** staged_dev) {
  if (!q || slot < 0 || slot >= q->n_slots || field < 0 || field >= q->n_fields || !staged_dev) return PG_ERR_INVALID;
  *staged_dev = q->slots[slot].staged_d[field];
  return PG_OK;
}

int pg_missq_staged_stride(pg_missq_t* q, int field, int32_t* stride_out) {
  if (!q || field < 0 || field >= q->n_fields || !stride_out) return PG_ERR_INVALID;
  *stride_out = q->sstride[field];
  return PG_OK;
}

int pg_missq_slot_dup_buffers(pg_missq_t* q, int slot, int32_t** dup_pos_dev, int32_t** dup_src_dev,
                              int32_t** dup_count_dev) {
  if (!q || slot < 0 || slot >= q->n_slots) return PG_ERR_INVALID;
  pg_missq_slot& s = q->slots[slot];
  if (dup_pos_dev) *dup_pos_dev = s.dup_pos_d;
  if (dup_src_dev) *dup_src_dev = s.dup_src_d;
  if (dup_count_dev) *dup_count_dev = s.dup_count_d;
  return PG_OK;
}

int pg_missq_submit(pg_missq_t* q, int slot, float* const* out_ptrs, const int32_t* out_strides,
                    const int32_t* pos_lo, const int32_t* slots_dev, pg_stream_t stream) {
  if (!q || slot < 0 || slot >= q->n_slots || !out_ptrs || !out_strides) return PG_ERR_INVALID;
  // the repeats are filled from the worker's staged rows: every primary row must go through the worker
  if (slots_dev && q->cpu_share.load(std::memory_order_relaxed) != 256) return PG_ERR_UNSUPPORTED;
  pg_missq_slot& s = q->slots[slot];
  uint32_t seq;
  {
    for (int i = 0; i < 20000; ++i) {        // (as pg_missq_wait_idle: poll before sleeping)
      if (__atomic_load_n(&s.done_pub, __ATOMIC_ACQUIRE) == s.submitted) break;
      __builtin_ia32_pause();
    }
    std::unique_lock<std::mutex> l(q->m);
    // the slot's previous submission must have left the worker (its buffers are about to be reused)
    q->cv_done.wait(l, [&] { return s.done == s.submitted || q->error != PG_OK; });
    if (q->error != PG_OK) return q->error;
    seq = ++s.submitted;
    s.t_submit = std::chrono::steady_clock::now();
    for (int f = 0; f < q->n_fields; ++f) {
      s.out[f] = out_ptrs[f];
      s.out_stride[f] = out_strides[f];
      s.pos_lo[f] = pos_lo ? pos_lo[f] : 0;
    }
    s.dedup = slots_dev != nullptr;
    s.share = q->cpu_share.load(std::memory_order_relaxed);
    // every wanted field is read in place from the staged block and the copies go straight to an SDMA engine:
    // the job never touches the copy stream (see k_wait_direct)
    static const bool no_direct = getenv("PG_MISSQ_NO_DIRECT") != nullptr && atoi(getenv("PG_MISSQ_NO_DIRECT")) != 2;
    bool any = false, all_staged = true;
    for (int f = 0; f < q->n_fields; ++f) {
      if (!s.out[f] && s.out_stride[f] != -1) continue;
      any = true;
      if (s.out[f]) all_staged = false;
    }
    // ... and a job WITH rows to scatter (GraphSAGE's layers 1-2, --fetch-all) stays off the copy stream as well: its
    // consumer scatters them on its own stream right after its wait (scatter_on_consumer). The wait-for-SDMA kernel
    // such a job used to park on the copy stream for the length of the copy (0.34 ms for 19 MB) held back whatever
    // shared its hardware queue — with the load stream's k_publish behind it, gather(k+1) could not overlap copy(k):
    // the all-layer leg read 0.52-0.58 instead of 0.36-0.38 ms/step whenever the runtime's queue assignment fell
    // that way. PG_MISSQ_NO_DIRECT=1: everything through the copy stream; =2: only staged-only jobs are direct.
    static const int no_direct_mode = getenv("PG_MISSQ_NO_DIRECT") ? atoi(getenv("PG_MISSQ_NO_DIRECT")) : 0;
    // (with cpu_share < 1 the tail of the list is written into the staged block by the device — pg_missq_device_tail —
    // so the consumer's scatter still covers the whole list)
    s.direct = any && (all_staged || no_direct_mode != 2) && q->hsa_ok && !no_direct && !q->wait_value;
  }
  if (slots_dev)
    hipLaunchKernelGGL(k_publish_dedup, dim3(1), dim3(256), 0, as_stream(stream), s.count_d, s.count_h, s.flag_h, seq,
                       const_cast<int32_t*>(slots_dev), s.dup_src_d, s.dup_pos_d, s.dup_count_d);
  else
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, as_stream(stream), s.count_d, s.count_h, s.flag_h, seq);
  PG_LAUNCH_CHECK();
  {
    std::lock_guard<std::mutex> l(q->m);
    q->jobs.emplace_back(slot, seq);
  }
  q->cv_job.notify_one();
  return PG_OK;
}

int pg_missq_wait(pg_missq_t* q, int slot, pg_stream_t stream, int32_t* miss_count_out) {
  if (!q || slot < 0 || slot >= q->n_slots) return PG_ERR_INVALID;
  pg_missq_slot& s = q->slots[slot];
  {
    std::unique_lock<std::mutex> l(q->m);
    static const bool dbg3 = getenv("PG_MISSQ_DEBUG") && atoi(getenv("PG_MISSQ_DEBUG")) >= 3;
    if (dbg3) fprintf(stderr, "[missq-wait] slot %d submitted %u done %u | queue %zu\n", slot, s.submitted, s.done, q->jobs.size());
    q->cv_done.wait(l, [&] { return s.done == s.submitted || q->error != PG_OK; });
    if (q->error != PG_OK) return q->error;
    if (miss_count_out) *miss_count_out = s.last_count;
  }
  if (s.tail_pending) PG_HIP(hipStreamWaitEvent(as_stream(stream), s.tail_done, 0));
  if (s.direct) {   // the copies are in no stream: wait for the engine here (they were issued before `done` moved)
    for (int f = 0; f < q->n_fields; ++f)
      if (s.sig[f].handle &&
          hsa_signal_wait_scacquire(s.sig[f], HSA_SIGNAL_CONDITION_LT, 1, 3000000000ull, HSA_WAIT_STATE_ACTIVE) > 0)
        return PG_ERR_HIP;
    return scatter_on_consumer(q, s, stream);
  }
  PG_HIP(hipStreamWaitEvent(as_stream(stream), s.filled, 0));
  return PG_OK;
}

/* host-side: returns once the worker has left the slot's latest submission (its copy, scatter and signal are
 * enqueued on the copy stream); no stream is touched. See the note on barrier packets in pg_missq_create. */
int pg_missq_wait_idle(pg_missq_t* q, int slot) {
  if (!q || slot < 0 || slot >= q->n_slots) return PG_ERR_INVALID;
  pg_missq_slot& s = q->slots[slot];
  // the caller is the launch thread: spin briefly before sleeping on the condition variable (a futex wake-up costs
  // ~50 us on an idle host and milliseconds on a loaded one; the worker is normally within a few microseconds of done)
  for (int i = 0; i < 20000; ++i) {
    if (__atomic_load_n(&s.done_pub, __ATOMIC_ACQUIRE) == s.submitted) break;
    __builtin_ia32_pause();
  }
  std::unique_lock<std::mutex> l(q->m);
  q->cv_done.wait(l, [&] { return s.done == s.submitted || q->error != PG_OK; });
  return q->error;
}

/* device-side variant: enqueues a one-wave kernel on `stream` that sleeps until the slot's rows (of its
 * latest submission) are in place. Never blocks the host. */
int pg_missq_wait_device(pg_missq_t* q, int slot, pg_stream_t stream) {
  if (!q || slot < 0 || slot >= q->n_slots) return PG_ERR_INVALID;
  pg_missq_slot& s = q->slots[slot];
  uint32_t seq;
  bool enqueued, direct;
  {
    std::lock_guard<std::mutex> l(q->m);
    if (q->error != PG_OK) return q->error;
    seq = s.submitted;
    enqueued = s.done == s.submitted;   // the worker has already put this submission's copy on its stream
    direct = s.direct;
    if (enqueued) ++q->n_wait_event; else ++q->n_wait_spin;
  }
  if (seq == 0) return PG_OK;
  if (s.tail_pending) PG_HIP(hipStreamWaitEvent(as_stream(stream), s.tail_done, 0));   // the device's share of the list
  if (direct) {
    DirectSignals sg{};
    bool pending = !enqueued;
    for (int f = 0; f < q->n_fields; ++f)
      if (s.sig[f].handle) {
        sg.v[f] = &reinterpret_cast<amd_signal_t*>(s.sig[f].handle)->value;
        if (enqueued && hsa_signal_load_scacquire(s.sig[f]) > 0) pending = true;
      }
    if (pending) {                     // (issued and already landed: nothing to wait for)
      constexpr int poll_sleeps = 3;
      hipLaunchKernelGGL(k_wait_direct, dim3(1), dim3(1), 0, as_stream(stream), s.issued_h, seq, sg, q->timeout_d,
                         poll_sleeps);
      PG_LAUNCH_CHECK();
    }
    return scatter_on_consumer(q, s, stream);
  }
  if (enqueued) {
    // the usual case with two batches of look-ahead: an ordinary event dependency, no kernel parked on the
    // consumer's hardware queue (a spinning kernel stalls whatever else the runtime multiplexes onto that queue)
    PG_HIP(hipStreamWaitEvent(as_stream(stream), s.filled, 0));
    return PG_OK;
  }
  if (q->wait_value) {
    // experiment (PG_MISSQ_WAITVALUE=1): the command processor waits on the value, no kernel spins
    PG_HIP(hipStreamWaitValue32(as_stream(stream), s.landed_d, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
    return PG_OK;
  }
  hipLaunchKernelGGL(k_wait_landed, dim3(1), dim3(1), 0, as_stream(stream), s.landed_d, seq, q->timeout_d);
  PG_LAUNCH_CHECK();
  return PG_OK;
}

/* cpu_share < 1: the rows of the slot's latest submission that the worker leaves alone — [count * share / 256, count) of
 * its miss list — are read from the pinned host table by the device itself, on `stream` (the fetching stream, right after
 * the submit). They land in the slot's staged block in miss-list order, next to the rows the worker copies; fields with
 * an output frame whose job is NOT scattered by its consumer (no direct SDMA path) are also scattered from the host
 * table to the frame here. A no-op at share 1.                                                                  */
int pg_missq_device_tail(pg_missq_t* q, int slot, pg_stream_t stream) {
  if (!q || slot < 0 || slot >= q->n_slots) return PG_ERR_INVALID;
  pg_missq_slot& s = q->slots[slot];
  const int share = s.share;             // of the slot's latest submission (this thread made it)
  s.tail_pending = false;
  if (share >= 256) return PG_OK;
  // The reads run on the queue's own copy stream (idle otherwise: direct jobs never touch it), ordered after the
  // caller's split by an event, and the consumer waits for `tail_done`: on the fetching stream they serialised with
  // the block transposes and the label lookup of the same minibatch (2 gather threads, share 0.31: 0.212 ms/step with
  // the tail on the load stream).
  {
    PG_HIP(hipEventRecord(s.tail_go, as_stream(stream)));
    PG_HIP(hipStreamWaitEvent(q->copy_stream, s.tail_go, 0));
    stream = (pg_stream_t)q->copy_stream;
  }
  for (int f = 0; f < q->n_fields; ++f) {
    if (!s.out[f] && s.out_stride[f] != -1) continue;
    const pg_missq_field_t& fd = q->fields[f];
    int rc = PG_OK;
    if (s.direct || !s.out[f])
      rc = pg::scatter_host_tail(fd.table, fd.table_stride, nullptr, 0, s.fullid_h, q->max_rows, s.count_d, share, fd.dim,
                                 s.staged_d[f], q->sstride[f], stream);
    else
      rc = pg::scatter_host_tail(fd.table, fd.table_stride, s.pos_d, s.pos_lo[f], s.fullid_h, q->max_rows, s.count_d, share,
                                 fd.dim, s.out[f], s.out_stride[f], stream);
    if (rc != PG_OK) return rc;
  }
  {
    PG_HIP(hipEventRecord(s.tail_done, q->copy_stream));
    s.tail_pending = true;               // the slot's consumer orders itself after tail_done (pg_missq_wait*)
  }
  return PG_OK;
}

/* before the split of a NEW submission rewrites the slot's miss list: order `stream` after the device tail of the slot's
 * previous submission (it reads that list on the copy stream). A no-op when there was none.                     */
int pg_missq_order_after_tail(pg_missq_t* q, int slot, pg_stream_t stream) {
  if (!q || slot < 0 || slot >= q->n_slots) return PG_ERR_INVALID;
  pg_missq_slot& s = q->slots[slot];
  if (s.tail_pending) PG_HIP(hipStreamWaitEvent(as_stream(stream), s.tail_done, 0));
  return PG_OK;
}

int pg_missq_set_cpu_share(pg_missq_t* q, int32_t share_of_256) {
  if (!q || share_of_256 < 0 || share_of_256 > 256) return PG_ERR_INVALID;
  q->cpu_share.store(share_of_256, std::memory_order_relaxed);
  return PG_OK;
}

/* 1 if a device-side wait ever gave up (worker failure); synchronises the device */
int pg_missq_timed_out(pg_missq_t* q, int* out) {
  if (!q || !out) return PG_ERR_INVALID;
  uint32_t v[3] = {0, 0, 0};
  PG_HIP(hipMemcpy(v, q->timeout_d, 12, hipMemcpyDeviceToHost));
  *out = (int)v[0];
  if (v[0] && getenv("PG_MISSQ_DEBUG"))
    fprintf(stderr, "[pg_missq] device-side wait timed out: %s (consumer gave up on seq %u)\n",
            v[1] ? "copy engine's completion signal" : "rows never landed", v[2]);
  return PG_OK;
}

int pg_missq_drain(pg_missq_t* q) {
  if (!q) return PG_ERR_INVALID;
  std::unique_lock<std::mutex> l(q->m);
  q->cv_done.wait(l, [&] {
    if (q->error != PG_OK) return true;
    for (auto& s : q->slots)
      if (s.done != s.submitted) return false;
    return true;
  });
  // ... and the copies the worker handed straight to an SDMA engine have landed: they sit in no HIP stream, so a device-wide
  // synchronise does not wait for them — a timed region that ends with drain + synchronise would otherwise leave the copies
  // of its last prepared batches outside (bounded: 1 s per signal; nothing is outstanding on a queue that never went direct)
  if (q->error == PG_OK)
    for (auto& s : q->slots)
      for (int f = 0; f < PG_MAX_FIELDS; ++f)
        if (s.sig[f].handle &&
            hsa_signal_wait_scacquire(s.sig[f], HSA_SIGNAL_CONDITION_LT, 1, 1000000000ull, HSA_WAIT_STATE_ACTIVE) > 0) {
          static bool said = false;
          if (!said) fprintf(stderr, "[missq] pg_missq_drain: a direct copy's completion signal did not arrive within 1 s\n");
          said = true;
        }
  return q->error;
}

int pg_missq_copy_log(pg_missq_t* q, int64_t* bytes, float* ms, int64_t cap, int64_t* n_out) {
  if (!q || !n_out || cap < 0 || (cap > 0 && (!bytes || !ms))) return PG_ERR_INVALID;
  std::lock_guard<std::mutex> l(q->m);
  const int64_t n = std::min<int64_t>(cap, (int64_t)q->copies.size());
  const size_t first = q->copies.size() - (size_t)n;
  for (int64_t i = 0; i < n; ++i) {
    bytes[i] = q->copies[first + i].first;
    ms[i] = q->copies[first + i].second;
  }
  *n_out = n;
  return PG_OK;
}

// ONE statistics call (round 6: pg_missq_stats, _stats_max, _spared_jobs, _rescued_chunks and _copy_engine merged)
int pg_missq_stats(pg_missq_t* q, pg_missq_stats_t* out, int reset_max) {
  if (!q || !out) return PG_ERR_INVALID;
  std::lock_guard<std::mutex> l(q->m);
  const double n = q->n_jobs ? (double)q->n_jobs : 1.0;
  out->jobs = (int64_t)q->n_jobs; out->rows = (int64_t)q->n_rows;
  out->waits_by_event = (int64_t)q->n_wait_event; out->waits_by_spin_kernel = (int64_t)q->n_wait_spin;
  out->spared_jobs = q->n_spared;
  out->rescued_chunks = q->pool ? q->pool->rescued() : 0;
  out->us_submit_to_published = q->t_sub2flag / n; out->us_cpu_gather = q->t_gather / n;
  out->us_enqueue = q->t_enqueue / n; out->us_submit_to_done = q->t_total / n;
  out->max_us_wait_published = q->mx_flag; out->max_us_cpu_gather = q->mx_gather;
  out->max_us_enqueue = q->mx_enqueue; out->max_us_submit_to_done = q->mx_total;
  if (reset_max) q->mx_flag = q->mx_gather = q->mx_enqueue = q->mx_total = 0;
  out->sdma_engine_mask = q->hsa_ok ? q->engine : 0u;
  out->_pad = 0;
  for (int b = 0; b < 16; ++b) out->engine_GBps[b] = q->engine_GBps[b];
  return PG_OK;
}

}  // extern "C"
