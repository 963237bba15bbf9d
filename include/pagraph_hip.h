/* pagraph_hip.h — C-ABI of libpagraph_hip.so (MI355X / gfx950).
 *
 * The reference (zhiqi-0/PaGraph) has NO native code and NO FFI: its "plugin
 * boundary" for the minibatch hot path is a handful of Python call sites that
 * hand torch / DGL tensors to torch-CUDA index ops and to DGL 0.4.1's C++
 * sampler and SpMM kernels.  Each entry point below replaces one of those call
 * sites; the citation names the reference file:line it stands in for.
 *
 * Conventions
 *   - plain pointers + sizes only; no torch / DGL types.
 *   - every device pointer is HBM on the current HIP device unless it says
 *     "host" or "pinned"; pinned pointers must be hipHostMalloc'ed (or
 *     torch pin_memory) so the device can address them.
 *   - all kernels are enqueued on `stream` (a hipStream_t passed as void*) and
 *     are asynchronous w.r.t. the host.  Nothing here synchronises unless the
 *     comment says so.
 *   - return value: 0 = PG_OK, negative = error (pg_strerror()).  Nothing
 *     throws.  The caller owns every buffer; the library owns only the opaque
 *     handles it creates.
 *   - ids are int64 like the reference's LongTensors.  Adjacency `indices`
 *     are int32 (a per-GPU partition has < 2^31 vertices); `indptr` is int64
 *     (a partition may have > 2^31 edges).
 */
#ifndef PAGRAPH_HIP_H
#define PAGRAPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_OK 0
#define PG_ERR_INVALID (-1)  /* bad argument (null pointer, negative size, too many fields) */
#define PG_ERR_HIP (-2)      /* a HIP runtime call failed; pg_last_hip_error() has the code */
#define PG_ERR_NOMEM (-3)
#define PG_ERR_UNSUPPORTED (-4)
#define PG_ERR_OVERFLOW (-5) /* a caller-provided capacity was too small */

#define PG_MAX_FIELDS 4
#define PG_MAX_LAYERS 8

typedef void* pg_stream_t; /* hipStream_t */

int pg_version(void);
const char* pg_strerror(int code);
int pg_last_hip_error(void);

/* Debug build that bounds-checks ids (`make -C pagraph_amd/csrc bounds` -> libpagraph_hip_bounds.so, loaded by
 * pagraph_amd/_lib.py under PG_BOUNDS=1). The reference has one assert (storage.py:149) and no bounds checks on ids: an id
 * out of range is an illegal address somewhere behind the kernel that followed it. In the debug build every index a kernel
 * reads from a buffer and then follows (vertex ids, block edges, cache slots, staged-row numbers) is checked; the first
 * offender of the process is recorded and the access redirected to element 0, so the launch completes.
 * pg_bounds_enabled: 1 in the debug build, 0 in the product build.
 * pg_bounds_region:  (debug build; no-op otherwise) `bytes` bytes at device / pinned address `base` are one buffer; kernels'
 *                    bounds that the ABI does not carry (rows of a slot array, of the cache, of a staged block) are derived
 *                    from the registered buffer a pointer argument lies in. bytes <= 0 forgets the buffer.
 * pg_bounds_report:  (debug build; PG_ERR_UNSUPPORTED otherwise) synchronises the device; rec[8] = {hit, kernel (PG_K_* of
 *                    csrc/pg_common.h), site, value, bound, block, total offenders, 0}; unit = the .hip file of the kernel;
 *                    reset != 0 clears the record.                                                                        */
int pg_bounds_enabled(void);
int pg_bounds_region(const void* base, int64_t bytes);
int pg_bounds_report(uint64_t* rec, char* unit, int32_t unit_len, int32_t reset);
/* A captured step as plain launches (round 5; csrc/pg_tape.hip): pg_tape_from_graph walks a hipGraph_t that is a linear chain of
 * kernel / memset nodes (a single-stream capture of this library's launches) and keeps the nodes' launch parameters; the graph —
 * owner of the parameter storage and of the memory its pointers refer to — must outlive the tape. pg_tape_launch issues the same
 * kernels with the same arguments in the same order on `stream`, without hipGraphLaunch's ~12 us between two replays.
 * PG_ERR_UNSUPPORTED for any other graph shape (the caller keeps replaying the graph). */
typedef struct pg_tape pg_tape_t;
int pg_tape_from_graph(void* hip_graph, pg_tape_t** out, int32_t* n_kernels, int32_t* n_other);
int pg_tape_launch(const pg_tape_t* tape, pg_stream_t stream);
int pg_tape_destroy(pg_tape_t* tape);
/* number of CUs of the current device, or <0 when no HIP device is visible */
int pg_device_cu_count(void);

/* ------------------------------------------------------------------------
 * 1. Feature cache  —  PaGraph/storage/storage.py  (GraphCacheServer)
 * ------------------------------------------------------------------------
 * HBM layout: per field a row-major fp32 array `cache[slot, :]` (storage.py:151
 * `gpu_fix_cache[name]`), row stride `cache_stride` floats — the Python layer packs all fields of
 * a vertex into one 128-byte-aligned row and passes column views (cache pointer + stride).
 * The reference's two per-vertex arrays gpu_flag (bool, storage.py:38) and
 * localid2cacheid (int64, storage.py:50) are fused into ONE int32
 * `slot_map[local_id]`: >=0 cache slot, -1 not cached.
 */
typedef struct pg_field {
  const float* cache;   /* device [cached_rows, cache_stride]; may be NULL when nothing is cached */
  float* out;           /* device [n, out_stride] — the frame handed to the model (storage.py:186) */
  int32_t dim;          /* floats per row (storage.py:65 dims[name]) */
  int32_t cache_stride; /* floats */
  int32_t out_stride;   /* floats */
  int32_t _pad;
} pg_field_t;

/* storage.py:38,50 — slot_map[0:node_num] = -1 */
int pg_slot_map_reset(int32_t* slot_map, int64_t node_num, pg_stream_t stream);
/* storage.py:145,153 — slot_map[nids[j]] = j  (localid2cacheid[nids]=arange; gpu_flag[nids]=True) */
int pg_slot_map_assign(int32_t* slot_map, const int64_t* nids, int64_t rows, pg_stream_t stream);
/* views for API parity: gpu_flag (uint8 0/1) and localid2cacheid (int64, 0 where uncached) */
int pg_slot_map_export(const int32_t* slot_map, int64_t node_num, uint8_t* gpu_flag,
                       int64_t* localid2cacheid, pg_stream_t stream);

/* storage.py:176-204 (fetch_data inner loop, all layers in ONE launch).
 * For every r in [0,n): id = ids[r] (id < 0 = padding, row skipped); s = slot_map[id];
 *   s >= 0 : out_f[r,:] = cache_f[s,:]  for every field f            (storage.py:191-193)
 *   s <  0 : row r is appended to the miss list:
 *            miss->pos[j] = r, miss->fullid[j] = nid_map[id]         (storage.py:117,182)
 * `miss->count` (device or pinned int32, zeroed by this call) receives the
 * number of misses; miss list order is unspecified (a permutation of the
 * reference's mask order) — results do not depend on it.
 * slot_scratch: optional device int32[n]; when given, the split pass stores every row's slot
 * there (coalesced) and the copy pass reads it back instead of repeating the random lookup.
 * stats: optional device uint64[2], ACCUMULATED (not reset): [0] += rows looked up (ids >= 0),
 * [1] += misses — the reference's try_num / miss_num (storage.py:219-221) without a host sync.
 * timer: optional (pg_timer_create); its HIP events are attached to the dispatch of the copy kernel itself
 * (hipExtLaunchKernelGGL start / stop events; not the split pass), so pg_timer_elapsed_ms reads that kernel's
 * own begin-to-end time on `stream`, like rocprofv3's k_gather row.                                    */
typedef struct pg_timer pg_timer_t;
/* the miss list of one launch: miss->pos / fullid need capacity n and may be pinned-host pointers; miss->count (device or
 * pinned int32) is zeroed by the call */
typedef struct pg_miss_list {
  int32_t* pos;
  int64_t* fullid;
  int32_t* count;
} pg_miss_list_t;
/* Miss-list index dedup (north star: "index dedup" in the gather; `dedup` may be NULL). The reference looks every layer's ids
 * up on their own (storage.py:173-200), so a vertex that sits in two layers of a NodeFlow is fetched from the host twice
 * when it misses. With a pg_dedup_t the split pass keeps the FIRST occurrence of a missed id in the miss list and
 * diverts a later one to the dup list: rows [lo[r], lo[r+1]) are layer r of the launch (lo[0] = 0, lo[n_ranges] = n);
 * bit r of sorted_mask says layer r's ids ascend (sampler spec rule 5: every non-seed layer; the padding ids < 0 of a
 * fixed-shape layer sit at its end). A missed row of layer r searches layers 0..r-1 (binary search over `ids`
 * itself). A repeat: slots_out / slot_scratch[row] stays -1, it is NOT in the miss list, (row, earlier row) is
 * appended to dup_pos / dup_src (device int32[n]) and *dup_count (device; reset by the call). stats count it as a
 * miss, like the reference. The consumer resolves dup_src to staged rows once the split has run
 * (pg_missq_submit does: dup_src[k] = -slots[dup_src[k]] - 3) and fills the rows with pg_scatter_rows_dups
 * after the primary rows have landed. Same frames, bit for bit; fewer rows over PCIe.                        */
typedef struct pg_dedup {
  int32_t n_ranges;
  int32_t lo[PG_MAX_LAYERS + 1];
  uint32_t sorted_mask;
  int32_t* dup_pos;
  int32_t* dup_src;
  int32_t* dup_count;
} pg_dedup_t;
int pg_gather_rows(const int64_t* ids, int64_t n, const int32_t* slot_map, const int64_t* nid_map,
                   const pg_field_t* fields, int n_fields, const pg_miss_list_t* miss, int32_t* slot_scratch,
                   uint64_t* stats, pg_timer_t* timer, const pg_dedup_t* dedup, pg_stream_t stream);

/* The two halves of pg_gather_rows on their own, for callers that do not materialise every row:
 * pg_split_rows = the hit/miss split only (storage.py:176-182): slots_out[r] (device int32[n], required) receives
 *   slot_map[id] for a hit, -(j + 3) for the row that became entry j of the miss list, -2 for padding (id < 0);
 *   miss list / stats / dedup exactly as pg_gather_rows.
 * pg_gather_rows_presplit = the copy only, for n rows whose slots are given (slots[r] >= 0: out_f[r,:] =
 *   cache_f[slots[r],:]; anything else leaves the row untouched). `slots` / `fields[].out` may point into the middle
 *   of a NodeFlow's arrays: rows of layers whose features are consumed in place (pg_spmm_fwd_rows) are skipped. */
int pg_split_rows(const int64_t* ids, int64_t n, const int32_t* slot_map, const int64_t* nid_map,
                  const pg_miss_list_t* miss, int32_t* slots_out, uint64_t* stats, const pg_dedup_t* dedup,
                  pg_stream_t stream);
/* slots_out[i] = slot_map[ids[i]] (ids[i] < 0, the padding of a fixed-shape layer: -2) for a launch over a FULLY cached table
 * (storage.py:207-216): what pg_split_rows writes when nothing can miss, without the miss list and its counter's zero fill.
 * stats (may be NULL): stats[0] += rows looked up.                                                                    */
int pg_slots_full(const int64_t* ids, int64_t n, const int32_t* slot_map, int32_t* slots_out, uint64_t* stats,
                  pg_stream_t stream);
/* the general form behind pg_scatter_rows / _range / _dups: out[pos[j] - pos_lo, :dim] = staged[(src_row ? src_row[j] : j)
 * * staged_stride, :dim] for j < (n_dev ? *n_dev : n); rows below pos_lo or with a negative source are skipped.
 * staged_stride >= dim lets the staged block keep padded rows (the miss queue pads wide rows to whole 16-byte pieces so
 * that pg_spmm_fwd_rows can read a ragged width — 602 — in place). max_blocks > 0 caps the grid.                      */
int pg_scatter_rows_strided(const float* staged, int32_t staged_stride, const int32_t* pos, const int32_t* src_row,
                            int64_t n, const int32_t* n_dev, int32_t dim, float* out, int32_t out_stride, int32_t pos_lo,
                            int32_t max_blocks, pg_stream_t stream);
/* out[dup_pos[k] - pos_lo, :] = staged[dup_staged_row[k], :] for k < *dup_count_dev (entries below pos_lo or with a
 * negative staged row are skipped); cap = launch upper bound                                               */
int pg_scatter_rows_dups(const float* staged, const int32_t* dup_pos, const int32_t* dup_staged_row, int64_t cap,
                         const int32_t* dup_count_dev, int32_t dim, float* out, int32_t out_stride, int32_t pos_lo,
                         pg_stream_t stream);
int pg_gather_rows_presplit(const int32_t* slots, int64_t n, const pg_field_t* fields, int n_fields,
                            pg_timer_t* timer, pg_stream_t stream);

/* storage.py:207-216 (fetch_from_cache): full cache, slot == local id. out_f[r,:] = cache_f[ids[r],:] */
int pg_gather_rows_full(const int64_t* ids, int64_t n, const pg_field_t* fields, int n_fields,
                        pg_stream_t stream);

/* batch_labels = labels[batch_nids] (examples/profile/pa_gcn.py:99-100): out[i] = labels[ids[i]], or
 * `fill` where ids[i] is outside [0, n_labels) (padding ids of a fixed-shape NodeFlow are -1).
 * n_valid_out (device int32, may be NULL): number of i with out[i] != fill and >= 0 — the rows a loss with
 * ignore_index = fill will count (pg_head reads it).                                             */
int pg_gather_labels(const int64_t* ids, int64_t n, const int64_t* labels, int64_t n_labels, int64_t fill,
                     int64_t* out, int32_t* n_valid_out, pg_stream_t stream);
/* The same with a self-cleaning count: scratch2 = two device int32 words, zero before the FIRST call and owned by this call
 * sequence afterwards (one stream at a time); the last block to finish writes *n_valid_out and zeroes them again, so no
 * zero fill precedes the launch. n > 0. */
int pg_gather_labels_sc(const int64_t* ids, int64_t n, const int64_t* labels, int64_t n_labels, int64_t fill,
                        int64_t* out, int32_t* n_valid_out, int32_t* scratch2, pg_stream_t stream);

/* storage.py:199-200 — out[pos[j], :] = staged[j, :] for j < n (n from host, or *n_dev when n_dev != NULL
 * in which case `n` is the launch upper bound). `staged` is device memory, row stride = dim.        */
int pg_scatter_rows(const float* staged, const int32_t* pos, int64_t n, const int32_t* n_dev,
                    int32_t dim, float* out, int32_t out_stride, pg_stream_t stream);

/* the same for miss rows at positions >= pos_lo only: out[(pos[j] - pos_lo), :] = staged[j, :]; rows below pos_lo
 * stay in the staged block for a consumer that reads them there (pg_spmm_fwd_rows).                      */
int pg_scatter_rows_range(const float* staged, const int32_t* pos, int64_t n, const int32_t* n_dev,
                          int32_t dim, float* out, int32_t out_stride, int32_t pos_lo, pg_stream_t stream);

/* storage.py:128 — the server-side `table[nids]` on host memory, multi-threaded:
 * staged[j, 0:dim] = table[fullids[j], 0:dim]. Pure host function (blocks until done).             */
int pg_host_gather_rows(const float* table, int64_t table_stride, int32_t dim, const int64_t* fullids,
                        int64_t n, float* staged, int n_threads);

/* Zero-copy variant of the miss path: the device reads `table` (pinned host memory,
 * device-addressable) directly over PCIe: out[pos[j],:] = table[fullid[j],:] for j < *n_dev.        */
int pg_scatter_rows_from_host(const float* table_pinned, int64_t table_stride, const int32_t* pos,
                              const int64_t* fullid, int64_t n_max, const int32_t* n_dev, int32_t dim,
                              float* out, int32_t out_stride, pg_stream_t stream);
/* the same for rows j >= count * start_num / 256 only (count = *n_dev): the device-read tail of a miss list
 * whose head the CPU miss queue moves (pg_missq_set_cpu_share).                                         */
int pg_scatter_rows_from_host_tail(const float* table_pinned, int64_t table_stride, const int32_t* pos,
                                   const int64_t* fullid, int64_t n_max, const int32_t* n_dev,
                                   int32_t start_num, int32_t dim, float* out, int32_t out_stride,
                                   pg_stream_t stream);

/* Asynchronous miss path (storage.py:117-131,196-200 off the trainer's critical path): a worker
 * thread owned by the handle waits for the GPU to publish a slot's miss list, gathers table[fullid]
 * into pinned staging with `n_threads` helpers, and enqueues hipMemcpyAsync + the row scatter on its
 * own copy stream. One slot per in-flight minibatch.                                                */
typedef struct pg_missq pg_missq_t;
typedef struct pg_missq_field {
  const float* table;    /* host table [V, table_stride] (pageable or pinned) */
  int64_t table_stride;  /* floats */
  int32_t dim;
  int32_t _pad;
} pg_missq_field_t;
int pg_missq_create(int device, int n_slots, int64_t max_rows, const pg_missq_field_t* fields, int n_fields,
                    int n_threads, pg_missq_t** out);
int pg_missq_destroy(pg_missq_t* q);
/* the slot's miss-list buffers, to be passed to pg_gather_rows as its pg_miss_list_t */
int pg_missq_slot_buffers(pg_missq_t* q, int slot, int32_t** miss_pos_dev, int64_t** miss_fullid_pinned,
                          int32_t** miss_count_dev);
/* call right after pg_gather_rows / pg_split_rows (...slot buffers...) on the same stream: publishes the miss list to the
 * worker. out_ptrs[f] / out_strides[f]: destination of field f's rows (NULL with stride 0: field not wanted).
 * pos_lo (may be NULL = 0): a per-field first scattered position — miss rows at positions >= pos_lo[f] go to
 * out_ptrs[f][(pos - pos_lo[f]) * stride]; rows below it are only copied to the slot's device staging block
 * (pg_missq_slot_staged), where pg_spmm_fwd_rows reads them. out_ptrs[f] == NULL with out_strides[f] == -1: nothing of
 * field f is scattered (copy only).
 * slots_dev (may be NULL): for a launch that was split with a pg_dedup_t over the slot's dup buffers
 * (pg_missq_slot_dup_buffers) = the slot array that split wrote (still intact at this point of `stream`). The publish step
 * turns every dup entry's "earlier row" into that row's staged index; after the primary rows' scatter the worker fills the
 * repeats on the device (pg_scatter_rows_dups) — they never cross PCIe. The publish step also rewrites the repeats' OWN
 * entries of `slots_dev` to their primary's value, -(staged row + 3), so that a consumer reading rows in place through that
 * array (pg_spmm_fwd_rows, pg_linear_fwd with X1rows) finds them.
 * (Round 6: one entry point; pg_missq_submit_range / _submit_dedup were the same call with fewer arguments.) */
int pg_missq_submit(pg_missq_t* q, int slot, float* const* out_ptrs, const int32_t* out_strides,
                    const int32_t* pos_lo, const int32_t* slots_dev, pg_stream_t stream);
int pg_missq_slot_dup_buffers(pg_missq_t* q, int slot, int32_t** dup_pos_dev, int32_t** dup_src_dev,
                              int32_t** dup_count_dev);
/* device staging block of (slot, field): [max_rows, pg_missq_staged_stride] floats, row j = entry j of the slot's miss list once
 * the slot's wait (pg_missq_wait / _wait_device) has passed                                              */
int pg_missq_slot_staged(pg_missq_t* q, int slot, int field, float** staged_dev);
/* floats per row of field `field`'s staged block: dim rounded up to 4 for wide fields (dim >= 16), else dim */
int pg_missq_staged_stride(pg_missq_t* q, int field, int32_t* stride_out);
/* makes `stream` wait until the slot's miss rows have landed; blocks the HOST only until the worker has
 * enqueued the copy. miss_count_out (optional) receives the number of rows.                         */
int pg_missq_wait(pg_missq_t* q, int slot, pg_stream_t stream, int32_t* miss_count_out);
/* same ordering guarantee without ever blocking the host: a one-wave kernel on `stream` sleeps on a device
 * flag the worker's copy stream raises after the scatter (gives up after 3 s -> pg_missq_timed_out).   */
int pg_missq_wait_device(pg_missq_t* q, int slot, pg_stream_t stream);
/* blocks the HOST until the worker has left the slot's latest submission (its copy, scatter and signal sit in
 * the copy stream's queue); touches no stream. The launch thread calls it BEFORE it enqueues — on any stream — a
 * wait for an event recorded after that submission's consumer (frames consumed, ring slot free): a barrier
 * packet that shares a hardware queue with the copy stream would otherwise hold back the very copy the consumer's
 * pg_missq_wait_device kernel is spinning for (3 s stalls once a process owns more streams than hardware queues). */
int pg_missq_wait_idle(pg_missq_t* q, int slot);
/* Split every miss list between the two paths: rows [0, count * share / 256) go through the worker (CPU gather +
 * copy engine), the caller reads the rest over PCIe with pg_scatter_rows_from_host_tail(start_num = share) on
 * its own stream. Default 256 = everything through the worker.                                            */
int pg_missq_set_cpu_share(pg_missq_t* q, int32_t share_of_256);
int pg_missq_timed_out(pg_missq_t* q, int* out);
/* blocks the HOST until every submitted job has left the worker (its copy is enqueued); makes no HIP call, so
 * it is safe to call right before a device-wide synchronise (a worker that still has to enqueue a copy while
 * the trainer thread sits in hipDeviceSynchronize with a spin-wait kernel parked behind it was measured to
 * cost 15 ms). Round 3: it also waits (ROCr signals, no HIP call) until the copies handed straight to an SDMA
 * engine have landed — they sit in no HIP stream, so the synchronise that follows would not wait for them.   */
int pg_missq_drain(pg_missq_t* q);
/* Everything the queue counts, in one call (round 6: five getters merged). Counters since creation; the us_* fields are
 * mean per-job microseconds of the worker's phases (submit -> miss list published, CPU row gather, copy enqueue, submit ->
 * job done); max_us_*: the LONGEST single occurrence of each since the last call with reset_max != 0 — a stall of the miss
 * path shows in exactly one of them. rescued_chunks: 32-row chunks of the CPU gather the worker re-executed because the pool
 * thread that had claimed them was overdue (lost its CPU with the chunk in hand); spared_jobs: jobs that gathered into a
 * spare staging buffer because such a thread still held their slot's buffer — each a multi-millisecond stall that did not
 * happen. sdma_engine_mask: the SDMA engine the worker's host->device copies go to (hsa_amd_sdma_engine_id_t bit; 0 = the HIP
 * runtime's own choice through hipMemcpyAsync — the fallback path); engine_GBps: what every engine reached in the
 * calibration at creation (0 = engine not offered).                                                          */
typedef struct pg_missq_stats {
  int64_t jobs, rows, waits_by_event, waits_by_spin_kernel, spared_jobs, rescued_chunks;
  double us_submit_to_published, us_cpu_gather, us_enqueue, us_submit_to_done;
  double max_us_wait_published, max_us_cpu_gather, max_us_enqueue, max_us_submit_to_done;
  uint32_t sdma_engine_mask, _pad;
  double engine_GBps[16];
} pg_missq_stats_t;
int pg_missq_stats(pg_missq_t* q, pg_missq_stats_t* out, int reset_max);
/* cpu_share < 1 (pg_missq_set_cpu_share): the tail of the slot's latest miss list, [count * share / 256, count), read
 * from the pinned host table by the device on `stream` — call on the fetching stream right after pg_missq_submit.
 * The rows land in the slot's staged block in miss-list order (where an in-place consumer, or the consumer-side
 * scatter of a direct job, finds them); a field whose rows the WORKER scatters (no direct SDMA path) is scattered to
 * its frame here instead. No-op at share 1. The host table must be page-locked / registered.                       */
int pg_missq_device_tail(pg_missq_t* q, int slot, pg_stream_t stream);
/* call on the fetching stream BEFORE the split of the slot's next submission when cpu_share < 1: the previous
 * submission's device tail (which runs on the queue's own stream) still reads the slot's miss list              */
int pg_missq_order_after_tail(pg_missq_t* q, int slot, pg_stream_t stream);
/* diagnosis (env PG_MISSQ_COPYLOG=1 at creation): the last `cap` host->device copies of the worker's wide field as
 * (bytes, milliseconds on the copy stream), oldest first; *n_out = entries written                        */
int pg_missq_copy_log(pg_missq_t* q, int64_t* bytes, float* ms, int64_t cap, int64_t* n_out);

/* ------------------------------------------------------------------------
 * 2. Neighbour sampler  —  dgl.contrib.sampling.NeighborSampler as called at
 *    examples/profile/pa_gcn.py:71-76 and PaGraph/partition/utils.py:11-18.
 *    DGL 0.4.1 is not part of the reference checkout; semantics follow the
 *    build-defined spec in DESIGN.md §"Sampler spec" (parity unpinned).
 * ------------------------------------------------------------------------ */
typedef struct pg_sampler pg_sampler_t;
#define PG_HEAVY_ROW 32   /* a source with more edges than this in one block is a hub */

typedef struct pg_nodeflow_desc {
  /* outputs, all device memory owned by the caller */
  int64_t* node_mapping;   /* [cap_nodes] local ids, layer 0 first (NodeFlow._node_mapping)       */
  int32_t* layer_offsets;  /* [num_layers+1] device copy (NodeFlow._layer_offsets)                */
  int32_t* blk_indptr;     /* concatenated per block: block b at blk_indptr + blk_indptr_off[b],
                              length |layer b+1| + 1, edge offsets relative to the block          */
  int32_t* blk_src;        /* concatenated per block at blk_src + blk_src_off[b]: position of the
                              edge's source inside layer b                                        */
  int32_t* sizes_pinned;   /* pinned host [2*PG_MAX_LAYERS]: [0..L] layer sizes (layer 0 first),
                              [PG_MAX_LAYERS .. ] edges per block; valid once `stream` drains     */
  int64_t cap_nodes;
  int64_t blk_indptr_off[PG_MAX_LAYERS];
  int64_t blk_src_off[PG_MAX_LAYERS];
  int32_t padded;          /* 0: layers concatenated back to back (the DGL layout).
                              1: fixed-shape layout for hipGraph replay — layer l starts at the sum of the
                                 capacities of layers < l (pg_sampler_capacity), unused entries are -1, and
                                 layer_offsets holds those fixed offsets; sizes_pinned still has the real sizes.
                              Either way every block's indptr is padded with empty rows up to its capacity. */
  uint32_t transpose_mask; /* bit b: also emit block b source-major (needs blk_tptr / blk_tdst)          */
  int32_t* blk_tptr;       /* NULL = no transposes. Block b at blk_tptr + blk_tptr_off[b]: [|layer b| + 1]
                              offsets into the block's edge range, padded with empty rows up to the
                              layer's capacity                                                      */
  int32_t* blk_tdst;       /* at blk_tdst + blk_src_off[b]: for source s, the destinations (positions
                              inside layer b+1) of its edges, ascending                             */
  int64_t blk_tptr_off[PG_MAX_LAYERS];
  int32_t* blk_theavy;     /* may be NULL. Block b at blk_theavy + blk_theavy_off[b]: [0] = number of sources
                              with more than PG_HEAVY_ROW edges, [1..] those sources (any order); room for
                              1 + cap_edges(b) / PG_HEAVY_ROW entries                               */
  int64_t blk_theavy_off[PG_MAX_LAYERS];
  int32_t* sizes_dev;      /* may be NULL: device copy of sizes_pinned (needed by pg_sampler_transpose)         */
  int32_t defer_transpose; /* 1: pg_sampler_sample does NOT build the source-major copies; the caller runs
                              pg_sampler_transpose(s, desc, its_stream) once the sample is complete           */
  int32_t _pad2;
} pg_nodeflow_desc_t;

/* indptr/indices: CSC of the partition (in-neighbours of v = indices[indptr[v]:indptr[v+1]],
 * ascending).  max_seeds = batch size, fanout = expand_factor, num_hops -> num_hops+1 layers.   */
int pg_sampler_create(int64_t num_vertices, const int64_t* indptr, const int32_t* indices,
                      int32_t max_seeds, int32_t fanout, int32_t num_hops, pg_sampler_t** out);
int pg_sampler_destroy(pg_sampler_t* s);
/* worst-case capacities for one batch: total nodes over all layers, per-block dst rows and edges
 * (layer l's capacity = cap_blk_rows[l-1] for l >= 1, cap_blk_edges[0] for l = 0) */
int pg_sampler_capacity(const pg_sampler_t* s, int64_t* cap_nodes, int64_t* cap_blk_rows /*[num_hops]*/,
                        int64_t* cap_blk_edges /*[num_hops]*/);
/* one minibatch: seeds (device int64[n_seeds]) -> NodeFlow. RNG key (seed, epoch, batch).
 * `out` names an output slot (a set of caller-owned buffers). Fully asynchronous: the call enqueues five plain launches
 * (fan-out <= 64; the wide chain and the DGL layout a few more) that take the call's scalars as kernel arguments — nothing
 * of the call lives in memory a later call rewrites, so the host never waits for the device here (rounds 1-2 replayed a
 * hipGraph over a pinned parameter block and blocked until the previous sample into the same slot had run). `seeds` must
 * stay valid until the launches have run. One chain at a time per handle (shared scratch: bitmaps, rank table, look-back
 * granules): calls on one stream, or on streams the caller orders. Not capturable into a caller's graph (per-launch
 * look-back tags and the bitmap parity are host state).                                                   */
int pg_sampler_sample(pg_sampler_t* s, const int64_t* seeds, int32_t n_seeds, uint64_t seed,
                      uint32_t epoch, uint32_t batch, const pg_nodeflow_desc_t* out, pg_stream_t stream);
/* Health of the chain's in-kernel look-backs: *lookback_timeouts = how many polls for a predecessor block's aggregate gave
 * up (bounded at a few seconds so that nothing can hang the GPU). Non-zero means at least one NodeFlow was garbage: a
 * caller checks it where it checks for lost miss rows (GraphedTrainer.synchronize). Synchronises with the device.    */
int pg_sampler_status(pg_sampler_t* s, int32_t* lookback_timeouts);
/* The source-major block copies of a sampled slot (transpose_mask, blk_tptr / blk_tdst / blk_theavy) on `stream`,
 * which must already be ordered after the sample (defer_transpose = 1, sizes_dev set). The 8 latency-bound launches
 * then leave the sampler's chain — the stage that bounds the pipeline once the features are cached — for a stream
 * with slack (the trainer's load stream). One caller at a time per sampler handle (shared scratch).          */
int pg_sampler_transpose(pg_sampler_t* s, const pg_nodeflow_desc_t* out, pg_stream_t stream);

/* Full-neighbour frontier expansion used by the L-hop closure (PaGraph/partition/utils.py:11-28):
 * marks every in-neighbour of frontier[0:n] in `bitmap` (uint64 words, bit v = vertex v) and
 * optionally marks the frontier itself; returns nothing else — pair with pg_bitmap_to_ids.       */
int pg_frontier_mark_neighbors(const int64_t* indptr, const int32_t* indices, const int64_t* frontier,
                               int64_t n, uint64_t* bitmap, int mark_self, pg_stream_t stream);
/* sorted ids of the set bits; count written to *count_dev (device int64). word_rank (uint32 per word,
 * may be NULL) receives the exclusive popcount prefix for later rank queries. scratch: >= 4*(n_words/1024 + 2) bytes. */
int pg_bitmap_to_ids(const uint64_t* bitmap, int64_t n_words, int64_t* out_ids, int64_t cap,
                     int64_t* count_dev, uint32_t* word_rank, void* scratch, pg_stream_t stream);

/* ------------------------------------------------------------------------
 * 3. Block aggregation  —  nf.block_compute(i, copy_src, mean|sum, ...) at
 *    PaGraph/model/gcn_nssc.py:71-74,139-142 and graphsage_nssc.py:98-111.
 * ------------------------------------------------------------------------ */
#define PG_REDUCE_MEAN 0
#define PG_REDUCE_SUM 1
#define PG_REDUCE_MAX 2 /* fn.max, the 'pool' aggregator (graphsage_nssc.py:106-110): element-wise maximum of the in-edge
                         * messages; forward entry points only (pg_spmm_fwd / _fwd_drop / _fwd_rows) — its backward needs
                         * the forward's input and output: pg_spmm_bwd_max / pg_spmm_bwd_gather_max below             */
/* out[v,:] = reduce_{e in [indptr[v],indptr[v+1])} h[src[e],:]   (0 when v has no edge)          */
int pg_spmm_fwd(const int32_t* indptr, const int32_t* src, const float* h, int32_t h_stride,
                int64_t n_dst, int32_t dim, int reduce, float* out, int32_t out_stride,
                pg_stream_t stream);
/* grad_h[src[e],:] += grad_out[v,:] * (mean ? 1/deg(v) : 1). grad_h must be zeroed by the caller. */

/* The same aggregation with the model's nn.Dropout (gcn_nssc.py:66-69, graphsage_nssc.py:86-89: dropout
 * on a layer's input right before it is aggregated) folded in: out = reduce(dropout(h)[src]) without
 * materialising dropout(h) or a mask. Counter-based mask: element (r, col) of h, piece = col / 4,
 *   q = (piece / 128) * 64 + piece % 64, half = (piece / 64) % 2, j = col % 4,
 *   w = Philox4x32-10(counter (r, q, tag, (uint32)*step), key (seed & 0xffffffff, seed >> 32)),
 *   u16 = j even ? w[2*half + j/2] & 0xffff : w[2*half + j/2] >> 16;  keep iff u16 >= threshold,
 *   kept values * 65536 / (65536 - threshold).
 * threshold = round(p * 65536) in [0, 65535]; 0 (or drop == NULL) = no dropout. `step` is a DEVICE
 * pointer (NULL = 0) so that a replayed hipGraph sees a new mask every step: the caller bumps it with
 * its own kernel. Forward needs dim % 4 == 0 and 16-byte aligned rows (else PG_ERR_UNSUPPORTED).     */
typedef struct pg_dropout {
  uint32_t threshold;
  uint32_t tag;          /* distinguishes call sites (layer index, rank) */
  uint64_t seed;
  const uint64_t* step;  /* device; NULL: step_value is used */
  uint64_t step_value;   /* the step counter's value as an immediate (step == NULL): for a launch whose caller keeps the
                          * count on the host — GraphedTrainer's early layer-0 aggregation, which runs on the load stream
                          * several batches ahead of the counter the optimiser's launch advances */
} pg_dropout_t;
int pg_spmm_fwd_drop(const int32_t* indptr, const int32_t* src, const float* h, int32_t h_stride,
                     int64_t n_dst, int32_t dim, int reduce, float* out, int32_t out_stride,
                     const pg_dropout_t* drop, pg_stream_t stream);
/* Aggregation straight from the feature cache — fetch_data (storage.py:176-204) fused into the layer-0
 * block_compute (gcn_nssc.py:66-74, graphsage_nssc.py:98-101): the source layer's rows are never
 * materialised. Row p of the source layer is read where it lies: slots[p] >= 0 -> cache[slots[p], :];
 * slots[p] <= -3 -> staged[-slots[p] - 3, :] (the block of miss rows the miss path copied to the device, in
 * miss-list order: pg_split_rows + pg_missq with a staged-only field); -1 / -2 contribute nothing. Bit-identical
 * to pg_gather_rows + pg_spmm_fwd_drop (same summation order, same dropout counters: row index = p).
 * Needs dim >= 256 and 16-byte aligned rows whose strides (cache, staged, out) are multiples of 4 floats >= dim
 * rounded up to 4 (else PG_ERR_UNSUPPORTED): dim % 4 != 0 (602) is read and written in whole 16-byte pieces, the
 * columns past dim are masked on the way in and written as zeros. drop may be NULL.
 * prof (device uint64[PG_PROF_WORDS * prof_ring], zero-initialised, may be NULL): the kernel usually runs inside a
 * replayed hipGraph, where HIP events cannot be attached to it, so it times itself with the device wall clock (100 MHz
 * ticks). Entry i = (*drop->step, or 0) % prof_ring, words: [0] start of the first wave; [1] start of the first wave of
 * the next dependent dense / head launch issued by the same host thread (pg_linear_fwd,
 * pg_head) — [1] - [0] is the time this kernel occupies its stream: body, drain, end-of-kernel release and the
 * successor's launch latency, the figure rocprofv3's End - Start of the dispatch agrees with; [2] edges aggregated;
 * [PG_PROF_END0 + PG_PROF_SHARD_STRIDE * s], s < PG_PROF_SHARDS: latest end-of-block stamp of the blocks b with
 * b % PG_PROF_SHARDS == s (the kernel body ends at their maximum; one 128-byte line per shard — sixteen shards on ONE
 * line serialised 3000 atomics at a single L2 channel and took the kernel from 14 to 25 us). The launch clears entry i + 1. */
#define PG_PROF_END0 16
#define PG_PROF_SHARDS 16
#define PG_PROF_SHARD_STRIDE 16
#define PG_PROF_WORDS (PG_PROF_END0 + PG_PROF_SHARDS * PG_PROF_SHARD_STRIDE)
typedef struct pg_row_source {
  const int32_t* slots;   /* device int32 [rows of the source layer] */
  const float* cache;     /* device; may be NULL when nothing is cached */
  const float* staged;    /* device; may be NULL when nothing can miss (full cache) */
  int32_t cache_stride;   /* floats */
  int32_t staged_stride;  /* floats */
  const int32_t* edge_slots; /* optional, device int32 [edges of the block]: slots[src[e]] per edge, composed beforehand
                              * (pg_compose_edge_slots, off the consumer's stream): one dependent index load less */
} pg_row_source_t;
/* edge_slots[e] = slots[src[e]] for e < n_edges (-2 where src[e] is outside [0, n_src): the unused tail of a
 * fixed-shape block)                                                                                    */
int pg_compose_edge_slots(const int32_t* src, int64_t n_edges, const int32_t* slots, int64_t n_src,
                          int32_t* edge_slots, pg_stream_t stream);
int pg_spmm_fwd_rows(const int32_t* indptr, const int32_t* src, const pg_row_source_t* rows, int64_t n_dst,
                     int32_t dim, int reduce, float* out, int32_t out_stride, const pg_dropout_t* drop,
                     uint64_t* prof, int32_t prof_ring, pg_stream_t stream);
/* Profiling aid: one-thread marker kernel, ring[(*step or 0) % ring_len] = device wall clock (100 MHz ticks). Launched
 * right behind a kernel inside a captured step it gives that kernel's TRUE end as the stream sees it (the write-back of
 * what the kernel left dirty in L2 included): a dispatch cannot start before its predecessor has completed.          */
int pg_prof_stamp(uint64_t* ring, int32_t ring_len, const uint64_t* step, pg_stream_t stream);
/* Backward of a block aggregation (the autograd side of nf.block_compute, gcn_nssc.py:71-74). ONE entry point with a
 * descriptor (round 6: pg_spmm_bwd, _drop, _max, _gather, _gather_dz, _gather_max took 10 - 17 positional arguments):
 *  - scatter form (tptr == NULL): grad_h[src[e],:] += grad_out[v,:] * (mean ? 1/deg(v) : 1) * mask(src[e],:) * scale over the
 *    destination-major block (fp32 atomics; grad_h zeroed by the caller; src / grad_h may be NULL only for an edgeless block).
 *  - gather form (tptr != NULL) over the block's source-major copy (pg_nodeflow_desc_t.blk_tptr / blk_tdst):
 *    grad_h[s,:] = mask(s,:) * scale * sum over s's edges of grad_out[dst,:] * (mean ? 1/deg(dst) : 1), destinations
 *    ascending. Every one of the n_src rows is written (no zero fill, no atomics, deterministic). indptr = the
 *    destination-major indptr (degrees; PG_REDUCE_SUM: unused). heavy (device, may be NULL): the block's hub list
 *    (blk_theavy: [0] = count, then sources with more than PG_HEAVY_ROW edges; heavy_cap = room behind the count) — those
 *    rows get a block each in a second launch instead of one lane group.
 *    dz (may be NULL): dZ of the NodeUpdate that produced the aggregated rows, when that was a skip-concat y = [z | relu(z)]
 *    (gcn_nssc.py:20-21; dim = 2 N): dz[s, j] = grad_h[s, j] + (act_out[s, j] > 0 ? grad_h[s, N + j] : 0), j < N — what
 *    pg_linear_bwd_w(act = 2) derives with a launch of its own. act_out = the forward input of the aggregation (row stride
 *    act_stride), dz [n_src, N] dense. Needs 16-byte pieces and dim / 4 <= 64 (else PG_ERR_UNSUPPORTED).
 *  - PG_REDUCE_MAX (both forms). DGL 0.4.1's rule (its ReduceMax functor's backward is `val == accum` [recollection; DGL is
 *    not in the reference checkout — parity unpinned]): a destination's gradient goes, whole, to EVERY in-edge whose message
 *    equals the maximum (ties are not split): grad_h[s, c] (+)= sum over s's edges (s -> v) of
 *    (x[s, c] == out[v, c] ? grad_out[v, c] : 0) * dmask(s, c) with x = dropout(h) as the forward saw it. h = the forward's
 *    input [n_src, dim], out = its output [n_dst, dim]; dz as above with act_out = h.
 * has_drop: the model's dropout in front of the aggregation (the same counter-based mask as the forward).        */
typedef struct pg_spmm_bwd_desc {
  const int32_t* indptr;
  const int32_t* src;
  const int32_t* tptr;
  const int32_t* tdst;
  const int32_t* heavy;
  const float* grad_out;
  float* grad_h;
  const float* h;
  const float* out;
  const float* act_out;
  float* dz;
  int64_t n_dst, n_src;
  int32_t go_stride, gh_stride, h_stride, out_stride, act_stride, dim, reduce, heavy_cap, has_drop, _pad;
  pg_dropout_t drop;
} pg_spmm_bwd_desc_t;
int pg_spmm_bwd(const pg_spmm_bwd_desc_t* desc, pg_stream_t stream);

/* Skinny dense step of the layers — NodeUpdate.forward at PaGraph/model/gcn_nssc.py:18-23 and graphsage_nssc.py:21-30 — on
 * fp32 MFMA: Z = X1[n,K1] W1^T + bias1 (+ X2[n,K2] W2^T + bias2: GraphSAGE's `fc_self(h) + fc_neigh(neigh)`, K2 == 0: none)
 * with W = nn.Linear's weight [N,K], N <= 64, any K (Reddit's 602: the last octet is zero-filled past K; W rows are read with
 * the widest aligned access K allows), X 16-byte aligned with stride % 4 == 0. The epilogue applies NodeUpdate's activation:
 * act 0: Y = Z; 1: Y = relu(Z); 2: Y = [Z | relu(Z)] (2N columns, the skip connection). The FIRST operand is dense (X1) or
 * read in place from a row source (X1rows, round 3: fc_self(h) of a layer whose 'features' were never gathered into a frame —
 * h = the cache / the miss queue's staged block through pg_row_source_t, exactly as pg_spmm_fwd_rows reads them; the same
 * bytes as pg_gather_rows + the dense form: bit-identical; rows with slot -1 / -2 count as zero rows) — exactly one of the two.
 * Returns PG_ERR_UNSUPPORTED outside that envelope (callers then use the library GEMM).
 * Round 6: ONE entry point with a descriptor (pg_linear_fwd / pg_linear2_fwd / pg_linear2_fwd_rows took 11 - 16 arguments). */
typedef struct pg_linear_fwd_desc {
  const float* X1;
  const pg_row_source_t* X1rows;
  const float* W1;
  const float* bias1;            /* [N] or NULL */
  const float* X2;               /* second operand, or NULL with K2 == 0 */
  const float* W2;
  const float* bias2;
  float* Y;
  int64_t n;
  int32_t x1_stride, K1, x2_stride, K2, y_stride, N, act, _pad;
} pg_linear_fwd_desc_t;
int pg_linear_fwd(const pg_linear_fwd_desc_t* desc, pg_stream_t stream);
/* dW1[N,K1] = dZ^T X1 and (db1 != NULL) db1[N] = column sums of dZ, any N and K, where dZ is derived from dY = dL/dY and the
 * saved output Yout according to `act` (act 0: dZ = dY, Yout may be NULL). K2 > 0: BOTH weight gradients of GraphSAGE's
 * NodeUpdate over the same dZ in ONE launch (round 4: dW2 = dZ^T X2; the same blocks and arithmetic as one launch per operand:
 * bit-identical partial rows and sums). The first operand is dense (X1) or read in place (X1rows), the second dense.
 * dz_scratch: device fp32 [n, N], required when act != 0; it holds dZ afterwards (callers reuse it for dX = dZ W).
 * partials1 / partials2: device fp32 scratch of pg_linear_bwd_w_scratch(n, K1 | K2, N) floats (per-row-chunk partial tiles).
 * sum_partials = 1: the chunks are summed in chunk order into dW / db by an ordered second launch (deterministic);
 * 0: they stay un-summed ([rows][N*K + N]) for pg_adam_step, dW / db are not written.
 * Round 6: ONE entry point (pg_linear_bwd_w / _ex / _rows / pg_linear2_bwd_w took 15 - 23 positional arguments). */
int64_t pg_linear_bwd_w_scratch(int64_t n, int32_t K, int32_t N);
typedef struct pg_linear_bwd_desc {
  const float* dY;
  const float* X1;
  const pg_row_source_t* X1rows;
  const float* X2;               /* NULL with K2 == 0 */
  const float* Yout;
  float* dW1;
  float* db1;
  float* dW2;
  float* db2;
  float* dz_scratch;
  float* partials1;
  float* partials2;
  int64_t n;
  int32_t dy_stride, x1_stride, K1, x2_stride, K2, N, yo_stride, act, sum_partials, _pad;
} pg_linear_bwd_desc_t;
int pg_linear_bwd_w(const pg_linear_bwd_desc_t* desc, pg_stream_t stream);

/* Loss head — torch.nn.CrossEntropyLoss() of examples/profile/pa_gcn.py:80,101-104 (pa_gs.py likewise):
 * log-softmax + NLL over logits[n, C], mean over the rows whose label is neither ignore_index nor
 * outside [0, C) (torch aborts on such labels; here they are ignored). Writes
 *   dlogits[n, C]  (may be NULL)  softmax(x) - onehot(label), NOT yet divided by the row count
 *   row_loss[n]    per-row loss (scratch),   meta[0] = mean loss (nan when no row counts),
 *   meta[1] = 1 / #counted rows.                                                                   */
int pg_xent_fwd(const float* logits, int32_t stride, const int64_t* labels, int64_t n, int32_t C,
                int64_t ignore_index, float* dlogits, int32_t d_stride, float* row_loss, float* meta,
                pg_stream_t stream);
/* gx = dlogits * meta[1] * (*grad_out)  (grad_out: device scalar, NULL = 1)                         */
int pg_xent_bwd(const float* dlogits, int32_t d_stride, int64_t n, int32_t C, const float* meta,
                const float* grad_out, float* gx, int32_t gx_stride, pg_stream_t stream);

/* Output head of the sampled GCN in one pass (pg_head.hip): the last block's aggregation with the model's
 * dropout (gcn_nssc.py:66-74), the output NodeUpdate z = W agg + b (gcn_nssc.py:18,58: no activation), the
 * CrossEntropyLoss of pa_gcn.py:80,101-104 and their gradients. h [n_src, K] (K <= 64), W [C, K] (C <= 64).
 *   logits [n_dst, C] (may be NULL)     dagg [n_dst, K] = d loss / d agg (feed pg_spmm_bwd_gather / _bwd with it;
 *   dW [C, K], db_loss [C + 1] = db then the mean loss.   Everything is already scaled by *grad_scale_dev / #counted (device
 *   scalar = d(objective)/d(loss), NULL = 1), #counted read from *n_valid_dev (pg_gather_labels); labels outside [0, C) other than ignore_index are
 *   not counted. partials: pg_gcn_head_scratch(n_dst, K, C) floats. drop may be NULL. Deterministic.          */
int64_t pg_gcn_head_scratch(int64_t n_dst, int32_t K, int32_t C);
/* floats between two blocks' rows of that scratch: [C * K] dW | [C] db | [1] loss, padded to a multiple of 4 — the
 * `part_len` a caller puts into pg_adam_tensor_t when it lets the optimiser add the rows up (sum_partials = 0)   */
int32_t pg_gcn_head_row_len(int32_t K, int32_t C);
/* pg_head / pg_linear_bwd_w with `sum_partials` = 0 / without PG_HEAD_SUM_PARTIALS leave the per-block / per-chunk partial rows
 * un-summed in `partials` ([rows][C*K + C + 1] resp. [rows][N*K + N], rows = scratch size / row length) for
 * pg_adam_step; dW / db(_loss) are then not written.                                                     */
/* GraphSAGE's output NodeUpdate z = fc_neigh(agg) + fc_self(h_self) (graphsage_nssc.py:24, the last layer of
 * graphsage_nssc.py:55-72; round 4) is the same head with the descriptor's self_* fields set: h_self [n_dst, >= Ks] is the
 * destinations' own input (not aggregated, not dropped), W_self [C, Ks], bias_self [C] or NULL; K + Ks <= 64.
 * dself [n_dst, Ks] = dZ W_self. Partial rows: pg_gcn_head_scratch(n_dst, K + Ks, C) floats laid out
 * [C x K] dW | [C x Ks] dW_self | [C] db | loss per block (pg_gcn_head_row_len(K + Ks, C) apart); with PG_HEAD_SUM_PARTIALS dW
 * receives [C x K | C x Ks] contiguous and db_loss [C + 1]; both biases have the gradient db.
 *
 * flags: PG_HEAD_SUM_PARTIALS — the per-block partial rows are summed into dW / db_loss by an ordered second launch (without
 * it they stay un-summed in `partials` for pg_adam_step, dW / db_loss are not written); PG_HEAD_DAGG_PER_EDGE — under
 * PG_REDUCE_MEAN dagg[v] leaves already divided by v's in-degree in the block (what each in-edge carries back): feed it to
 * pg_spmm_bwd with PG_REDUCE_SUM, same operations in the same order without the backward's degree loads.
 * Round 6: ONE entry point with a descriptor (pg_gcn_head / pg_gcn_head_ex / pg_sage_head took 21 - 28 positional arguments). */
#define PG_HEAD_SUM_PARTIALS 1
#define PG_HEAD_DAGG_PER_EDGE 2
typedef struct pg_head_desc {
  const int32_t* indptr;         /* the last block, destination-major */
  const int32_t* src;
  const float* h;                /* [n_src, K] the layer below's output */
  const float* W;                /* [C, K] */
  const float* bias;             /* [C] or NULL */
  const int64_t* labels;         /* [n_dst] */
  const int32_t* n_valid_dev;    /* labels the loss counts (pg_gather_labels) */
  const float* grad_scale_dev;   /* device scalar d(objective)/d(loss), NULL = 1 */
  int64_t ignore_index, n_dst;
  int32_t h_stride, K, C, reduce, flags, has_drop;
  pg_dropout_t drop;             /* the model's dropout in front of the aggregation (has_drop) */
  float* logits;                 /* [n_dst, C] or NULL */
  float* dagg;                   /* [n_dst, K] */
  float* partials;               /* pg_gcn_head_scratch(n_dst, K + Ks, C) floats */
  float* dW;                     /* [C, K (+ Ks)] */
  float* db_loss;                /* [C + 1]: db, then the mean loss */
  /* GraphSAGE's self operand (Ks == 0: none) */
  const float* h_self;
  const float* W_self;
  const float* bias_self;
  float* dself;
  int32_t hs_stride, Ks;
} pg_head_desc_t;
int pg_head(const pg_head_desc_t* desc, pg_stream_t stream);

/* The load-stream half of one minibatch as ONE call (round 5; csrc/pg_pipeline.hip) for a table that is resident in HBM —
 * pa_gcn.py:86-91's 'gpu-load' range: the slot look-up of fetch_from_cache (storage.py:207-216; the rows are then read in place),
 * the sampled blocks' source-major copies, the aggregations of raw rows that run ahead of their step, labels[batch_nids]
 * (pa_gcn.py:89-90). Every argument that does not change from batch to batch is resolved once per ring slot in the plan; the
 * call issues, on plan->load_stream: wait(ev_sampled) · pg_slots_full · pg_sampler_transpose ·
 * pg_spmm_fwd_rows per early block (drop.step_value := drop_step_value) · pg_gather_labels_sc · record(ev_ready).
 * Parts whose pointers / counts are NULL / 0 are skipped. ev_* are hipEvent_t handles owned by the caller. */
typedef struct pg_batch_early {
  const int32_t* indptr;
  const int32_t* src;
  pg_row_source_t rows;
  int64_t n_dst;
  int32_t dim, reduce;
  float* out;
  int32_t out_stride, has_drop;
  pg_dropout_t drop;
  uint64_t* prof;
  int32_t prof_ring, _pad;
} pg_batch_early_t;
typedef struct pg_batch_plan {
  pg_stream_t load_stream;
  void* ev_sampled;          /* recorded by the caller behind pg_sampler_sample on the sampler's stream */
  void* ev_ready;            /* recorded here: the step's consumer waits for it */
  /* slots of the rows that are read in place */
  const int64_t* ids;
  int64_t rows;
  const int32_t* slot_map;
  int32_t* slots_out;
  uint64_t* stats;           /* may be NULL */
  /* source-major block copies */
  pg_sampler_t* sampler;     /* may be NULL */
  pg_nodeflow_desc_t desc;
  int32_t transpose, n_early;
  /* aggregations that run ahead of the step */
  pg_batch_early_t early[PG_MAX_LAYERS];
  /* labels of the seeds */
  const int64_t* label_ids;
  int64_t n_label_rows;
  const int64_t* labels;
  int64_t labels_len, label_fill;
  int64_t* label_out;
  int32_t* n_valid;
  int32_t* label_scratch;
} pg_batch_plan_t;
int pg_batch_prepare(const pg_batch_plan_t* plan, uint64_t drop_step_value);



/* Optimiser step — torch.optim.Adam(model.parameters(), lr, weight_decay) of examples/profile/pa_gcn.py:137-139
 * (amsgrad off, maximize off), same arithmetic as torch's: g += wd * p; m = b1 m + (1 - b1) g;
 * v = b2 v + (1 - b2) g^2; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), t = *step_dev + 1.
 * ONE launch for up to PG_ADAM_MAX_TENSORS fp32 tensors, described by a pg_adam_desc_t (round 6: the three entry points
 * with up to 24 positional arguments are gone). *step_dev (device int64, completed steps) is advanced by the kernel's last
 * block; ticket_dev is one device uint32 the caller zeroes once; bump_dev (device int64, may be NULL) is advanced by one
 * together with *step_dev — the model's dropout step counter, which saves the step's separate increment launch.
 *
 * Per tensor: partials == NULL -> the gradient is read from grad[] as is. Otherwise element o of the gradient is the sum
 * over part_chunks rows of partials[c * part_len + part_off + o] — the un-summed per-chunk rows pg_linear_bwd_w_ex /
 * pg_head leave with sum_partials = 0 — added in exactly pg_sum_partials' order (bit-identical) and stored to grad[].
 * partials2 (may be NULL): a SECOND set of rows for a parameter that is applied twice per step — GraphSAGE's NodeUpdate
 * `lid` runs on every block >= lid (graphsage_nssc.py:92-131) — gradient = sum(partials) + sum(partials2), what autograd's
 * accumulation of the two summed contributions yields. is_adam == 0: reduce only (param / exp_avg / exp_avg_sq may be
 * NULL) — pg_head's loss scalar lives in the same rows.
 *
 * mode PG_ADAM_FULL: sums + update. PG_ADAM_REDUCE_ONLY: only the sums are written to grad[] (every tensor must have
 * partials; step / ticket / bump are not touched) — the N > 1 step: grad[] are views of the flat buffer the gradient
 * all-reduce (pa_gcn.py:65,96: DDP) works on, and a second, plain PG_ADAM_FULL launch applies the update behind it.  */
#define PG_ADAM_MAX_TENSORS 16
#define PG_ADAM_FULL 0
#define PG_ADAM_REDUCE_ONLY 1
typedef struct pg_adam_tensor {
  float* param;
  float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  int64_t numel;
  const float* partials;
  const float* partials2;
  int32_t part_chunks, part_len, part_off;
  int32_t part2_chunks, part2_len, part2_off;
  int32_t is_adam, _pad;
} pg_adam_tensor_t;
typedef struct pg_adam_desc {
  int32_t n_tensors, mode;
  float lr, beta1, beta2, eps, weight_decay, _pad;
  int64_t* step_dev;
  uint32_t* ticket_dev;
  int64_t* bump_dev;
  pg_adam_tensor_t t[PG_ADAM_MAX_TENSORS];
} pg_adam_desc_t;
int pg_adam_step(const pg_adam_desc_t* desc, pg_stream_t stream);
/* Mirror a step counter to the host: from now on every pg_adam_step launch given `step_dev` also writes the new step
 * count to *mirror_host (pinned host memory mapped into the device; NULL = stop), from its last block, after the device
 * counter. A launch thread that recycles per-step buffers can then POLL "step n has run" in its own memory instead of
 * recording an event on the compute stream (an event costs the stream that records it ~5 us, ~13 when another stream
 * waits for it). The kernels are the last launches of a training step, so "the optimiser of step n has started its last
 * block" implies every earlier kernel of that step has finished. Up to 16 registered counters per process.          */
int pg_adam_step_mirror(int64_t* step_dev, int64_t* mirror_host);

/* ------------------------------------------------------------------------
 * 4. Offline partitioning (host)  —  PaGraph/partition/dg.py:59-103
 * ------------------------------------------------------------------------
 * CSC of the full graph on the host. belongs_out[V] (int8, -1 = unassigned). Returns r_vnum/p_vnum.
 * r_mask_out: P*V bytes (r_belongs, dg.py:64) or NULL. 2 <= P <= 127 (belongs is int8, dg.py:63; argsort[-2:]).
 * The arg-max of dg.py:30-35 takes the last two of np.argsort(score) with numpy's DEFAULT kind, whose tie order is part
 * of the result (every first assignment ties): stable up to 16 partitions, numpy's scalar introsort above
 * (pg_np_argsort_f64). numpy on AVX2 / AVX-512 hosts dispatches that call to x86-simd-sort, which orders ties
 * differently — the reference's own output is machine-dependent; this is its portable scalar behaviour.       */
int pg_dg_partition(int64_t V, const int64_t* indptr, const int32_t* indices, const int64_t* train_nids,
                    int64_t n_train, int32_t P, int32_t hops, int8_t* belongs_out, uint8_t* r_mask_out,
                    int64_t* p_vnum_out, int64_t* r_vnum_out);
/* the same with n_threads host threads when hops == 2: n_threads - 1 builders construct the two-hop neighbour sets
 * (which depend on the graph only) ahead of ONE committer that applies dg.py:71-83 strictly in train order, so the
 * partition is bit-identical. Other hops values and n_threads <= 1 run the sequential code. pg_dg_partition = this with n_threads from env PG_DG_THREADS (default 1). */
int pg_dg_partition_mt(int64_t V, const int64_t* indptr, const int32_t* indices, const int64_t* train_nids,
                       int64_t n_train, int32_t P, int32_t hops, int8_t* belongs_out, uint8_t* r_mask_out,
                       int64_t* p_vnum_out, int64_t* r_vnum_out, int32_t n_threads);
/* The same partition with the neighbour sets built on the DEVICE (round 6; csrc/pg_dg_gpu.hip): the CSC lives in HBM
 * (indptr_dev / indices_dev), everything else on the host as above. Batches of consecutive train vertices are expanded against
 * a snapshot of the assignment state; per vertex the device hands back the per-partition count of settled members (dg.py:47-50),
 * the members assigned inside the batch, and the members some partition's r_belongs still lacked; the host applies dg.py:51-83
 * strictly in train order with exact bitmaps — bit-identical to pg_dg_partition (hops 1 and 2). 10M / 100M graph, hops 2:
 * 4 s instead of 68 s; 10^8 / 10^9: 1 min instead of ~1000 s. Allocates its own scratch (a few hundred de-duplication
 * bitmaps of V / 4 bytes, at most 16 GB of them; one list buffer of max(V, min(32M, 64 V)) 4-byte entries per partition, on the
 * device and pinned on the host) and frees it before returning. PG_ERR_UNSUPPORTED (fall back to pg_dg_partition_mt):
 * P > 16, hops > 2, V >= 2^28, train ids not strictly ascending. Synchronises `stream`.
 * stats: fresh_entries = entries on the (vertex, candidate partition) lists, second_walks = multisets whose put-aside list
 * overflowed the scratch (walked three times instead of once), candidate_misses = batches cut short by a wrong guess.  */
typedef struct pg_dg_gpu_stats {
  int64_t batches, batches_redone, largest_batch, fresh_entries, corr_entries, workgroups, candidate_misses, second_walks;
  double seconds_total, seconds_expand, seconds_lists, seconds_commit, seconds_apply;
} pg_dg_gpu_stats_t;
int pg_dg_partition_gpu(int64_t V, const int64_t* indptr_dev, const int32_t* indices_dev, const int64_t* train_nids,
                        int64_t n_train, int32_t P, int32_t hops, int8_t* belongs_out, uint8_t* r_mask_out,
                        int64_t* vnum_out /* [2][P]: p_vnum then r_vnum, may be NULL */, pg_dg_gpu_stats_t* stats,
                        pg_stream_t stream);
/* order[0..n) = np.argsort(v) (default kind) of n <= 127 float64 on numpy 2.2's scalar path (npysort/quicksort.cpp,
 * heapsort.cpp restated): what dg's arg-max sorts its scores with. Exported so the tests can pin it against numpy.  */
int pg_np_argsort_f64(const double* v, int32_t n, int32_t* order);

/* ------------------------------------------------------------------------
 * 5. Synthetic inputs  —  PaGraph/data/preprocess.py:50-114 + PaRMAT (README.md:36-41)
 * ------------------------------------------------------------------------ */
/* RMAT candidate edges [first, first+n): integer-threshold recursive quadrant descent keyed by
 * (seed, edge index). a,b,c in Q32 fixed point (0.45 -> 0.45*2^32). out on device.               */
int pg_rmat_edges(uint64_t seed, int32_t scale, uint32_t a_q32, uint32_t b_q32, uint32_t c_q32,
                  int64_t first, int64_t n, int64_t* src, int64_t* dst, pg_stream_t stream);
/* feat[r, c] = U[0,1) fp32 from Philox keyed (seed; row0+r, c) — rows [row0,row0+rows)          */
int pg_random_features(uint64_t seed, int64_t row0, int64_t rows, int32_t dim, float* out,
                       int64_t out_stride, pg_stream_t stream);

/* ------------------------------------------------------------------------
 * 6. Timing helper for bench.py — HIP events on the stream the kernels run on.
 * ------------------------------------------------------------------------ */
int pg_timer_create(pg_timer_t** t);
int pg_timer_destroy(pg_timer_t* t);
int pg_timer_start(pg_timer_t* t, pg_stream_t stream);
int pg_timer_stop(pg_timer_t* t, pg_stream_t stream);
/* blocks on the stop event */
int pg_timer_elapsed_ms(pg_timer_t* t, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* PAGRAPH_HIP_H */
