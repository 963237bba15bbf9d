"""ctypes binding of libpagraph_hip.so (the C-ABI declared in include/pagraph_hip.h).

There is NO CPU fallback: if the shared library is missing or a call fails,
this raises.  torch is imported first on purpose — the library is linked against
libamdhip64.so.7, and loading it after torch makes the dynamic loader reuse the
HIP runtime torch already mapped (one runtime, shared streams and allocations).
"""
import ctypes
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpagraph_hip.so")
# PG_BOUNDS=1: the debug build that bounds-checks ids (`make -C pagraph_amd/csrc bounds`; include/pagraph_hip.h
# pg_bounds_*): every tensor handed to the library has its extent registered (ptr / note below), kernels check the indices
# they follow against those extents, bounds_report() says which kernel met which bad index
BOUNDS_LIB_PATH = os.path.join(_HERE, "libpagraph_hip_bounds.so")
BOUNDS = False

PG_MAX_FIELDS = 4
PG_MAX_LAYERS = 8
PG_HEAVY_ROW = 32
PG_ADAM_MAX_TENSORS = 16
PG_HEAD_SUM_PARTIALS, PG_HEAD_DAGG_PER_EDGE = 1, 2
PG_ADAM_FULL, PG_ADAM_REDUCE_ONLY = 0, 1
PG_REDUCE_MEAN = 0
PG_REDUCE_SUM = 1
PG_REDUCE_MAX = 2
PG_PROF_END0, PG_PROF_SHARDS, PG_PROF_SHARD_STRIDE = 16, 16, 16      # pg_spmm_fwd_rows' self-timing ring (include/pagraph_hip.h)
PG_PROF_WORDS = PG_PROF_END0 + PG_PROF_SHARDS * PG_PROF_SHARD_STRIDE

c_i32, c_i64, c_u32, c_u64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint32, ctypes.c_uint64
vp = ctypes.c_void_p


class PgField(ctypes.Structure):
    _fields_ = [("cache", vp), ("out", vp), ("dim", c_i32), ("cache_stride", c_i32),
                ("out_stride", c_i32), ("_pad", c_i32)]


class PgNodeflowDesc(ctypes.Structure):
    _fields_ = [("node_mapping", vp), ("layer_offsets", vp), ("blk_indptr", vp), ("blk_src", vp),
                ("sizes_pinned", vp), ("cap_nodes", c_i64),
                ("blk_indptr_off", c_i64 * PG_MAX_LAYERS), ("blk_src_off", c_i64 * PG_MAX_LAYERS),
                ("padded", c_i32), ("transpose_mask", c_u32), ("blk_tptr", vp), ("blk_tdst", vp),
                ("blk_tptr_off", c_i64 * PG_MAX_LAYERS), ("blk_theavy", vp), ("blk_theavy_off", c_i64 * PG_MAX_LAYERS),
                ("sizes_dev", vp), ("defer_transpose", c_i32), ("_pad2", c_i32)]


class PgMissqField(ctypes.Structure):
    _fields_ = [("table", vp), ("table_stride", c_i64), ("dim", c_i32), ("_pad", c_i32)]


class PgRowSource(ctypes.Structure):
    _fields_ = [("slots", vp), ("cache", vp), ("staged", vp), ("cache_stride", c_i32), ("staged_stride", c_i32),
                ("edge_slots", vp)]


class PgMissqStats(ctypes.Structure):
    _fields_ = [(n, c_i64) for n in ("jobs", "rows", "waits_by_event", "waits_by_spin_kernel", "spared_jobs", "rescued_chunks")] + \
               [(n, ctypes.c_double) for n in ("us_submit_to_published", "us_cpu_gather", "us_enqueue", "us_submit_to_done",
                                               "max_us_wait_published", "max_us_cpu_gather", "max_us_enqueue",
                                               "max_us_submit_to_done")] + \
               [("sdma_engine_mask", c_u32), ("_pad", c_u32), ("engine_GBps", ctypes.c_double * 16)]


class PgMissList(ctypes.Structure):
    _fields_ = [("pos", vp), ("fullid", vp), ("count", vp)]


class PgDedup(ctypes.Structure):
    _fields_ = [("n_ranges", c_i32), ("lo", c_i32 * (PG_MAX_LAYERS + 1)), ("sorted_mask", c_u32), ("dup_pos", vp),
                ("dup_src", vp), ("dup_count", vp)]


class PgDropout(ctypes.Structure):
    _fields_ = [("threshold", c_u32), ("tag", c_u32), ("seed", c_u64), ("step", vp), ("step_value", c_u64)]


class PgSpmmBwdDesc(ctypes.Structure):
    _fields_ = [("indptr", vp), ("src", vp), ("tptr", vp), ("tdst", vp), ("heavy", vp), ("grad_out", vp), ("grad_h", vp), ("h", vp),
                ("out", vp), ("act_out", vp), ("dz", vp), ("n_dst", c_i64), ("n_src", c_i64), ("go_stride", c_i32),
                ("gh_stride", c_i32), ("h_stride", c_i32), ("out_stride", c_i32), ("act_stride", c_i32), ("dim", c_i32),
                ("reduce", c_i32), ("heavy_cap", c_i32), ("has_drop", c_i32), ("_pad", c_i32), ("drop", PgDropout)]


class PgLinearFwdDesc(ctypes.Structure):
    _fields_ = [("X1", vp), ("X1rows", vp), ("W1", vp), ("bias1", vp), ("X2", vp), ("W2", vp), ("bias2", vp), ("Y", vp),
                ("n", c_i64), ("x1_stride", c_i32), ("K1", c_i32), ("x2_stride", c_i32), ("K2", c_i32), ("y_stride", c_i32),
                ("N", c_i32), ("act", c_i32), ("_pad", c_i32)]


class PgLinearBwdDesc(ctypes.Structure):
    _fields_ = [("dY", vp), ("X1", vp), ("X1rows", vp), ("X2", vp), ("Yout", vp), ("dW1", vp), ("db1", vp), ("dW2", vp),
                ("db2", vp), ("dz_scratch", vp), ("partials1", vp), ("partials2", vp), ("n", c_i64), ("dy_stride", c_i32),
                ("x1_stride", c_i32), ("K1", c_i32), ("x2_stride", c_i32), ("K2", c_i32), ("N", c_i32), ("yo_stride", c_i32),
                ("act", c_i32), ("sum_partials", c_i32), ("_pad", c_i32)]


class PgHeadDesc(ctypes.Structure):
    _fields_ = [("indptr", vp), ("src", vp), ("h", vp), ("W", vp), ("bias", vp), ("labels", vp), ("n_valid_dev", vp),
                ("grad_scale_dev", vp), ("ignore_index", c_i64), ("n_dst", c_i64), ("h_stride", c_i32), ("K", c_i32), ("C", c_i32),
                ("reduce", c_i32), ("flags", c_i32), ("has_drop", c_i32), ("drop", PgDropout), ("logits", vp), ("dagg", vp),
                ("partials", vp), ("dW", vp), ("db_loss", vp), ("h_self", vp), ("W_self", vp), ("bias_self", vp), ("dself", vp),
                ("hs_stride", c_i32), ("Ks", c_i32)]


class PgBatchEarly(ctypes.Structure):
    _fields_ = [("indptr", vp), ("src", vp), ("rows", PgRowSource), ("n_dst", c_i64), ("dim", c_i32), ("reduce", c_i32),
                ("out", vp), ("out_stride", c_i32), ("has_drop", c_i32), ("drop", PgDropout), ("prof", vp),
                ("prof_ring", c_i32), ("_pad", c_i32)]


class PgBatchPlan(ctypes.Structure):
    _fields_ = [("load_stream", vp), ("ev_sampled", vp), ("ev_ready", vp), ("ids", vp), ("rows", c_i64), ("slot_map", vp),
                ("slots_out", vp), ("stats", vp), ("sampler", vp), ("desc", PgNodeflowDesc), ("transpose", c_i32),
                ("n_early", c_i32), ("early", PgBatchEarly * PG_MAX_LAYERS), ("label_ids", vp), ("n_label_rows", c_i64),
                ("labels", vp), ("labels_len", c_i64), ("label_fill", c_i64), ("label_out", vp), ("n_valid", vp),
                ("label_scratch", vp)]


class PgAdamTensor(ctypes.Structure):
    _fields_ = [("param", vp), ("grad", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("numel", c_i64), ("partials", vp),
                ("partials2", vp), ("part_chunks", c_i32), ("part_len", c_i32), ("part_off", c_i32), ("part2_chunks", c_i32),
                ("part2_len", c_i32), ("part2_off", c_i32), ("is_adam", c_i32), ("_pad", c_i32)]


class PgAdamDesc(ctypes.Structure):
    _fields_ = [("n_tensors", c_i32), ("mode", c_i32), ("lr", ctypes.c_float), ("beta1", ctypes.c_float),
                ("beta2", ctypes.c_float), ("eps", ctypes.c_float), ("weight_decay", ctypes.c_float), ("_pad", ctypes.c_float),
                ("step_dev", vp), ("ticket_dev", vp), ("bump_dev", vp), ("t", PgAdamTensor * PG_ADAM_MAX_TENSORS)]


class PgDgGpuStats(ctypes.Structure):
    _fields_ = [(n, c_i64) for n in ("batches", "batches_redone", "largest_batch", "fresh_entries", "corr_entries", "workgroups", "candidate_misses", "second_walks")] + \
               [(n, ctypes.c_double) for n in ("seconds_total", "seconds_expand", "seconds_lists", "seconds_commit", "seconds_apply")]


class PgError(RuntimeError):
    pass


_SIGS = {
    "pg_version": (ctypes.c_int, []),
    "pg_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "pg_last_hip_error": (ctypes.c_int, []),
    "pg_bounds_enabled": (ctypes.c_int, []),
    "pg_bounds_region": (ctypes.c_int, [vp, c_i64]),
    "pg_bounds_report": (ctypes.c_int, [ctypes.POINTER(c_u64), ctypes.c_char_p, c_i32, c_i32]),
    "pg_device_cu_count": (ctypes.c_int, []),
    "pg_batch_prepare": (ctypes.c_int, [ctypes.POINTER(PgBatchPlan), c_u64]),
    "pg_tape_from_graph": (ctypes.c_int, [vp, ctypes.POINTER(vp), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32)]),
    "pg_tape_launch": (ctypes.c_int, [vp, vp]),
    "pg_tape_destroy": (ctypes.c_int, [vp]),
    "pg_slot_map_reset": (ctypes.c_int, [vp, c_i64, vp]),
    "pg_slot_map_assign": (ctypes.c_int, [vp, vp, c_i64, vp]),
    "pg_slot_map_export": (ctypes.c_int, [vp, c_i64, vp, vp, vp]),
    "pg_gather_rows": (ctypes.c_int, [vp, c_i64, vp, vp, ctypes.POINTER(PgField), ctypes.c_int, ctypes.POINTER(PgMissList), vp, vp,
                                      vp, ctypes.POINTER(PgDedup), vp]),
    "pg_split_rows": (ctypes.c_int, [vp, c_i64, vp, vp, ctypes.POINTER(PgMissList), vp, vp, ctypes.POINTER(PgDedup), vp]),
    "pg_scatter_rows_dups": (ctypes.c_int, [vp, vp, vp, c_i64, vp, c_i32, vp, c_i32, c_i32, vp]),
    "pg_gather_rows_presplit": (ctypes.c_int, [vp, c_i64, ctypes.POINTER(PgField), ctypes.c_int, vp, vp]),
    "pg_gather_rows_full": (ctypes.c_int, [vp, c_i64, ctypes.POINTER(PgField), ctypes.c_int, vp]),
    "pg_gather_labels": (ctypes.c_int, [vp, c_i64, vp, c_i64, c_i64, vp, vp, vp]),
    "pg_gather_labels_sc": (ctypes.c_int, [vp, c_i64, vp, c_i64, c_i64, vp, vp, vp, vp]),
    "pg_scatter_rows": (ctypes.c_int, [vp, vp, c_i64, vp, c_i32, vp, c_i32, vp]),
    "pg_scatter_rows_range": (ctypes.c_int, [vp, vp, c_i64, vp, c_i32, vp, c_i32, c_i32, vp]),
    "pg_host_gather_rows": (ctypes.c_int, [vp, c_i64, c_i32, vp, c_i64, vp, ctypes.c_int]),
    "pg_scatter_rows_from_host": (ctypes.c_int, [vp, c_i64, vp, vp, c_i64, vp, c_i32, vp, c_i32, vp]),
    "pg_scatter_rows_from_host_tail": (ctypes.c_int, [vp, c_i64, vp, vp, c_i64, vp, c_i32, c_i32, vp, c_i32, vp]),
    "pg_missq_set_cpu_share": (ctypes.c_int, [vp, c_i32]),
    "pg_missq_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_i64, ctypes.POINTER(PgMissqField), ctypes.c_int,
                                       ctypes.c_int, ctypes.POINTER(vp)]),
    "pg_missq_destroy": (ctypes.c_int, [vp]),
    "pg_missq_slot_buffers": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]),
    "pg_missq_submit": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), vp, vp]),
    "pg_missq_stats": (ctypes.c_int, [vp, ctypes.POINTER(PgMissqStats), ctypes.c_int]),
    "pg_missq_slot_dup_buffers": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                                 ctypes.POINTER(vp)]),
    "pg_missq_staged_stride": (ctypes.c_int, [vp, ctypes.c_int, ctypes.POINTER(c_i32)]),
    "pg_scatter_rows_strided": (ctypes.c_int, [vp, c_i32, vp, vp, c_i64, vp, c_i32, vp, c_i32, c_i32, c_i32, vp]),
    "pg_missq_slot_staged": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp)]),
    "pg_missq_wait": (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.POINTER(c_i32)]),
    "pg_missq_wait_device": (ctypes.c_int, [vp, ctypes.c_int, vp]),
    "pg_missq_wait_idle": (ctypes.c_int, [vp, ctypes.c_int]),
    "pg_missq_timed_out": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_int)]),
    "pg_missq_drain": (ctypes.c_int, [vp]),
    "pg_missq_device_tail": (ctypes.c_int, [vp, ctypes.c_int, vp]),
    "pg_missq_order_after_tail": (ctypes.c_int, [vp, ctypes.c_int, vp]),
    "pg_missq_copy_log": (ctypes.c_int, [vp, vp, vp, c_i64, ctypes.POINTER(c_i64)]),
    "pg_sampler_create": (ctypes.c_int, [c_i64, vp, vp, c_i32, c_i32, c_i32, ctypes.POINTER(vp)]),
    "pg_sampler_destroy": (ctypes.c_int, [vp]),
    "pg_sampler_capacity": (ctypes.c_int, [vp, ctypes.POINTER(c_i64), ctypes.POINTER(c_i64), ctypes.POINTER(c_i64)]),
    "pg_sampler_status": (ctypes.c_int, [vp, ctypes.POINTER(c_i32)]),
    "pg_sampler_sample": (ctypes.c_int, [vp, vp, c_i32, c_u64, c_u32, c_u32, ctypes.POINTER(PgNodeflowDesc), vp]),
    "pg_sampler_transpose": (ctypes.c_int, [vp, ctypes.POINTER(PgNodeflowDesc), vp]),
    "pg_frontier_mark_neighbors": (ctypes.c_int, [vp, vp, vp, c_i64, vp, ctypes.c_int, vp]),
    "pg_bitmap_to_ids": (ctypes.c_int, [vp, c_i64, vp, c_i64, vp, vp, vp, vp]),
    "pg_spmm_fwd": (ctypes.c_int, [vp, vp, vp, c_i32, c_i64, c_i32, ctypes.c_int, vp, c_i32, vp]),
    "pg_spmm_bwd": (ctypes.c_int, [ctypes.POINTER(PgSpmmBwdDesc), vp]),
    "pg_spmm_fwd_drop": (ctypes.c_int, [vp, vp, vp, c_i32, c_i64, c_i32, ctypes.c_int, vp, c_i32, vp, vp]),
    "pg_spmm_fwd_rows": (ctypes.c_int, [vp, vp, ctypes.POINTER(PgRowSource), c_i64, c_i32, ctypes.c_int, vp, c_i32, vp, vp,
                                        c_i32, vp]),
    "pg_prof_stamp": (ctypes.c_int, [vp, c_i32, vp, vp]),
    "pg_slots_full": (ctypes.c_int, [vp, c_i64, vp, vp, vp, vp]),
    "pg_compose_edge_slots": (ctypes.c_int, [vp, c_i64, vp, c_i64, vp, vp]),
    "pg_linear_fwd": (ctypes.c_int, [ctypes.POINTER(PgLinearFwdDesc), vp]),
    "pg_linear_bwd_w_scratch": (c_i64, [c_i64, c_i32, c_i32]),
    "pg_linear_bwd_w": (ctypes.c_int, [ctypes.POINTER(PgLinearBwdDesc), vp]),
    "pg_xent_fwd": (ctypes.c_int, [vp, c_i32, vp, c_i64, c_i32, c_i64, vp, c_i32, vp, vp, vp]),
    "pg_xent_bwd": (ctypes.c_int, [vp, c_i32, c_i64, c_i32, vp, vp, vp, c_i32, vp]),
    "pg_gcn_head_scratch": (c_i64, [c_i64, c_i32, c_i32]),
    "pg_gcn_head_row_len": (c_i32, [c_i32, c_i32]),
    "pg_head": (ctypes.c_int, [ctypes.POINTER(PgHeadDesc), vp]),
    "pg_adam_step_mirror": (ctypes.c_int, [vp, vp]),
    "pg_adam_step": (ctypes.c_int, [ctypes.POINTER(PgAdamDesc), vp]),
    "pg_dg_partition": (ctypes.c_int, [c_i64, vp, vp, vp, c_i64, c_i32, c_i32, vp, vp, vp, vp]),
    "pg_dg_partition_mt": (ctypes.c_int, [c_i64, vp, vp, vp, c_i64, c_i32, c_i32, vp, vp, vp, vp, c_i32]),
    "pg_dg_partition_gpu": (ctypes.c_int, [c_i64, vp, vp, vp, c_i64, c_i32, c_i32, vp, vp, vp, ctypes.POINTER(PgDgGpuStats), vp]),
    "pg_np_argsort_f64": (ctypes.c_int, [vp, c_i32, vp]),
    "pg_rmat_edges": (ctypes.c_int, [c_u64, c_i32, c_u32, c_u32, c_u32, c_i64, c_i64, vp, vp, vp]),
    "pg_random_features": (ctypes.c_int, [c_u64, c_i64, c_i64, c_i32, vp, c_i64, vp]),
    "pg_timer_create": (ctypes.c_int, [ctypes.POINTER(vp)]),
    "pg_timer_destroy": (ctypes.c_int, [vp]),
    "pg_timer_start": (ctypes.c_int, [vp, vp]),
    "pg_timer_stop": (ctypes.c_int, [vp, vp]),
    "pg_timer_elapsed_ms": (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_float)]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def load():
    """dlopen the HIP library; raises PgError (never falls back) when it is missing."""
    global _lib, BOUNDS
    if _lib is None:
        path = LIB_PATH
        if os.environ.get("PG_BOUNDS") not in (None, "", "0"):
            path = BOUNDS_LIB_PATH
            if not os.path.exists(path):
                raise PgError(f"PG_BOUNDS is set but {path} is not built: `make -C pagraph_amd/csrc bounds`")
        if not os.path.exists(path):
            raise PgError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C pagraph_amd/csrc`. pagraph_amd has no CPU fallback.")
        L = ctypes.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError here = header / library mismatch
            fn.restype = res
            fn.argtypes = args
        BOUNDS = bool(L.pg_bounds_enabled())
        if os.environ.get("PG_HOST_TIMING"):
            L = _TimedLib(L)
        _lib = L
    return _lib


_K_NAMES = ("?", "k_spmm_fwd_rows(_w)", "k_compose_edge_slots", "k_sx (sample)", "k_sx (relabel)", "k_bm_rank", "k_t_keys",
            "k_t_block", "k_split / k_slots_full", "k_gather", "k_gather_labels", "k_linear_fwd<ROWS>", "k_linear_bwd_w<ROWS>",
            "k_gcn_head", "k_spmm_bwd_gather", "k_spmm_fwd(_drop)", "k_spmm_bwd", "k_scatter")


def note(t):
    """debug build: register the extent of the allocation behind tensor `t` (bounds of the indices kernels follow into it)"""
    if BOUNDS and t is not None and torch.is_tensor(t) and t.numel():
        st = t.untyped_storage()
        _lib.pg_bounds_region(ctypes.c_void_p(st.data_ptr()), st.nbytes())
    return t


def bounds_report(reset=True):
    """debug build: None when no kernel met an out-of-range index since the last reset, else a dict naming the first one
    (synchronises the device). Product build: None."""
    if not BOUNDS:
        return None
    rec = (c_u64 * 8)()
    unit = ctypes.create_string_buffer(256)
    check(_lib.pg_bounds_report(rec, unit, 256, 1 if reset else 0), "pg_bounds_report")
    if not rec[0]:
        return None
    k = int(rec[1])
    v = int(rec[3])
    return {"kernel": _K_NAMES[k] if 0 <= k < len(_K_NAMES) else str(k), "site": int(rec[2]),
            "value": v - (1 << 64) if v >= 1 << 63 else v, "bound": int(rec[4]), "block": int(rec[5]),
            "offenders": int(rec[6]), "unit": os.path.basename(unit.value.decode() or "?")}


class _TimedLib:
    """PG_HOST_TIMING=1: host time spent inside every C-ABI call (per-call mean printed at exit) — the
    replayed step is short enough for the launch thread to be the bottleneck"""

    def __init__(self, lib):
        import atexit
        import time
        self._lib, self._acc, self._clock = lib, {}, time.perf_counter
        atexit.register(self._report)

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        acc = self._acc.setdefault(name, [0, 0.0])
        clock = self._clock

        def timed(*a):
            t0 = clock()
            r = fn(*a)
            acc[0] += 1
            acc[1] += clock() - t0
            return r
        setattr(self, name, timed)
        return timed

    def _report(self):
        import sys
        for name, (n, t) in sorted(self._acc.items(), key=lambda kv: -kv[1][1]):
            if n:
                print(f"[host-timing] {name:28s} calls {n:7d}  mean {t / n * 1e6:9.1f} us  total {t:8.3f} s", file=sys.stderr)


def check(rc, what=""):
    if rc != 0:
        L = load()
        msg = L.pg_strerror(rc).decode()
        hip = L.pg_last_hip_error() if rc == -2 else 0
        raise PgError(f"{what or 'libpagraph_hip'}: {msg} (code {rc}, hip error {hip})")


def stream_ptr(stream=None):
    """hipStream_t of a torch stream (default: current stream of the current device)"""
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def del_waits_enabled():
    """PG_NO_DEL_WAIT=1 switches the finalizers' stream waits off (diagnosis: with them off, only the allocator-side
    lifetime rule below protects a pipeline that is dropped with work in flight — DESIGN section 3 'The rare illegal address')"""
    return not os.environ.get("PG_NO_DEL_WAIT")


def safe_stream_wait(stream):
    """a finalizer's best-effort wait: never inside a capture of this thread (a synchronize there is illegal and
    invalidates the capture), never raises"""
    try:
        if stream is None or not del_waits_enabled() or torch.cuda.is_current_stream_capturing():
            return
        stream.synchronize()
    except Exception:
        pass


def record_streams(root, streams, _seen=None, _depth=0):
    """tensor.record_stream(s) for every CUDA tensor reachable from `root` and every stream in `streams`.

    torch's caching allocator hands a freed block back to the stream it was ALLOCATED on at once: a buffer allocated on the
    default stream and written by kernels on a sampler / load / compute stream is recycled under those kernels' feet when its
    owner is dropped with work in flight (round 4's rare hipErrorIllegalAddress in whole-suite runs). record_stream is the
    allocator's own cure: at free time it records an event on every recorded stream and defers the block's reuse until those
    events have completed — whatever drops the owner (refcount, cyclic GC, interpreter exit) and whether or not a
    finalizer ran. Called ONCE per buffer, where the buffer is created (the cost is paid at free time only).
    Walks tensors, lists / tuples / dicts / sets, nn.Modules (parameters + buffers), optimizers (state) and plain
    pagraph_amd objects."""
    if root is None:
        return
    streams = [s for s in streams if s is not None]
    if not streams:
        return
    if _seen is None:
        _seen = set()
    if id(root) in _seen or _depth > 6:
        return
    _seen.add(id(root))
    if torch.is_tensor(root):
        if root.is_cuda:
            for st in streams:
                try:
                    root.record_stream(st)
                except RuntimeError:
                    pass              # memory the caching allocator does not own (an external / mapped allocation)
        return
    if isinstance(root, (str, bytes, int, float, bool, ctypes.Structure, ctypes.Array, torch.cuda.Stream, torch.cuda.Event)):
        return
    if isinstance(root, dict):
        for v in root.values():
            record_streams(v, streams, _seen, _depth + 1)
        return
    if isinstance(root, (list, tuple, set, frozenset)):
        for v in root:
            record_streams(v, streams, _seen, _depth + 1)
        return
    if isinstance(root, torch.nn.Module):
        for t in list(root.parameters()) + list(root.buffers()):
            record_streams(t, streams, _seen, _depth + 1)
            if t.grad is not None:
                record_streams(t.grad, streams, _seen, _depth + 1)
        return
    if isinstance(root, torch.optim.Optimizer):
        record_streams(dict(root.state), streams, _seen, _depth + 1)
    if type(root).__module__.startswith("pagraph_amd"):
        d = getattr(root, "__dict__", None)
        if d is not None:
            for k, v in list(d.items()):
                if k in ("model", "optimizer", "cacher", "sampler", "g", "store", "lib", "_lib"):
                    continue          # other owners: their creators / the trainer record them explicitly
                record_streams(v, streams, _seen, _depth + 1)
        for k in getattr(type(root), "__slots__", ()):
            record_streams(getattr(root, k, None), streams, _seen, _depth + 1)


def pipeline_stream(device, role, priority=0, name=None):
    """A stream for one role of the training pipeline: 'side' (sampler chain, load stream: a handful of small latency-bound
    launches per step) or 'compute' (the replayed step). A plain torch stream of the given priority (rounds 4-5 also had an
    experimental CU-masked variant, PG_CU_SIDE: measured, no gain, removed in round 6 — HISTORY.md)."""
    return torch.cuda.Stream(device=device, priority=priority)


def ptr(t):
    """raw pointer of a tensor (None -> NULL)"""
    if t is None:
        return ctypes.c_void_p(0)
    if BOUNDS:
        note(t)
    return ctypes.c_void_p(t.data_ptr())


def miss_list(pos, fullid, count):
    """pg_miss_list_t from three tensors / raw pointers (kept alive by the caller)"""
    as_ptr = lambda x: x if isinstance(x, (int, type(None))) else (x.value if isinstance(x, ctypes.c_void_p) else ptr(x).value)
    return PgMissList(as_ptr(pos), as_ptr(fullid), as_ptr(count))


def make_fields(items):
    """items: iterable of (cache_tensor_or_None, out_tensor, dim, cache_stride, out_stride)"""
    items = list(items)
    if len(items) > PG_MAX_FIELDS:
        raise PgError(f"at most {PG_MAX_FIELDS} fields per gather")
    arr = (PgField * max(1, len(items)))()
    for i, (cache, out, dim, cs, os_) in enumerate(items):
        if BOUNDS:
            note(cache), note(out)
        arr[i].cache = cache.data_ptr() if cache is not None else 0
        arr[i].out = out.data_ptr()
        arr[i].dim = dim
        arr[i].cache_stride = cs
        arr[i].out_stride = os_
    return arr, len(items)
