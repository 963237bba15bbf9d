"""NeighborSampler — iterable with the call signature of
dgl.contrib.sampling.NeighborSampler as used at examples/profile/pa_gcn.py:71-76,
backed by the HIP frontier-expand sampler (pagraph_amd/csrc/pg_sample.hip).

Semantics: the build-defined spec in DESIGN.md ("Sampler spec"); the reference
never seeds DGL's sampler, so exact parity is defined against oracle/ only.
One `for nf in sampler` pass = one epoch; batch b+1 is sampled on a side stream
while the caller works on batch b (the role of DGL's prefetch=True thread).
"""
import ctypes
import time

import torch

from .. import _lib as L
from .nodeflow import NodeFlow


class _Slot:
    def __init__(self, lib, handle, hops, device, padded=False, transpose_mask=0, defer_transpose=False):
        cap_nodes = L.c_i64()
        rows = (L.c_i64 * L.PG_MAX_LAYERS)()
        edges = (L.c_i64 * L.PG_MAX_LAYERS)()
        L.check(lib.pg_sampler_capacity(handle, ctypes.byref(cap_nodes), rows, edges), "pg_sampler_capacity")
        self.cap_nodes = cap_nodes.value
        # per-layer vertex capacities, layer 0 first: layer l >= 1 is block l-1's destination side
        self.layer_caps = [self.cap_nodes - sum(rows[b] for b in range(hops))] + [rows[b] for b in range(hops)]
        self.edge_caps = [edges[b] for b in range(hops)]
        self.node_mapping = torch.empty(self.cap_nodes, dtype=torch.int64, device=device)
        self.layer_offsets = torch.empty(L.PG_MAX_LAYERS + 1, dtype=torch.int32, device=device)
        self.ip_off, self.src_off = [], []
        ip = sc = 0
        for b in range(hops):
            self.ip_off.append(ip)
            self.src_off.append(sc)
            ip += rows[b] + 1
            sc += edges[b]
        self.blk_indptr = torch.empty(ip, dtype=torch.int32, device=device)
        self.blk_src = torch.empty(max(1, sc), dtype=torch.int32, device=device)
        # source-major copies of the blocks in `transpose_mask` (gather-form backward aggregation)
        self.transpose_mask = int(transpose_mask)
        self.tp_off = []
        tp = 0
        for b in range(hops):
            self.tp_off.append(tp)
            tp += self.layer_caps[b] + 1
        self.blk_tptr = torch.zeros(tp, dtype=torch.int32, device=device) if transpose_mask else None
        self.blk_tdst = torch.zeros(max(1, sc), dtype=torch.int32, device=device) if transpose_mask else None
        # hub lists: [count, sources with more than PG_HEAVY_ROW edges in the block ...]
        self.hv_off, hv = [], 0
        for b in range(hops):
            self.hv_off.append(hv)
            hv += 1 + edges[b] // L.PG_HEAVY_ROW
        self.blk_theavy = torch.zeros(hv, dtype=torch.int32, device=device) if transpose_mask else None
        self.sizes = torch.zeros(2 * L.PG_MAX_LAYERS, dtype=torch.int32).pin_memory()
        self.sizes_dev = torch.zeros(2 * L.PG_MAX_LAYERS, dtype=torch.int32, device=device)
        self.ready = torch.cuda.Event()
        self.free = torch.cuda.Event()
        self.free_recorded = False
        self.held = False            # sampled into and not yet released by its consumer
        d = L.PgNodeflowDesc()
        d.node_mapping = self.node_mapping.data_ptr()
        d.layer_offsets = self.layer_offsets.data_ptr()
        d.blk_indptr = self.blk_indptr.data_ptr()
        d.blk_src = self.blk_src.data_ptr()
        d.sizes_pinned = self.sizes.data_ptr()
        d.sizes_dev = self.sizes_dev.data_ptr()
        d.defer_transpose = 1 if (defer_transpose and transpose_mask) else 0
        d.cap_nodes = self.cap_nodes
        d.padded = 1 if padded else 0
        for b in range(hops):
            d.blk_indptr_off[b] = self.ip_off[b]
            d.blk_src_off[b] = self.src_off[b]
            d.blk_tptr_off[b] = self.tp_off[b]
            d.blk_theavy_off[b] = self.hv_off[b]
        if transpose_mask:
            d.transpose_mask = self.transpose_mask
            d.blk_tptr = self.blk_tptr.data_ptr()
            d.blk_tdst = self.blk_tdst.data_ptr()
            d.blk_theavy = self.blk_theavy.data_ptr()
        self.desc = d


_SPIN_POLLS = 4000          # ~ a few hundred microseconds of back-to-back polling before the thread starts yielding


def _poll(event):
    """wait for `event` on the launch thread without sleeping on an interrupt (a blocking wait can wake up milliseconds
    late on a shared host) — but not at any price: past _SPIN_POLLS polls the thread yields its CPU between polls (the
    gather pool and the miss queue's worker, whose progress it is usually waiting for, run on the same quota-limited cores:
    ADVICE r03), and a wait that is badly late ends in a blocking synchronize()."""
    polls = 0
    while not event.query():
        polls += 1
        if polls > _SPIN_POLLS:
            time.sleep(0)
            if polls > 400000:        # something is badly late: stop burning the CPU
                event.synchronize()
                return


def _poll_until(pred, what, seconds=60.0):
    """the same for a host-side predicate (a word of pinned memory a kernel writes); raises after `seconds`"""
    polls = 0
    deadline = None
    while not pred():
        polls += 1
        if polls > _SPIN_POLLS:
            time.sleep(0)
            if deadline is None:
                deadline = time.monotonic() + seconds
            elif (polls & 1023) == 0 and time.monotonic() > deadline:
                raise L.PgError(what)


class NeighborSampler:
    def __init__(self, g, batch_size, expand_factor, num_hops=1, neighbor_type='in', seed_nodes=None,
                 shuffle=False, num_workers=1, prefetch=False, seed=0, copy_out=True, static=False, ring=None,
                 transpose='auto', defer_transpose=False):
        if neighbor_type != 'in':
            raise L.PgError("only neighbor_type='in' is on the hot path (pa_gcn.py:72)")
        self.lib = L.load()
        self.g = g
        self.device = g.device
        self.batch_size = int(batch_size)
        self.fanout = int(expand_factor)
        self.num_hops = int(num_hops)
        self.seed = int(seed)
        self.prefetch = bool(prefetch)
        self.copy_out = copy_out
        # static=True: fixed-shape NodeFlows (padded layers, no host sync per batch) for hipGraph replay
        self.static = bool(static)
        seeds = torch.as_tensor(seed_nodes if seed_nodes is not None else torch.arange(g.number_of_nodes()))
        seeds = seeds.to(torch.int64)
        if shuffle:
            # spec rule (1): shuffled once per sampler construction, CPU torch RNG seeded by `seed`
            gen = torch.Generator().manual_seed(self.seed)
            seeds = seeds.cpu()[torch.randperm(seeds.numel(), generator=gen)]
        self.seeds = seeds.to(self.device).contiguous()
        self.num_batches = (self.seeds.numel() + self.batch_size - 1) // self.batch_size
        self.epoch = 0
        self._ring_pos = 0
        h = L.vp()
        with torch.cuda.device(self.device):
            L.check(self.lib.pg_sampler_create(g.number_of_nodes(), L.ptr(g.indptr), L.ptr(g.indices), self.batch_size,
                                               min(self.fanout, 2 ** 31 - 1), self.num_hops, ctypes.byref(h)),
                    "pg_sampler_create")
        self.handle = h
        import os as _os
        self.stream = L.pipeline_stream(self.device, "side", int(_os.environ.get("PG_PRIO_SAMPLER", -1)), name="sampler")   # ~12 tiny latency-bound launches
        # stream on which the consumer finishes with a NodeFlow (None = current stream at hand-back);
        # a ring slot is re-sampled only after the event recorded there
        self.consumer_stream = None
        # True: the consumer calls release(nf) itself once the work that reads the NodeFlow is enqueued
        # (needed when it holds several prepared batches at once); False: released when the iterator advances
        self.manual_release = False
        # callable(ring slot index) run on the launch thread right before this stream is made to wait for a slot's
        # "free" event (GraphedTrainer: GraphCacheServer.wait_worker — the barrier must not overtake the miss copy the
        # slot's consumer is waiting for)
        self.before_slot_reuse = None
        # True: a ring slot's "free" event is polled by the thread that enqueues the next sample into it instead of being
        # waited for by the sampler's stream (GraphedTrainer sets it; needs a ring deep enough for the run-ahead)
        self.host_gated = False
        # callable(token) -> bool for slots released with a token instead of an event (see release)
        self.free_reached = None
        # transpose: blocks that also come out source-major (NodeFlow.blk_tptr / blk_tdst) so that the backward
        # aggregation is a gather. 'auto' = every block whose input can carry a gradient (all but block 0,
        # whose input is the raw feature frame); or an iterable of block indices; None = none.
        if transpose == 'auto':
            transpose = range(1, self.num_hops)
        self.transpose_mask = sum(1 << int(b) for b in (transpose or ()) if 0 <= int(b) < self.num_hops)
        # defer_transpose: the source-major copies are NOT built inside sample(); the consumer calls
        # transpose_blocks(nf, stream) on a stream ordered after the sample (GraphedTrainer: its load stream).
        # The 8 launches then leave the sampler's chain, which bounds the pipeline once misses are cheap.
        self.defer_transpose = bool(defer_transpose) and self.transpose_mask != 0
        if self.defer_transpose and not self.static:
            raise L.PgError("defer_transpose needs static=True (fixed-shape NodeFlows)")
        self.slots = [_Slot(self.lib, self.handle, self.num_hops, self.device, padded=self.static,
                            transpose_mask=self.transpose_mask, defer_transpose=self.defer_transpose)
                      for _ in range(ring if ring else (8 if self.static else 3))]
        # the slots' buffers were zero-filled on the CURRENT stream; the sampling chain writes them on self.stream, which
        # does not synchronise with it: a late fill would wipe the first samples
        torch.cuda.current_stream(self.device).synchronize()
        # Lifetimes: the ring slots, the seed list and the graph's arrays are allocated on the current stream and WRITTEN /
        # read by the sampling chain on self.stream. The allocator is told so (L.record_streams): dropped with a chain in
        # flight, their memory is not reused before the chain has finished — whatever drops them.
        L.record_streams([self.slots, self.seeds, g.indptr, g.indices], [self.stream])

    def close(self):
        """deterministic teardown: wait for the sampling chain, then release the library's handle (its own scratch is
        hipMalloc'ed: pg_sampler_destroy frees it behind a device-wide wait). Idempotent; also the context manager's exit."""
        L.safe_stream_wait(getattr(self, "stream", None))
        try:
            if getattr(self, "handle", None):
                self.lib.pg_sampler_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        # best-effort backstop only (close() is the deterministic path; the buffers' lifetimes are the allocator's business,
        # see __init__): a chain still in flight when the sampler is dropped must not outlive the handle's scratch
        try:
            self.close()
        except Exception:                # (interpreter shutdown: module globals may be gone already)
            pass

    def __len__(self):
        return self.num_batches

    def check(self):
        """raise if a look-back poll inside the sampling chain ever gave up (a NodeFlow since then was garbage).
        Synchronises with the device: the trainers call it where they check for lost miss rows, once per epoch."""
        v = L.c_i32(0)
        with torch.cuda.device(self.device):
            L.check(self.lib.pg_sampler_status(self.handle, ctypes.byref(v)), "pg_sampler_status")
        if v.value:
            raise L.PgError(f"NeighborSampler: {v.value} look-back poll(s) inside the sampling chain timed out; at least one "
                            "NodeFlow was not sampled correctly")

    def transpose_blocks(self, nf, stream):
        """build the deferred source-major block copies of `nf` (a NodeFlow of this sampler) on `stream`, which the
        caller has ordered after the sample (stream.wait_event(nf._slot.ready))"""
        if self.defer_transpose:
            with torch.cuda.device(self.device):
                L.check(self.lib.pg_sampler_transpose(self.handle, ctypes.byref(nf._slot.desc),
                                                      ctypes.c_void_p(stream.cuda_stream)), "pg_sampler_transpose")

    def _release_slot(self, slot):
        slot.free.record(self.consumer_stream or torch.cuda.current_stream(self.device))
        slot.free_token = None
        slot.free_recorded = True
        slot.held = False

    def release(self, nf, token=None):
        """manual_release mode: the ring slot behind `nf` may be re-sampled once the work enqueued so far
        on the consumer stream has finished. `token` (host-gated consumers with `free_reached` set): no event is recorded;
        the slot is free once free_reached(token) is true."""
        slot = getattr(nf, "_slot", None)
        if slot is None:
            return
        if token is not None and self.host_gated and self.free_reached is not None:
            slot.free_token = token
            slot.free_recorded = True
            slot.held = False
        else:
            self._release_slot(slot)

    def _enqueue(self, b, epoch):
        # ring position is global (not b % n): an epoch's last batch and the next epoch's first one
        # must not land in the same slot while the former is still in flight
        slot = self.slots[self._ring_pos % len(self.slots)]
        self._ring_pos += 1
        if slot.held:
            raise L.PgError("NeighborSampler ring overrun: a NodeFlow handed out earlier was never released "
                            "(manual_release consumers must call sampler.release(nf) before the ring wraps)")
        slot.held = True
        lo = b * self.batch_size
        n = min(self.batch_size, self.seeds.numel() - lo)
        if slot.free_recorded and self.host_gated and getattr(slot, "free_token", None) is not None:
            # (the consumer handed back a token instead of an event: "done" = its step counter has reached the token —
            # a word of pinned memory its last kernel writes, GraphedTrainer / optim.Adam.enable_step_mirror)
            token = slot.free_token
            _poll_until(lambda: self.free_reached(token),
                        "NeighborSampler: the consumer's step counter never reached the token of a released ring slot "
                        "(did the step that owned it run its optimiser?)")
        elif slot.free_recorded and self.host_gated:
            # the consumer of the batch that used this slot is done — checked HERE, on the launch thread, instead of with a
            # wait on the sampler's stream: an event that another stream waits for costs the stream that records it ~13 us
            # (tools/exp_graph_gap.py: 4.7 us for a record nobody waits for in-stream), and that stream is the compute
            # stream, the one that bounds the step once the features are cached. Polled, not synchronize(): a blocking
            # wait sleeps on an interrupt and, on a shared host, can wake up milliseconds late. The ring is deep enough
            # for the launch thread to stay ahead of the GPU (see GraphedTrainer).
            _poll(slot.free)
        elif getattr(slot, "n_seeds", None) is not None:
            # back-pressure: the previous sample into this ring slot (one revolution ago) has run. Nothing in the chain
            # needs it any more (the call's scalars are kernel arguments since round 3), but it bounds how far the launch
            # thread runs ahead of the GPU.
            _poll(slot.ready)
        if slot.free_recorded and not self.host_gated:
            if self.before_slot_reuse is not None:
                self.before_slot_reuse((self._ring_pos - 1) % len(self.slots))
            self.stream.wait_event(slot.free)  # the consumer of the batch that used this slot is done
        with torch.cuda.device(self.device):
            L.check(self.lib.pg_sampler_sample(self.handle, ctypes.c_void_p(self.seeds.data_ptr() + lo * 8), n,
                                               self.seed, epoch, b, ctypes.byref(slot.desc),
                                               L.stream_ptr(self.stream)), "pg_sampler_sample")
        slot.ready.record(self.stream)
        slot.n_seeds = n
        return slot

    def _finalize_static(self, slot, n_seeds):
        """no host sync: the consumer's stream waits on the sampler's event, shapes are the capacities"""
        if not self.manual_release:
            # a pipeline-managed consumer (GraphedTrainer) orders its own streams after slot.ready; making
            # the CURRENT stream wait here would also stall every stream that implicitly synchronises
            # with the legacy default stream
            torch.cuda.current_stream(self.device).wait_event(slot.ready)
        nf = getattr(slot, "static_nf", None)
        if nf is not None:
            # the fixed-shape NodeFlow of a ring slot is the same set of views every time (~30 us of tensor slicing on
            # the launch thread per minibatch): hand the same object out again with empty frames
            nf._node_frames = [None] * nf.num_layers
            nf.num_seeds = n_seeds
            return nf
        offs = [0]
        for c in slot.layer_caps:
            offs.append(offs[-1] + c)
        ips = [slot.blk_indptr[slot.ip_off[b]:slot.ip_off[b] + slot.layer_caps[b + 1] + 1] for b in range(self.num_hops)]
        srcs = [slot.blk_src[slot.src_off[b]:slot.src_off[b] + slot.edge_caps[b]] for b in range(self.num_hops)]
        nf = NodeFlow(slot.node_mapping[:offs[-1]], offs, ips, srcs)
        for b in range(self.num_hops):
            if (slot.transpose_mask >> b) & 1:
                nf.blk_tptr[b] = slot.blk_tptr[slot.tp_off[b]:slot.tp_off[b] + slot.layer_caps[b] + 1]
                nf.blk_tdst[b] = slot.blk_tdst[slot.src_off[b]:slot.src_off[b] + slot.edge_caps[b]]
                nf.blk_theavy[b] = slot.blk_theavy[slot.hv_off[b]:slot.hv_off[b] + 1 + slot.edge_caps[b] // L.PG_HEAVY_ROW]
        nf.padded = True
        nf.num_seeds = n_seeds
        nf._slot = slot
        slot.static_nf = nf
        return nf

    def _finalize(self, slot):
        if self.static:
            return self._finalize_static(slot, slot.n_seeds)
        slot.ready.synchronize()  # host needs the layer sizes (4 ints, written to pinned memory by k_pack)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(slot.ready)
        L_ = self.num_hops + 1
        sizes = slot.sizes.tolist()
        offs = [0]
        for l in range(L_):
            offs.append(offs[-1] + sizes[l])
        nm = slot.node_mapping[:offs[-1]]
        ips, srcs = [], []
        for b in range(self.num_hops):
            nd = sizes[b + 1]
            ne = sizes[L.PG_MAX_LAYERS + b]
            ips.append(slot.blk_indptr[slot.ip_off[b]:slot.ip_off[b] + nd + 1])
            srcs.append(slot.blk_src[slot.src_off[b]:slot.src_off[b] + ne])
        tps, tds, hvs = [None] * self.num_hops, [None] * self.num_hops, [None] * self.num_hops
        for b in range(self.num_hops):
            if (slot.transpose_mask >> b) & 1:
                tps[b] = slot.blk_tptr[slot.tp_off[b]:slot.tp_off[b] + sizes[b] + 1]
                tds[b] = slot.blk_tdst[slot.src_off[b]:slot.src_off[b] + sizes[L.PG_MAX_LAYERS + b]]
                hvs[b] = slot.blk_theavy[slot.hv_off[b]:slot.hv_off[b] + 1 + slot.edge_caps[b] // L.PG_HEAVY_ROW]
        if self.copy_out:
            # detach from the ring so a NodeFlow stays valid after the iterator moves on
            nm = nm.clone()
            ips = [t.clone() for t in ips]
            srcs = [t.clone() for t in srcs]
            tps = [None if t is None else t.clone() for t in tps]
            tds = [None if t is None else t.clone() for t in tds]
            hvs = [None if t is None else t.clone() for t in hvs]
        nf = NodeFlow(nm, offs, ips, srcs)
        nf.blk_tptr, nf.blk_tdst, nf.blk_theavy = tps, tds, hvs
        return nf

    def __iter__(self):
        epoch = self.epoch
        self.epoch += 1
        nb = self.num_batches
        if nb == 0:
            return
        pending = self._enqueue(0, epoch)
        slot = None
        try:
            for b in range(nb):
                slot, pending = pending, None
                if b + 1 < nb and self.prefetch:
                    pending = self._enqueue(b + 1, epoch)
                nf = self._finalize(slot)
                yield nf
                if not self.manual_release:
                    self._release_slot(slot)
                slot = None
                if b + 1 < nb and not self.prefetch:
                    pending = self._enqueue(b + 1, epoch)
        finally:
            # iterator abandoned mid-epoch (break / cycle_batches): hand back what was never consumed.
            # Anything already handed out in manual_release mode stays with its consumer.
            if pending is not None:
                pending.held = False
            if slot is not None and not self.manual_release:
                slot.held = False
