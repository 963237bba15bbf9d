"""The hot loop of examples/profile/pa_gcn.py:82-103 (and pa_gs.py) as a reusable
object, with the load/compute overlap the north star asks for:

  main stream   : model forward / backward / optimizer of batch k
  load stream   : cacher.fetch_data (gather kernel, miss path) + label lookup of batch k+1
  sampler stream: NeighborSampler's own stream, one more batch ahead

The host enqueues the compute of batch k first (asynchronous), then prepares batch
k+1 — so the only host-blocking part (the staged miss path's wait for the miss list
and its CPU row gather) runs while the GPU computes batch k.
"""
import ctypes
import time

import torch

from . import _lib as L
from . import ops


def _record_stream(nf, stream):
    ts = [nf._node_mapping.tousertensor()] + list(nf.blk_indptr) + list(nf.blk_src)
    ts += [t for t in list(nf.blk_tptr) + list(nf.blk_tdst) + list(nf.blk_theavy) if t is not None]
    for fr in nf._node_frames:
        if fr:
            ts += [v.slots if isinstance(v, ops.RowSource) else v for v in fr.values()]
    for t in ts:
        if t.is_cuda:
            t.record_stream(stream)


class Prepared:
    __slots__ = ("nf", "label", "event", "slot")


class MinibatchTrainer:
    def close(self):
        """deterministic teardown: wait for the load stream (it may still be writing this trainer's frames)"""
        L.safe_stream_wait(getattr(self, "load_stream", None))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()                 # best-effort backstop; the buffers' lifetimes are the allocator's (L.record_streams)
        except Exception:                # (interpreter shutdown: module globals may be gone already)
            pass

    def __init__(self, model, loss_fcn, optimizer, cacher, sampler, labels, device, overlap=True, need=None):
        self.need = need             # fetch_data(need=...): None = every layer and field, like the reference
        self.model, self.loss_fcn, self.optimizer = model, ops.fused_loss(loss_fcn), optimizer
        self.cacher, self.sampler, self.labels = cacher, sampler, labels
        self.device = device
        self.overlap = overlap
        self.load_stream = torch.cuda.Stream(device=device, priority=-1) if overlap else None
        self.on_step = None          # callback(step_in_epoch, loss_tensor)
        self.after_first_step = None  # callback() — pa_gcn.py:99-100 (auto_cache)
        self._first_done = False
        self._nprep = 0
        # with `need`: layers the model only aggregates stay un-materialised (model.virtual_inputs)
        self.fuse_gather = True
        # per miss-queue slot: recorded on the compute stream after the step that read the slot's staged block in place
        # (ops.RowSource). The next fetch into that slot waits for it — the worker's copy of batch k + slots would
        # otherwise overwrite rows batch k's aggregation has not read yet when the GPU runs behind the host
        self._consumed = {}
        # Lifetimes (L.record_streams): what the load stream touches of objects allocated on other streams
        L.record_streams([cacher, self.labels, getattr(sampler, "slots", None)], [self.load_stream])

    def _virtual(self, nf):
        m = getattr(self.model, 'module', self.model)
        if self.fuse_gather and self.need is not None and hasattr(m, 'virtual_inputs'):
            return m.virtual_inputs(nf.num_layers)
        return None

    # -- 'gpu-load' (pa_gcn.py:87-91) ----------------------------------------
    def prepare(self, nf):
        p = Prepared()
        p.nf = nf
        p.slot = self._nprep % self.cacher.missq_slots     # async miss path: one slot per in-flight batch
        self._nprep += 1
        ev = self._consumed.pop(p.slot, None)
        fetch_stream = self.load_stream if self.load_stream is not None else torch.cuda.current_stream(self.device)
        if ev is not None:
            # (the copy of the slot's previous job must be in its queue before a wait for something behind that job's
            # consumer goes into any stream: pg_missq_wait_idle, DESIGN "Miss path")
            self.cacher.wait_worker(p.slot)
            fetch_stream.wait_event(ev)
        if self.load_stream is None:
            with torch.autograd.profiler.record_function('gpu-load'):
                self.cacher.fetch_data(nf, need=self.need, slot=p.slot, virtual=self._virtual(nf))
                p.label = self.labels[nf.layer_parent_nid(-1)]
            p.event = None
            return p
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.load_stream):
            with torch.autograd.profiler.record_function('gpu-load'):
                self.cacher.fetch_data(nf, need=self.need, slot=p.slot, virtual=self._virtual(nf))
                p.label = self.labels[nf.layer_parent_nid(-1)]
            p.event = torch.cuda.Event()
            p.event.record(self.load_stream)
        _record_stream(nf, main)
        p.label.record_stream(main)
        return p

    # -- 'gpu-compute' (pa_gcn.py:92-97) --------------------------------------
    def compute(self, p):
        if p.event is not None:
            torch.cuda.current_stream(self.device).wait_event(p.event)
        self.cacher.wait_misses(p.slot)
        with torch.autograd.profiler.record_function('gpu-compute'):
            pred = self.model(p.nf)
            loss = self.loss_fcn(pred, p.label)
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
        if getattr(p.nf, '_fetch_plan', None) is not None and self.cacher.miss_mode == "async":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._consumed[p.slot] = ev
        return loss

    def _next(self, it):
        if self.load_stream is None:
            return next(it, None)
        with torch.cuda.stream(self.load_stream):   # the sampler's hand-off copies land on the load stream
            return next(it, None)

    def run_steps(self, it, steps=None):
        """drive `steps` minibatches from iterator `it` (None = until exhausted); returns #steps"""
        done = 0
        nf = self._next(it)
        if nf is None:
            return 0
        cur = self.prepare(nf)
        while cur is not None:
            loss = self.compute(cur)
            done += 1
            if not self._first_done:
                self._first_done = True
                if self.after_first_step is not None:
                    torch.cuda.current_stream(self.device).synchronize()
                    self.after_first_step()
            nxt = None
            if steps is None or done < steps:
                nf = self._next(it)
                if nf is not None:
                    nxt = self.prepare(nf)
            if self.on_step is not None:
                self.on_step(done, loss)
            cur = nxt
        return done

    def run_epoch(self):
        """one pass over the sampler; returns (steps, seconds) with a device sync on both sides
        (the reference times without syncing, pa_gcn.py:84,105 — the sync makes the number honest)"""
        torch.cuda.synchronize(self.device)
        t0 = time.time()
        n = self.run_steps(iter(self.sampler))
        self.cacher.drain_misses()
        torch.cuda.synchronize(self.device)
        dt = time.time() - t0
        self.cacher.check_misses()
        if hasattr(self.sampler, "check"):
            self.sampler.check()         # a look-back poll of the sampling chain that gave up = a garbage NodeFlow was trained on
        return n, dt

    def synchronize(self):
        """wait for everything enqueued so far, then raise if the miss path or the sampling chain gave up on something
        (callers of run_steps that only synchronise a stream never see those flags: ADVICE r03)"""
        self.cacher.drain_misses()
        torch.cuda.synchronize(self.device)
        self.cacher.check_misses()
        if hasattr(self.sampler, "check"):
            self.sampler.check()


def cycle_batches(sampler, steps):
    """iterator over exactly `steps` NodeFlows, wrapping into the next epoch when the
    sampler is exhausted (multi-GPU step equalisation, parallel.equalize_steps)"""
    produced = 0
    while produced < steps:
        got = False
        for nf in sampler:
            got = True
            yield nf
            produced += 1
            if produced >= steps:
                return
        if not got:
            return


import os as _os_mod


class GraphedTrainer:
    """The same loop with the whole compute step (forward, loss, backward, Adam) replayed as a
    hipGraph.  Everything between "seed ids" and "logits" has a fixed shape: the sampler emits
    padded NodeFlows (layer capacities B*k^(L-l), padding ids -1, empty CSR rows), the gather
    writes into static frames, padded seeds get label -100 (ignored by the loss).  Per step the
    host enqueues: 1 sampler call, 1 gather (+ miss scatter), 3 tiny label ops, 1 graph replay —
    and never waits for the GPU.

    One graph per sampler ring slot (the graph holds the addresses of that slot's CSR buffers
    and of its static frames; batch k computes while k+1 loads and k+2 is sampled).  Parameters
    and optimizer state are shared by the graphs; the optimizer must be capture-safe
    (torch.optim.Adam(..., capturable=True)).

    The model must not have a LIVE autograd graph from another stream when the first step is captured (e.g. the result
    of an eager `model(nf)` on the default stream that was never back-propagated nor dropped): autograd binds a
    parameter's gradient accumulator to the stream of the graph that created it, and the capture then tries to record a
    wait on that foreign stream — the runtime dies in capture_end. Evaluate under torch.no_grad() or drop the result.

    Multi-GPU (world_size > 1, no DDP wrapper): every parameter's .grad is a view of ONE flat
    buffer.  Graph A (per slot) = zero the flat buffer, forward, loss / world, backward (autograd
    accumulates straight into the views); then ONE eager all-reduce (RCCL) of the flat buffer —
    ~90 KB, the collective of pa_gcn.py:65,96 without DDP's bucket machinery — then graph B
    (shared) = optimizer.step().  Initial parameters are broadcast from rank 0 like DDP does."""

    def __init__(self, model, loss_fcn, optimizer, cacher, sampler, labels, device, warmup_eager=3, need=None,
                 process_group=None, world_size=1, keep_losses=True, lookahead=None):
        self.need = need
        # True: compute() returns a private copy of the step's loss (one more launch per step). False: it
        # returns the slot's static loss tensor, valid until that slot's graph is replayed again
        # (len(sampler.slots) steps later) — enough for a trainer that prints the loss every N steps.
        self.keep_losses = bool(keep_losses)
        self._on_main = False            # run_steps made the compute stream current for the whole loop
        self.fuse_head = True            # use model.forward_loss (fused output layer + loss) when the model has one
        # layer-0 features are aggregated straight from the cache (never materialised) when the model says it only
        # aggregates them (model.virtual_inputs) and `need` is given — SURVEY 8f-2
        self.fuse_gather = True
        # the two ordered partial sums of the weight-gradient kernels ride in the optimiser's launch (ops.defer_partials)
        # when the model says every parameter gets at most TWO gradient contributions per step (GCN: one; GraphSAGE with
        # n_layers == 1: its first NodeUpdate runs on both blocks) and the optimiser is pagraph_amd.optim.Adam on one GPU
        self.fuse_partials = True
        self.world = int(world_size)
        self.pg = process_group
        self.flat = None
        self.graph_b = None
        self.tape_b = None
        self._bump_b = None
        self._bump = None                # the dropout step counter the N > 1 optimiser launch advances (deferred step body)
        # world > 1: is the gradient all-reduce captured INSIDE the step's graph (one launch per step) or issued
        # eagerly between two graphs (three launches)? None = not probed yet; see _probe_graph_allreduce.
        # PG_GRAPH_ALLREDUCE=0 forces the eager collective.
        import os as _os
        self.allreduce_in_graph = False if _os.environ.get("PG_GRAPH_ALLREDUCE") == "0" else None
        self.tape_collectives = False   # replay a captured step that holds an all-reduce as plain launches too? (_tape_of)
        # world > 1: EVERY eager collective of this trainer runs on a communication stream of its own, ordered with the compute
        # stream by wait_stream, never on the compute stream itself. ProcessGroupNCCL's watchdog thread polls the end event of
        # each eager collective until it has completed; on ROCm an event query fails with hipErrorCapturedEvent as soon as the
        # stream the event was last recorded on is CAPTURING — also for an event recorded long before the capture began — and
        # the watchdog turns that into std::terminate. The trainer captures a step graph on the compute stream microseconds
        # after the previous step's eager all-reduce: with the collective on that stream every rank of the first real N > 1 run
        # would have died at its first capture (tools/exp_rccl_capture.py: `nosleep` aborts, `other` / `flow` run). Collectives
        # captured INSIDE a graph are fine (the process group does not hand them to the watchdog).
        # Created AFTER the pipeline's own streams (below): the runtime multiplexes a priority class's streams onto a handful of
        # hardware queues in the order they are first used, and a communication stream taken first pushes the compute stream
        # onto a queue it then shares (round 6: the N > 1 step on one GPU at 0.20 instead of 0.10 ms, tools/exp_rccl_presence.sh)
        self.comm_stream = None
        if self.world > 1:
            import torch.distributed as dist
            params = [p for p in model.parameters() if p.requires_grad]
            for p in params:
                dist.broadcast(p.data, src=0, group=self.pg)
            self.flat = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=device)
            o = 0
            self._flat_params, self._flat_views = params, {}
            for p in params:
                p.grad = self.flat[o:o + p.numel()].view_as(p)
                self._flat_views[p.data_ptr()] = p.grad
                o += p.numel()
        assert sampler.static, "GraphedTrainer needs NeighborSampler(static=True)"
        self.model, self.loss_fcn, self.optimizer = model, ops.fused_loss(loss_fcn), optimizer
        self.cacher, self.sampler, self.labels = cacher, sampler, labels
        self.device = device
        # high priority: the short HBM-bound gather should not queue behind the compute stream's GEMMs
        import os as _os
        self.load_stream = L.pipeline_stream(device, "side", int(_os.environ.get("PG_PRIO_LOAD", -1)))
        # eager warm-up, capture and replay all run on ONE non-default stream, so autograd's
        # AccumulateGrad nodes and the captured graphs agree on the stream
        self.compute_stream = L.pipeline_stream(device, "compute", int(_os.environ.get("PG_PRIO_COMPUTE", 0)))
        if self.world > 1:
            self.comm_stream = torch.cuda.Stream(device=device)
        sampler.consumer_stream = self.compute_stream   # ring slots are recycled after the graph that read them
        sampler.manual_release = True
        # the sampler's own "slot free" event (recorded on the compute stream by sampler.release right after the step)
        # already orders every later use of the slot's buffers after that step: see prepare()
        self._free_orders_slot = True
        # ... and that event is polled by the launch thread before it samples into the slot again, not waited for by the
        # sampler's stream (sampler.host_gated): the compute stream records an event no stream waits for. With the default
        # ring of 8 slots the launch thread may be 8 - (lookahead + 2) = 4 steps ahead of the GPU before it has to wait.
        sampler.host_gated = True
        # ... and when the optimiser can mirror its step counter into pinned memory (pagraph_amd.optim.Adam: written by the
        # last block of the step's last launch) there is no event at all: a slot is free once the counter has reached the
        # step that read it. tools/exp_graph_gap.py: an event record nobody waits for in-stream still costs the compute
        # stream 4.7 us per step.
        # The token of a step is the OPTIMISER's count of step launches enqueued once that step's launch is in
        # (optim.Adam.steps_issued: eager steps count themselves, replays are reported below) — one sequence per optimiser,
        # so a second trainer on the same optimiser, or steps somebody runs between two run_steps calls, cannot make this
        # trainer's tokens lag the counter they are compared with (every released slot then read as free at once).
        self._step_cell = None
        if sampler.host_gated and hasattr(optimizer, "enable_step_mirror"):
            self._step_cell = optimizer.enable_step_mirror(device)
            if self._step_cell is not None:
                cell = self._step_cell
                sampler.free_reached = lambda token: cell.value >= token
        cacher.missq_slots = len(sampler.slots)
        # ring slot i of the sampler carries the batch whose miss job sits in queue slot i: before the sampler waits for
        # "slot free" (recorded after that batch's consumer) the job's copy must be in its queue — see prepare()
        sampler.before_slot_reuse = cacher.wait_worker
        # batches prepared ahead of the one being computed. The async miss path needs 2: its worker thread
        # must have finished batch k+1 (GPU publishes the miss list -> CPU gather -> copy enqueued) by the time
        # the host wants to enqueue compute(k+1), i.e. one whole step after it was submitted.
        self.lookahead = int(lookahead) if lookahead else (2 if cacher.miss_mode == "async" else 1)
        self._lib = L.load()
        self.labels = labels.to(device, torch.int64).contiguous()
        assert self.lookahead + 2 <= len(sampler.slots)   # prepared (+1 transient) + the sampler's own prefetch
        self._prepared = []
        self.slots = {}
        self.warmup_eager = warmup_eager
        self.steps_done = 0
        self.on_step = None
        self.after_first_step = None
        self._first_done = False
        self.last_loss = None
        self._gseed = None
        # True: run_steps(it, n) keeps `lookahead` batches prepared when it returns (as long as `it` has more), so
        # the next run_steps call starts on a primed pipeline instead of paying sample -> gather -> miss-path
        # latency for its first batches again. False: every call drains what it prepared.
        # Off by default: a primed pipeline has consumed `lookahead` batches of `it` that the call did not
        # train on — right for a caller that passes the SAME iterator again (bench.py's timed windows), wrong for one that
        # makes a fresh iterator per epoch.
        self.keep_primed = False
        # Block 0's aggregation (layer-0 features -> the first hidden layer's input, dropout included) depends on no parameter:
        # it can run AHEAD of its step, on the load stream, while the previous steps' head / backward / optimiser kernels
        # (small, latency-bound, leaving the memory system idle) run on the compute stream — the replayed step then starts at
        # the dense kernel. Only when the whole table is cached (no miss rows to wait for: a wait parked on the load stream
        # would hold up the NEXT batch's miss list, and at a partial cache the step sits on its PCIe time anyway).
        # 'auto' / '1' (default): then; '0': never. PG_EARLY_AGG overrides.
        self.early_aggregate = _os.environ.get("PG_EARLY_AGG", "auto")
        # The dropout mask of an early aggregation is keyed by the value the model's step counter WILL hold when the batch is
        # computed (counted on the host from one read of the counter: pg_dropout_t.step_value), so the early path draws exactly
        # the masks the in-step path would have drawn — same losses, bit for bit, dropout on (tools/dbg_early.py). If somebody
        # runs the model in between the values drift apart, which is harmless: a mask only has to differ from step to step.
        self.early_ordinal = 0           # the dropout step value of the last early aggregation (0: none yet)
        self._early_next = None
        self.keep_gc = False             # True: leave the interpreter's cyclic garbage collector on inside run_steps
        # True: prepare / compute run inside the reference's profiler ranges 'gpu-load' / 'gpu-compute' (pa_gcn.py:87,92);
        # off by default — a record_function costs the launch thread a few microseconds per step
        self.profile_ranges = False
        # Lifetimes: everything the load / compute (/ communication) streams read or write of objects that were allocated on
        # other streams — the sampler's ring, the cacher's cache / slot map / staging, the labels, the model and the optimiser
        # state — is recorded on those streams once, here (L.record_streams): dropped with steps in flight, their memory is
        # not reused before those steps have finished, with or without a finalizer.
        L.record_streams([sampler.slots, sampler.seeds, cacher, self.labels, model, optimizer, self.flat],
                         [self.load_stream, self.compute_stream, self.comm_stream])

    class _Slot:
        pass

    def close(self):
        """deterministic teardown: wait for the load / compute / communication streams, then drop the captured step graphs
        (a graph's private pool is released with it: never under a replay still in flight). Idempotent."""
        for name in ("load_stream", "compute_stream", "comm_stream"):
            L.safe_stream_wait(getattr(self, name, None))
        if L.del_waits_enabled():
            for s_ in list(getattr(self, "slots", {}).values()):
                self._drop_graph(s_)
            self._drop_graph_b()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        # best-effort backstop (never inside a capture: L.safe_stream_wait). What keeps a dropped trainer's buffers from being
        # recycled under kernels still in flight is the allocator itself: every buffer another stream touches is recorded on
        # that stream where it is created (L.record_streams in __init__ / _make_slot / _aggregate_early)
        try:
            self.close()
        except Exception:                # (interpreter shutdown: module globals may be gone already)
            pass

    def _make_slot(self, nf):
        s = GraphedTrainer._Slot()
        R = nf._node_mapping.tousertensor().numel()
        # The initial fills run ON THE LOAD STREAM, the stream that writes these buffers first (gather, label lookup).
        # They used to run on whatever stream was current — the compute stream inside run_steps — where they queue behind
        # up to `lookahead` steps of work while the load stream runs ahead: the slot's first pg_gather_labels could land
        # BEFORE the fill that then wiped labels and count (every label ignored: a NaN loss and a zero gradient for that
        # step — the flaky test_zerocopy_refused_... of round 2, ~5 % of runs on a busy host), and a late zero fill of a
        # frame would silently wipe gathered rows.
        with torch.cuda.stream(self.load_stream):
            # wide rows whose width is not a multiple of 8 floats (Reddit's 602) get padded rows — zeros in the padding — so
            # that the MFMA dense kernel reads them in whole octets; the views handed to the model are [R, d]
            s.out = {n: torch.zeros((R, (d + 7) & ~7 if d >= 64 else d), dtype=torch.float32, device=self.device)[:, :d]
                     for n, d in self.cacher.dims.items()}
            s.label = torch.full((nf.layer_size(-1),), -100, dtype=torch.int64, device=self.device)
            # labels the loss will count (+ the label lookup's two self-cleaning scratch words, pg_gather_labels_sc)
            s.n_valid3 = torch.zeros(4, dtype=torch.int32, device=self.device)
            s.n_valid = s.n_valid3[:1]
        s.ready = torch.cuda.Event()
        s.done = torch.cuda.Event()
        s.done_recorded = False
        s.graph = None
        s.tape = None
        s.graph_synced = False
        s.graph_plan = None
        s.graph_epoch = -1
        s.nf = None
        s.loss = None
        s.plan = None
        s.slot_index = None
        s.ext_drop = None      # the model whose dropout counter this slot's (deferred) step body expects to be primed
        s.early = None         # [(block, field, RowSource)]: the aggregations of this slot's plan that run in prepare()
        s.agg0 = None          # ... and their outputs {block: [destination capacity, padded dim]}
        s.early_call = None    # ... and the cached arguments of its launch
        s.early_training = None  # ... and the model's mode (train / eval) they were aggregated in
        s.batch_plan = None      # pg_batch_plan_t of the slot's load-stream work (prepare as ONE C call), or False
        s.batch_key = None       # ... and what it was built over
        with torch.cuda.stream(self.load_stream):
            s.ready.record()     # (torch creates an event's handle at its first record: pg_batch_prepare needs it)
        # allocated on the load stream, read by the captured step on the compute stream (and by the sampler's chain when it
        # clears through the previous ids): see L.record_streams
        L.record_streams([s.out, s.label, s.n_valid3], [self.compute_stream, self.load_stream, self.sampler.stream])
        return s

    def prepare(self, nf):
        key = id(nf._slot)
        if key not in self.slots:
            self.slots[key] = self._make_slot(nf)
        s = self.slots[key]
        ls = self.load_stream
        dbg = getattr(self, "debug_events", None)
        if dbg is not None:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record(ls)
            ls.wait_event(nf._slot.ready)
            ev[1].record(ls)
        # "The graph that read this slot's buffers has finished" needs no wait of its own: the sampler stream waited for the
        # ring slot's `free` event — recorded on the compute stream right after that graph (sampler.release) — before it
        # sampled into the slot, and this stream has just waited for that sample. (Until round 3 a second event, `done`, was
        # recorded beside `free` and waited for here: one more marker between two replays on the compute stream, one more
        # barrier on this one, ~10 us of launch-thread time per step.) A sampler that does not recycle its slots that way
        # (no manual_release / another consumer stream) still gets the explicit wait.
        if s.done_recorded:
            # recorded AFTER the consumer of this slot's previous miss job (possibly a spin kernel waiting for that job's
            # copy): the worker must have enqueued that copy before this barrier goes into a queue the copy stream may
            # share (pg_missq_wait_idle). The submit below would block for the same condition anyway.
            self.cacher.wait_worker(s.slot_index)
            ls.wait_event(s.done)                # the graph that read these buffers has finished
        if dbg is not None:
            ev[2].record(ls)
        if s.batch_plan and dbg is None and not s.done_recorded and self._prepare_native(nf, s):
            if s.nf is None:
                s.nf = nf
            s.nf_cur = nf
            return s
        ls.wait_event(nf._slot.ready)            # the sampler wrote this NodeFlow on its own stream
        # no `with torch.cuda.stream(...)`, no tensor ops: everything below is a C-ABI call given the stream
        # explicitly (the launch thread is the bottleneck of a ~0.2 ms step)
        ids = nf._node_mapping.tousertensor()
        if s.plan is None or (s.plan is not False and s.plan.cache_epoch != self.cacher._cache_epoch):
            # first use of the slot, or the cache was (re)built since (auto_cache after the first step)
            s.slot_index = self.sampler.slots.index(nf._slot)
            # The plan's buffers (its slot array first of all) are FIRST WRITTEN by this stream's k_split: they must come out
            # of THIS stream's allocator pool. Allocated on whatever stream is current — the compute stream inside run_steps —
            # the allocator may hand out a block an eager warm-up step has just freed while its kernels are still running
            # (legal for a tensor that is next used on that same stream), and those kernels then write their floats over the
            # slot array the load stream has filled meanwhile: the rare hipErrorIllegalAddress of rounds 4-5, named by the
            # debug build as k_spmm_fwd_rows following float bit patterns (DESIGN section 3 'The rare illegal address').
            # (PG_PLAN_ON_CURRENT_STREAM=1 restores the old allocation for the test that demonstrates the hazard)
            if _os_mod.environ.get("PG_PLAN_ON_CURRENT_STREAM"):
                s.plan = self._plan_for(nf, s)
            else:
                with torch.cuda.stream(ls):
                    s.plan = self._plan_for(nf, s)
            s.batch_plan = None
            # ... and the compute stream READS them (the fused layer-0 aggregation follows plan.slots): recorded there, or a
            # trainer dropped with a step in flight hands the block to the next allocation on a recycled load stream
            L.record_streams(s.plan, [self.compute_stream])
        if s.plan is not False:
            self.cacher.fetch_planned(s.plan, ids, ls, slot=s.slot_index)      # (allocates under `ls` itself where it must)
        else:
            with torch.cuda.stream(ls):
                self.cacher.fetch_data(nf, out=s.out, need=self.need, slot=s.slot_index)
        if self.sampler.defer_transpose:
            self.sampler.transpose_blocks(nf, ls)     # after the gather: the miss path starts first
        if s.early is not None:
            self._aggregate_early(nf, s, ls)
        o0, o1 = nf._layer_offsets[-2], nf._layer_offsets[-1]
        sp = ctypes.c_void_p(ls.cuda_stream)
        if o1 > o0:
            L.check(self._lib.pg_gather_labels_sc(ctypes.c_void_p(ids.data_ptr() + 8 * o0), o1 - o0, L.ptr(self.labels),
                                                  self.labels.numel(), -100, L.ptr(s.label), L.ptr(s.n_valid),
                                                  ctypes.c_void_p(s.n_valid3.data_ptr() + 4), sp), "pg_gather_labels_sc")
        else:
            L.check(self._lib.pg_gather_labels(ctypes.c_void_p(ids.data_ptr() + 8 * o0), o1 - o0, L.ptr(self.labels),
                                               self.labels.numel(), -100, L.ptr(s.label), L.ptr(s.n_valid), sp),
                    "pg_gather_labels")
        s.ready.record(ls)
        if dbg is not None:
            ev[3].record(ls)
            dbg.append(ev)
        if s.batch_plan is None:
            s.batch_plan = self._build_batch_plan(nf, s)      # from the slot's next batch on: one C call (False: not this shape)
        # the static NodeFlow views of a slot are rebuilt per batch but alias the same memory:
        # keep the first one (the graph captured ITS tensors) and only refresh the frames
        if s.nf is None:
            s.nf = nf
        s.nf_cur = nf
        return s

    def _plan_for(self, nf, s):
        """fetch plan of this slot (False: the cacher's mode has no planned path)"""
        if self.cacher.miss_mode == "staged" and not self.cacher.full_cached:
            return False
        virtual = None
        if self.fuse_gather and self.need is not None and hasattr(self._bare_model(), 'virtual_inputs'):
            virtual = self._bare_model().virtual_inputs(nf.num_layers)
        plan = self.cacher.plan_fetch(nf._layer_offsets, s.out, self.need, virtual=virtual, slot=s.slot_index)
        s.early = self._early_for(plan, virtual)
        s.agg0, s.early_call = None, None
        return plan

    def _early_for(self, plan, virtual):
        """[(block, field, RowSource)] for the aggregations of this plan that are to run ahead of their step, else None"""
        mode = str(self.early_aggregate).lower()
        m = self._bare_model()
        if mode in ("0", "false", "off") or plan is False or not virtual or not hasattr(m, "early_aggregations"):
            return None
        if not self.cacher.full_cached:
            # (a forced mode for partial caches existed for a day in round 4: the launch then waits for the batch's miss rows on
            # the load stream — 0.32 instead of 0.15 ms/step; removed)
            return None
        out = []
        for blk, field, _red, _drop in m.early_aggregations(plan.num_layers, 0):
            rows = plan.row_sources.get((blk, field))
            if rows is None or not rows.aligned():
                return None            # all of the model's raw-row aggregations or none: the model skips them as a set
            out.append((blk, field, rows))
        return out or None

    def _aggregate_early(self, nf, s, ls):
        call = self._early_args(nf, s, ls)
        for args, d in call[2]:
            if d is not None:
                d.step_value = self.early_ordinal
            L.check(self._lib.pg_spmm_fwd_rows(*args), "pg_spmm_fwd_rows")
        s.early_training = self._bare_model().training

    def _early_args(self, nf, s, ls):
        """the slot's early launches (output buffers, argument tuples) and this batch's dropout step value (early_ordinal)"""
        m = self._bare_model()
        if s.agg0 is None:
            with torch.cuda.stream(ls):
                s.agg0 = {blk: torch.empty((nf.layer_size(blk + 1), (rows.dim + 7) & ~7), dtype=torch.float32,
                                           device=self.device) for blk, _f, rows in s.early}
            L.record_streams(s.agg0, [self.compute_stream])
            s.early_call = None
        if self._early_next is None or not self._prepared:
            # nothing of this trainer is prepared ahead: one read of the device counter (the pipeline is empty anyway)
            self.compute_stream.synchronize()
            with torch.cuda.stream(self.compute_stream):
                v = int(m._drop_step.item())
            self._early_next = v + (0 if m._drop_step_primed else 1) + len(self._prepared)
        self.early_ordinal = self._early_next
        self._early_next += 1
        call = s.early_call
        if call is None or call[0] is not s.early or call[1] != m.training:
            # the launches' arguments, built once per (slot, plan): the slot's static NodeFlow and frames never move; only the
            # step value changes from batch to batch (the launch thread is what bounds the step once the table is cached)
            specs = {blk: (red, drop) for blk, _f, red, drop in m.early_aggregations(nf.num_layers, self.early_ordinal)}
            launches, keep = [], []
            for blk, _field, rows in s.early:
                red, drop = specs[blk]
                rs = rows.struct()
                d = drop.struct() if drop is not None else None
                prof, ring = (rows.prof[0], rows.prof[1]) if rows.prof is not None else (None, 0)
                out = s.agg0[blk]
                args = (L.ptr(nf.blk_indptr[blk]), L.ptr(nf.blk_src[blk]), ctypes.byref(rs), int(out.size(0)), rows.dim,
                        ops._REDUCE[red], L.ptr(out), out.stride(0), ctypes.byref(d) if d is not None else None, L.ptr(prof),
                        ring, ctypes.c_void_p(ls.cuda_stream))
                launches.append((args, d))
                keep.append((rs, nf.blk_indptr[blk], nf.blk_src[blk], prof, rows))
            call = s.early_call = (s.early, m.training, launches, keep)
        return call

    # -- prepare() as one C call (round 5) --------------------------------------------------------------------------
    def _build_batch_plan(self, nf, s):
        """pg_batch_plan_t for this ring slot, or False when the slot's load-stream work is not the shape pg_batch_prepare
        covers: a table resident in HBM whose fetched rows are all read in place (the slot look-up is the whole fetch),
        deferred transposes, the self-cleaning label look-up. PG_NATIVE_PREPARE=0 keeps the call-by-call sequence."""
        import os as _os
        c, plan = self.cacher, s.plan
        if (_os.environ.get("PG_NATIVE_PREPARE", "1") == "0" or plan is None or plan is False or not plan.virtual
                or plan.dense_rows != 0 or not c.full_cached):
            return False
        o0, o1 = nf._layer_offsets[-2], nf._layer_offsets[-1]
        if o1 <= o0 or getattr(nf._slot.ready, "cuda_event", None) in (None, 0) or getattr(s.ready, "cuda_event", None) in (None, 0):
            return False
        ids = nf._node_mapping.tousertensor()
        bp = L.PgBatchPlan()
        bp.load_stream = self.load_stream.cuda_stream
        bp.ev_sampled = nf._slot.ready.cuda_event
        bp.ev_ready = s.ready.cuda_event
        bp.ids = ids.data_ptr() + 8 * plan.row_lo
        bp.rows = plan.rows
        bp.slot_map = L.ptr(c.slot_map).value
        bp.slots_out = L.ptr(plan.slots).value
        bp.stats = L.ptr(c._stats).value if c.log else None
        bp.transpose = 1 if self.sampler.defer_transpose else 0
        bp.sampler = self.sampler.handle.value if self.sampler.defer_transpose else None
        bp.desc = nf._slot.desc
        bp.n_early = 0
        bp.label_ids = ids.data_ptr() + 8 * o0
        bp.n_label_rows = o1 - o0
        bp.labels = L.ptr(self.labels).value
        bp.labels_len = self.labels.numel()
        bp.label_fill = -100
        bp.label_out = L.ptr(s.label).value
        bp.n_valid = L.ptr(s.n_valid).value
        bp.label_scratch = s.n_valid3.data_ptr() + 4
        s.batch_key = (plan, c.log, None)
        return bp

    def _prepare_native(self, nf, s):
        """prepare() of a slot that has a batch plan; False = the plan no longer fits (the caller runs the generic sequence
        and builds a new one)"""
        c, bp = self.cacher, s.batch_plan
        if s.plan is None or s.plan is False or s.plan.cache_epoch != c._cache_epoch or s.batch_key[0] is not s.plan \
                or s.batch_key[1] != c.log:
            s.batch_plan = None
            return False
        step_value = 0
        if s.early is not None:
            if len(s.early) > L.PG_MAX_LAYERS:
                # (checked BEFORE _early_args, which hands out this batch's dropout step value: the generic sequence the caller
                # falls back to asks for it again — ADVICE r05)
                s.batch_plan = False
                return False
            call = self._early_args(nf, s, self.load_stream)
            if s.batch_key[2] is not call:
                # the early launches' arguments, once per (slot, plan, model mode): the C struct's copies of what
                # _aggregate_early passes call by call
                for i, ((args, d), (blk, _f, rows)) in enumerate(zip(call[2], s.early)):
                    e = bp.early[i]
                    out = s.agg0[blk]
                    e.indptr, e.src = L.ptr(nf.blk_indptr[blk]).value, L.ptr(nf.blk_src[blk]).value
                    e.rows = rows.struct()
                    e.n_dst, e.dim, e.reduce = int(out.size(0)), rows.dim, args[5]
                    e.out, e.out_stride = L.ptr(out).value, out.stride(0)
                    e.has_drop = 1 if d is not None else 0
                    if d is not None:
                        e.drop = d
                    prof, ring = (rows.prof[0], rows.prof[1]) if rows.prof is not None else (None, 0)
                    e.prof, e.prof_ring = (L.ptr(prof).value if prof is not None else None), ring
                bp.n_early = len(call[2])
                s.batch_key = (s.batch_key[0], s.batch_key[1], call)
            step_value = self.early_ordinal
        L.check(self._lib.pg_batch_prepare(ctypes.byref(bp), step_value), "pg_batch_prepare")
        if s.early is not None:
            s.early_training = self._bare_model().training
        return True

    def _frames_for(self, s):
        rs = s.plan.row_sources if s.plan else {}
        for i in range(s.nf.num_layers):
            o0, o1 = s.nf._layer_offsets[i], s.nf._layer_offsets[i + 1]
            s.nf._node_frames[i] = {n: (rs[(i, n)] if (i, n) in rs else t[o0:o1]) for n, t in s.out.items()
                                    if self.need is None or n in self.need.get(i, ())}
        s.nf._pre_agg = None
        # rows aggregated ahead of the step carry the model's mode of THAT moment (dropout on / off): a model toggled between
        # prepare() and compute() (a validation callback) must not consume them — the step then aggregates in place
        if s.early is not None and s.agg0 is not None and s.early_training == self._bare_model().training:
            s.nf._pre_agg = {blk: s.agg0[blk][:, :rows.dim] for blk, _f, rows in s.early}

    def _bare_model(self):
        return getattr(self.model, 'module', self.model)

    def _step_body(self, s):
        self._frames_for(s)
        if self._gseed is None:                 # persistent d loss / d loss: no ones_like fill (nor a divide) per step
            self._gseed = torch.full((), 1.0 / self.world, dtype=torch.float32, device=self.device)
        loss = None
        if self.fuse_head and isinstance(self.loss_fcn, ops.CrossEntropyLoss) and hasattr(self.model, 'forward_loss'):
            # output layer + loss + their gradients in one kernel (GCN); None = not applicable
            loss = self.model.forward_loss(s.nf, s.label, s.n_valid, self._gseed, self.loss_fcn.ignore_index)
        if loss is None:
            pred = self.model(s.nf)
            loss = self.loss_fcn(pred, s.label)
        if self.world > 1:
            self.flat.zero_()
            loss.backward(self._gseed)          # loss / world: the SUM all-reduce then yields DDP's mean gradient
            if self.allreduce_in_graph:         # the collective and the optimizer belong to the same captured step
                if torch.cuda.is_current_stream_capturing():
                    import torch.distributed as dist
                    dist.all_reduce(self.flat, group=self.pg)
                else:                           # the same body run eagerly: never an eager collective on this stream
                    self._eager_all_reduce(self.flat)
                self.optimizer.step()
        else:
            loss.backward(self._gseed)
            self.optimizer.step()
        s.nf._pre_agg = None                    # consumed: an eager forward on this NodeFlow later aggregates for itself
        return loss

    # -- the captured step as plain launches (round 5; csrc/pg_tape.hip) --------------------------------------------------
    def _flat_wanted(self):
        """replay a captured step as its kernels launched one by one instead of with hipGraphLaunch: between two graph replays
        the stream idles ~12 us, between two dependent kernels of one stream ~3.4 us. N > 1 (round 6): the same — when every node
        of the captured step is a plain kernel / memset launch (pg_tape_from_graph refuses anything else) and the graph holds no
        collective (_tape_of): with the all-reduce in the graph the step is one hipGraphLaunch, with the eager all-reduce it is
        tape A, the collective, tape B. PG_FLAT_REPLAY=0 keeps the graph launch everywhere."""
        import os as _os
        return _os.environ.get("PG_FLAT_REPLAY", "1") != "0"

    def _tape_of(self, graph, holds_collective=False):
        """(tape or None, executed) for a freshly captured torch.cuda.CUDAGraph(keep_graph=True) on the CURRENT stream.
        The graph is replayed ONCE through torch (executed = True) with the default generator's Philox offset read on both
        sides: CUDAGraph.replay() refills the seed / offset tensors of every RNG kernel the capture holds and advances the
        offset (CUDAGeneratorState::replay_prologue) — a tape does neither, so a captured torch RNG kernel (nn.Dropout of a
        model with fuse_dropout=False, a user's loss) would draw the SAME numbers on every replay. A capture
        that consumed torch RNG therefore keeps hipGraphLaunch; so does one with a node that is not a plain launch."""
        if not self._flat_wanted():
            return None, False
        if holds_collective and not self.tape_collectives:
            # A captured RCCL all-reduce is a kernel node like any other to pg_tape_from_graph, and launching it again with the
            # node's arguments is what the graph launch does too — but only the graph launch is what RCCL documents and what
            # _probe_graph_allreduce has verified on THIS group, and no box of rounds 1-6 had a second GPU to try the other on:
            # the one graph of the step that holds a collective keeps hipGraphLaunch (tape_collectives = True to try).
            return None, False
        executed = False
        try:
            graph.instantiate()                       # (keep_graph=True defers it; the graph launch stays available)
            gen = torch.cuda.default_generators[self.device.index if self.device.index is not None
                                                else torch.cuda.current_device()]
            off0 = gen.get_offset()
            graph.replay()
            executed = True
            if gen.get_offset() != off0:
                return None, executed
            raw = graph.raw_cuda_graph()
            tape, nk, no = L.vp(), L.c_i32(0), L.c_i32(0)
            rc = self._lib.pg_tape_from_graph(ctypes.c_void_p(int(raw)), ctypes.byref(tape), ctypes.byref(nk), ctypes.byref(no))
            if rc == 0 and nk.value > 0:
                return tape, executed
        except Exception:
            pass
        return None, executed

    def _make_tape(self, s):
        """-> True when the slot's graph has been executed once on the way (the caller then skips its first replay)"""
        s.tape, executed = self._tape_of(s.graph, holds_collective=bool(self.world > 1 and self.allreduce_in_graph))
        return executed

    def _replay(self, s, stream):
        if getattr(s, "tape", None) is not None:
            L.check(self._lib.pg_tape_launch(s.tape, ctypes.c_void_p(stream.cuda_stream)), "pg_tape_launch")
        else:
            s.graph.replay()

    def _drop_graph(self, s):
        if getattr(s, "tape", None) is not None:
            self._lib.pg_tape_destroy(s.tape)          # (holds pointers into the graph's nodes: goes first)
            s.tape = None
        s.graph = None

    def _can_defer_partials(self):
        from .optim import Adam
        m = self._bare_model()
        return (self.fuse_partials and isinstance(self.optimizer, Adam)
                and getattr(m, "deferrable_parameters", False) and len(self.optimizer.param_groups) == 1)

    def _prime_drop_step(self):
        """before a deferred step body (eager or captured): the optimiser's launch will advance the model's dropout
        counter AFTER the step, so the counter must hold the value this step uses"""
        import os as _os
        m = self._bare_model()
        if self._can_defer_partials() and m.training and hasattr(m, "externalise_drop_step"):
            m.externalise_drop_step()

    def _step_body_deferred(self, s):
        """_step_body with the partial sums (and the dropout step counter's increment) folded into the optimiser's launch.
        One GPU: [dense, head, backward aggregation, weight gradient, sums + Adam]. N > 1 (round 6): the same kernels, then
        ONE reduce-only launch that writes every summed gradient straight into the flat buffer (no zero fill, no
        AccumulateGrad add per parameter), the all-reduce of that buffer, and the plain Adam launch, which also advances the
        dropout counter: two launches more than the one-GPU step, whatever the model."""
        m = self._bare_model()
        import contextlib
        import os as _os
        bump, scope = None, contextlib.nullcontext()
        if m.training and hasattr(m, "externalise_drop_step"):
            # the counter was primed by compute() (outside any capture); only this forward skips the model's own bump
            bump, scope = m._drop_step, m.drop_step_external()
            s.ext_drop = m
        self._bump = bump
        with scope, ops.defer_partials() as reg:
            self._frames_for(s)
            if self._gseed is None:       # persistent d loss / d loss (1 / world: the SUM all-reduce then yields DDP's mean)
                self._gseed = torch.full((), 1.0 / self.world, dtype=torch.float32, device=self.device)
            loss = None
            if self.fuse_head and isinstance(self.loss_fcn, ops.CrossEntropyLoss) and hasattr(self.model, 'forward_loss'):
                loss = self.model.forward_loss(s.nf, s.label, s.n_valid, self._gseed, self.loss_fcn.ignore_index)
            if loss is None:
                pred = self.model(s.nf)
                loss = self.loss_fcn(pred, s.label)
            loss.backward(self._gseed)
        if self.world == 1:
            self.optimizer.step(deferred=reg, bump=bump)
        else:
            self.optimizer.reduce_deferred(reg, self._flat_views)
            for p_ in self._flat_params:
                p_.grad = self._flat_views[p_.data_ptr()]          # (no kernel: the attribute only)
            if self.allreduce_in_graph:
                if torch.cuda.is_current_stream_capturing():
                    import torch.distributed as dist
                    dist.all_reduce(self.flat, group=self.pg)
                else:
                    self._eager_all_reduce(self.flat)
                self.optimizer.step(bump=bump)
            # else: compute() issues the eager all-reduce and the optimiser's launch (_sync_and_step) behind this body
        s.nf._pre_agg = None
        return loss

    def _probe_graph_allreduce(self):
        """Can this process group's all-reduce live inside a captured graph? RCCL's can (as NCCL's); gloo's cannot.
        Every rank captures a tiny all-reduce; the ranks AGREE (eager MIN) that all captures succeeded before
        anyone replays — a rank replaying a collective its peers never enqueue would hang — then the replayed
        result is checked twice. Any failure -> the eager collective between two graphs (the known-good path)."""
        import torch.distributed as dist
        if str(dist.get_backend(self.pg)).lower() != "nccl":
            return False                 # gloo & co. synchronise with the host inside the collective: not capturable
        rank = dist.get_rank(self.pg)
        t = torch.full((8,), float(rank + 1), dtype=torch.float32, device=self.device)
        flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        ok = 1
        g = torch.cuda.CUDAGraph()
        probe_stream = torch.cuda.Stream(device=self.device)     # a failed capture must not wedge the compute stream
        try:
            self._eager_all_reduce(t.clone())                     # communicator set-up happens outside the capture
            self.compute_stream.synchronize()
            with torch.cuda.stream(probe_stream):
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    dist.all_reduce(t, group=self.pg)
                except Exception:
                    ok = 0
                finally:
                    try:
                        g.capture_end()                            # always leave capture mode
                    except Exception:
                        ok = 0
        except Exception:
            ok = 0
        with torch.cuda.stream(self.compute_stream):
            flag.fill_(ok)
        self._eager_all_reduce(flag, op=dist.ReduceOp.MIN)
        self.compute_stream.synchronize()
        if int(flag.item()) == 0:
            return False
        n_ = dist.get_world_size(self.pg)         # (the group's size: a one-rank group may stand in for a larger job)
        want = float(n_ * (n_ + 1) // 2)
        good = 1
        try:
            for _ in range(2):
                with torch.cuda.stream(probe_stream):
                    t.fill_(float(rank + 1))
                    g.replay()
                probe_stream.synchronize()
                if not bool((t == want).all()):
                    good = 0
        except Exception:
            good = 0
        with torch.cuda.stream(self.compute_stream):
            flag.fill_(good)
        self._eager_all_reduce(flag, op=dist.ReduceOp.MIN)
        self.compute_stream.synchronize()
        return int(flag.item()) == 1

    def _eager_all_reduce(self, tensor, op=None):
        """an all-reduce of `tensor` ordered after and before the compute stream's work, issued on the communication stream
        (see __init__: an eager collective must never sit on a stream that is captured later)"""
        import torch.distributed as dist
        import os as _os
        cs = self.comm_stream
        cs.wait_stream(self.compute_stream)
        with torch.cuda.stream(cs):
            if op is None:
                dist.all_reduce(tensor, group=self.pg)
            else:
                dist.all_reduce(tensor, op=op, group=self.pg)
        self.compute_stream.wait_stream(cs)

    def _sync_and_step(self, capture_ok):
        """world > 1: all-reduce the flat gradient (eager), then the optimizer step (graph B, replayed as a plain launch
        when it is one)"""
        self._eager_all_reduce(self.flat)
        kw = {"bump": self._bump} if self._bump is not None else {}
        if self.graph_b is not None and self._bump_b is self._bump:
            if self.tape_b is not None:
                L.check(self._lib.pg_tape_launch(self.tape_b, ctypes.c_void_p(self.compute_stream.cuda_stream)), "pg_tape_launch")
            else:
                self.graph_b.replay()
        elif not capture_ok:
            self.optimizer.step(**kw)             # (an eager step counts itself)
            return
        else:
            self._drop_graph_b()
            g = torch.cuda.CUDAGraph(keep_graph=True) if self._flat_wanted() else torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.compute_stream, capture_error_mode="thread_local"):
                self.optimizer.step(**kw)
            self.graph_b, self._bump_b = g, self._bump
            self.tape_b, executed = self._tape_of(g)
            if not executed:
                g.replay()
        if self._step_cell is not None:
            self.optimizer.note_replayed_steps(1)

    def _drop_graph_b(self):
        if getattr(self, "tape_b", None) is not None:
            self._lib.pg_tape_destroy(self.tape_b)
            self.tape_b = None
        self.graph_b = None

    def compute(self, s):
        main = self.compute_stream
        main.wait_event(s.ready)
        self.cacher.wait_misses(s.slot_index, main)
        if s.graph is not None and s.graph_plan is not s.plan:
            # the captured step reads the cache / the slot array / the staged block of the fetch plan it was captured
            # over: a new plan (the cache changed, the miss queue was rebuilt) needs a new capture
            self._drop_graph(s)
        if s.ext_drop is not None and not s.ext_drop._drop_step_primed:
            # this slot's (captured) step expects the dropout counter to hold the NEXT value; somebody ran the model the
            # other way since (an eager forward bumps first, then uses): one eager increment puts it back
            with torch.cuda.stream(main):
                s.ext_drop.externalise_drop_step()
        counted = self._step_cell is not None
        issued0 = self.optimizer.steps_issued() if counted else 0
        if s.graph is not None and (self.world == 1 or s.graph_synced) and self._on_main:
            self._replay(s, main)                             # steady state: one launch
            if counted:
                self.optimizer.note_replayed_steps(1)
            loss = s.loss.clone() if self.keep_losses else s.loss
        else:
            with torch.cuda.stream(main):
                warm = self.steps_done < self.warmup_eager
                synced = bool(self.allreduce_in_graph)       # did the body below already all-reduce and step?
                if s.graph is not None:
                    self._replay(s, main)
                    synced = s.graph_synced
                    if counted and (self.world == 1 or synced):
                        self.optimizer.note_replayed_steps(1)
                elif warm:
                    if self.world == 1 or self._can_defer_partials():
                        self.optimizer.zero_grad(set_to_none=True)   # (deferred: autograd takes the placeholders, no add)
                    self._prime_drop_step()
                    s.loss = (self._step_body_deferred(s) if self._can_defer_partials() else self._step_body(s)).detach()
                else:
                    if self.world > 1 and self.allreduce_in_graph is None:
                        self.allreduce_in_graph = self._probe_graph_allreduce()
                    g = torch.cuda.CUDAGraph(keep_graph=True) if self._flat_wanted() else torch.cuda.CUDAGraph()
                    if self.world == 1 or self._can_defer_partials():
                        self.optimizer.zero_grad(set_to_none=True)
                    self._prime_drop_step()                      # an eager increment: must stay outside the capture
                    # thread_local: RCCL's watchdog thread may touch the runtime while this thread captures
                    with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
                        s.loss = (self._step_body_deferred(s) if self._can_defer_partials() else self._step_body(s)).detach()
                    s.graph = g
                    executed = self._make_tape(s)                # (replays once through torch when it builds a tape)
                    s.graph_epoch = self.cacher._cache_epoch
                    s.graph_plan = s.plan
                    s.graph_synced = synced = bool(self.allreduce_in_graph)
                    if not executed:
                        self._replay(s, main)                    # capture does not execute
                    if counted and (self.world == 1 or synced):
                        self.optimizer.note_replayed_steps(1)
                if self.world > 1 and not synced:
                    self._sync_and_step(capture_ok=not warm)
                # the slot's static loss tensor is overwritten when its graph is replayed again
                loss = s.loss.clone() if self.keep_losses else s.loss
        if not self._free_orders_slot:
            s.done.record(main)
            s.done_recorded = True
        # NOTE: the returned loss lives on the compute stream. No wait is queued on the caller's (default)
        # stream on purpose — a pending wait there delayed the sampler / load streams of LATER batches until
        # this step had finished (measured: the sampler started only when the current graph ended). Call
        # synchronize() (or compute_stream.synchronize()) before reading it.
        self.steps_done += 1
        # the token of this step's buffers: the optimiser's launch count with this step's launch in. A step that enqueued
        # no optimiser launch at all (every gradient None) has no "last launch" to stand for it: None = release by event.
        self._last_token = None
        if counted:
            issued1 = self.optimizer.steps_issued()
            if issued1 != issued0 + 1 and issued1 != issued0:
                raise L.PgError(f"GraphedTrainer: one step advanced the optimiser's launch count by {issued1 - issued0}")
            self._last_token = issued1 if issued1 == issued0 + 1 else None
        self.last_loss = loss
        return loss

    def synchronize(self):
        """wait for everything enqueued so far: first (on the host, without touching the runtime) for the miss
        queue's worker to enqueue its outstanding copies, then for the device"""
        self.cacher.drain_misses()
        self.compute_stream.synchronize()
        self.cacher.check_misses()       # a device-side wait that gave up means a step trained on rows that never landed
        self.sampler.check()             # ... a look-back poll of the sampling chain that gave up, on a garbage NodeFlow

    def run_steps(self, it, steps=None):
        # the compute stream is made current for the whole loop (a graph replays on the current stream;
        # entering a stream context per step costs more launch-thread time than the replay itself)
        prev = torch.cuda.current_stream(self.device)
        torch.cuda.set_stream(self.compute_stream)
        self._on_main = True
        # the launch thread has ~0.3 ms of slack (two prepared batches): a cyclic-GC pass of the interpreter (10 ms with
        # torch's object graph) drains the pipeline — seen as an isolated 20-step window at 0.5-0.7 ms/step
        import gc
        gc_on = gc.isenabled() and not self.keep_gc
        if gc_on:
            gc.disable()
        try:
            return self._run_steps(it, steps)
        finally:
            if gc_on:
                gc.enable()
            self._on_main = False
            torch.cuda.set_stream(prev)

    def _run_steps(self, it, steps=None):
        done = 0
        if self._step_cell is not None and not self._prepared:
            # nothing of this trainer is in flight: the optimiser's host count and its device counter must agree once the
            # device is idle — unless somebody replayed a step of this optimiser without reporting it (their own captured
            # graph): then the device is right
            self.compute_stream.synchronize()
            if int(self._step_cell.value) != self.optimizer.steps_issued():
                torch.cuda.synchronize(self.device)
                self.optimizer.resync_steps_issued(self.device)

        ranges = self.profile_ranges

        def prepare_one():
            nf = next(it, None)
            if nf is None:
                return False
            if ranges:
                with torch.autograd.profiler.record_function('gpu-load'):
                    self._prepared.append(self.prepare(nf))
            else:
                self._prepared.append(self.prepare(nf))
            return True

        while len(self._prepared) < self.lookahead and prepare_one():
            pass
        # PG_TRACE_LAUNCH=1 (diagnosis): per iteration, how long the launch thread spent in sample + prepare / in compute /
        # in release — the three phases a host stall can hide in (GraphedTrainer.launch_trace: [ms, ms, ms] per step)
        trace = self.launch_trace if getattr(self, "launch_trace", None) is not None else None
        if trace is None and __import__("os").environ.get("PG_TRACE_LAUNCH"):
            trace = self.launch_trace = []
        while self._prepared and (steps is None or done < steps):
            # top the pipeline up BEFORE (possibly) blocking on the oldest batch's miss rows: the sampler
            # and the load stream of later batches must never wait for the host
            t_0 = time.perf_counter() if trace is not None else 0.0
            if steps is None or self.keep_primed or done + len(self._prepared) < steps:
                prepare_one()
            t_1 = time.perf_counter() if trace is not None else 0.0
            cur = self._prepared.pop(0)
            if ranges:
                with torch.autograd.profiler.record_function('gpu-compute'):
                    loss = self.compute(cur)
            else:
                loss = self.compute(cur)
            t_2 = time.perf_counter() if trace is not None else 0.0
            self.sampler.release(cur.nf_cur, token=self._last_token)
            if trace is not None:
                trace.append(((t_1 - t_0) * 1e3, (t_2 - t_1) * 1e3, (time.perf_counter() - t_2) * 1e3))
            done += 1
            if not self._first_done:
                self._first_done = True
                if self.after_first_step is not None:
                    torch.cuda.synchronize(self.device)
                    self.after_first_step()
            if self.on_step is not None:
                self.on_step(done, loss)
        return done
