"""torch.optim.Adam for the trainer scripts (examples/profile/pa_gcn.py:137-139) as ONE HIP launch per step
(pg_adam_step, pagraph_amd/csrc/pg_optim.hip): same constructor, same arithmetic as torch's Adam
(amsgrad / maximize / foreach-only options are not offered), state_dict-compatible keys (`step`, `exp_avg`,
`exp_avg_sq`). The step counter lives on the device and is advanced by the kernel, so the optimiser can be
captured into a hipGraph as is (torch needs capturable=True and spends two launches: 7 + 15 us)."""
import ctypes

import torch

from . import _lib as L


# Pinned words the optimiser kernels write their step count to (enable_step_mirror). Never released: a replayed graph that
# outlives its optimiser object for a few microseconds (a trainer dropped without a device sync) must not write into a block
# the pinned-memory allocator has handed to somebody else. Eight bytes per optimiser that asked for a mirror.
_MIRROR_KEEPALIVE = []


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._lib = L.load()
        self._dev_state = {}      # per group: (step int64[1], ticket int32[1]) on the group's device
        # Host-side count of the step launches ENQUEUED so far for the (single) parameter group: every eager step() bumps it,
        # and whoever replays a captured graph that contains a step reports the replay (note_replayed_steps). It belongs to
        # the optimiser, not to a trainer: two trainers that share one optimiser (bench.py's reference-equivalent leg) then
        # hand out tokens from the same sequence the mirrored device counter advances through.
        self._issued = 0

    def enable_step_mirror(self, device):
        """Mirror the (single) parameter group's device step counter into pinned host memory: every later step's launch
        writes the new count there from its last block (pg_adam_step_mirror). Returns a ctypes.c_int64 living IN that
        memory — `.value` is "steps whose optimiser launch has run", readable by a launch thread without touching the HIP
        runtime — or None (several groups: no single counter). The optimiser's launch is the last of a training step, so
        a trainer can recycle a step's buffers once the count has reached that step (GraphedTrainer does, instead of
        recording an event on its compute stream)."""
        if getattr(self, "_mirror", None) is None:
            if len(self.param_groups) != 1:
                return None
            step_dev, _ = self._group_state(0, device)
            m = torch.zeros(1, dtype=torch.int64).pin_memory()
            m[0] = int(step_dev.item())                    # (synchronises once)
            if self._lib.pg_adam_step_mirror(L.ptr(step_dev), ctypes.c_void_p(m.data_ptr())) != 0:
                return None        # the library's table of mirrored counters is full (16 per process): the caller keeps its events
            self._mirror = m
            _MIRROR_KEEPALIVE.append(m)
            self._mirror_cell = ctypes.c_int64.from_address(m.data_ptr())
            self._issued = int(m[0])
        return self._mirror_cell

    def steps_issued(self):
        """step launches enqueued so far (eager steps + reported graph replays): the value the mirrored device counter
        (enable_step_mirror) reaches once all of them have run"""
        return self._issued

    def note_replayed_steps(self, n=1):
        """a captured graph holding `n` of this optimiser's step launches was replayed (a capture itself launches nothing
        and is not counted)"""
        self._issued += int(n)

    def resync_steps_issued(self, device):
        """with NOTHING of this optimiser in flight: line the host count up with the device counter (somebody replayed a
        graph with a step in it without reporting it). Synchronises. Returns the count."""
        st = self._dev_state.get(0)
        if st is not None:
            self._issued = int(st[0].item())
        return self._issued

    def __del__(self):
        try:
            if getattr(self, "_mirror", None) is not None:
                st = self._dev_state.get(0)
                if st is not None:
                    self._lib.pg_adam_step_mirror(L.ptr(st[0]), None)
                self._mirror = None
        except Exception:
            pass

    def _group_state(self, gi, device):
        st = self._dev_state.get(gi)
        if st is None:
            st = (torch.zeros(1, dtype=torch.int64, device=device), torch.zeros(1, dtype=torch.int32, device=device))
            self._dev_state[gi] = st
        return st

    def _init_state(self, ps, step_dev):
        for p in ps:
            st = self.state[p]
            if not st:
                st['step'] = step_dev          # shared by the group's tensors (torch keeps one per tensor)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise L.PgError("pagraph_amd.optim.Adam needs contiguous fp32 parameters and gradients")

    def _desc(self, group, mode, step_dev, ticket, bump):
        d = L.PgAdamDesc()
        d.mode = mode
        b1, b2 = group['betas']
        d.lr, d.beta1, d.beta2, d.eps, d.weight_decay = (float(group['lr']), float(b1), float(b2), float(group['eps']),
                                                         float(group['weight_decay']))
        d.step_dev, d.ticket_dev = L.ptr(step_dev).value, L.ptr(ticket).value
        d.bump_dev = L.ptr(bump).value if bump is not None else None
        return d

    @torch.no_grad()
    def step(self, closure=None, deferred=None, bump=None):
        """deferred: an ops.DeferredPartials registry — gradients whose per-chunk partial rows were left un-summed by
        the weight-gradient kernels are added up (pg_sum_partials' order) inside this step's single launch.
        bump: optional device int64[1] advanced by the step's (last) launch — the model's dropout step counter"""
        if deferred is not None and deferred.mixed():
            raise L.PgError("deferred partial sums: a parameter was used by a deferring dense step AND by a library "
                            "nn.Linear in the same step; its gradient would be incomplete (run without fuse_partials)")
        if deferred is not None and not deferred.conflict and (deferred.by_param or deferred.extra):
            return self._step_deferred(deferred, bump)
        if deferred is not None and deferred.conflict:
            raise L.PgError("deferred partial sums: a parameter received more than two gradient contributions in one step")
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        launched = False
        live = [(gi, group, [p for p in group['params'] if p.grad is not None]) for gi, group in enumerate(self.param_groups)]
        live = [e for e in live if e[2]]
        for li, (gi, group, ps) in enumerate(live):
            dev = ps[0].device
            if not ps[0].is_cuda:
                raise L.PgError("pagraph_amd.optim.Adam runs on the GPU only (no CPU fallback)")
            step_dev, ticket = self._group_state(gi, dev)
            self._init_state(ps, step_dev)
            for c0 in range(0, len(ps), L.PG_ADAM_MAX_TENSORS):
                chunk = ps[c0:c0 + L.PG_ADAM_MAX_TENSORS]
                last = c0 + L.PG_ADAM_MAX_TENSORS >= len(ps)
                # only the last chunk of a group advances the shared step counter (and, in the very last launch, `bump`)
                sd = step_dev if last else step_dev.clone()
                d = self._desc(group, L.PG_ADAM_FULL, sd, ticket, bump if (last and li == len(live) - 1) else None)
                d.n_tensors = len(chunk)
                for k, p in enumerate(chunk):
                    t = d.t[k]
                    t.param, t.grad = p.data_ptr(), p.grad.data_ptr()
                    t.exp_avg, t.exp_avg_sq = self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr()
                    t.numel, t.is_adam = p.numel(), 1
                with torch.cuda.device(dev):
                    L.check(self._lib.pg_adam_step(ctypes.byref(d), L.stream_ptr()), "pg_adam_step")
                launched = True
                if last and gi == 0 and not torch.cuda.is_current_stream_capturing():
                    self._issued += 1
        if bump is not None and not launched:
            bump.add_(1)           # no gradient anywhere: the counter still advances once per step
        return loss

    def _deferred_desc(self, reg, mode, bump, grads=None):
        """the descriptor of a launch that sums `reg`'s partial rows; grads: {parameter data_ptr: tensor the sum is written
        to} (default: the parameter's .grad)"""
        groups = [g for g in self.param_groups if any(p.data_ptr() in reg.by_param or p.grad is not None for p in g['params'])]
        if len(groups) != 1:
            raise L.PgError("deferred partial sums need exactly one parameter group")
        group = groups[0]
        gi = self.param_groups.index(group)
        ps = [p for p in group['params'] if p.grad is not None]
        if mode == L.PG_ADAM_REDUCE_ONLY:
            ps = [p for p in ps if p.data_ptr() in reg.by_param]
        dev = ps[0].device
        step_dev, ticket = self._group_state(gi, dev)
        self._init_state(ps, step_dev)
        n = len(ps) + len(reg.extra)
        if n > L.PG_ADAM_MAX_TENSORS:
            raise L.PgError("too many tensors for one pg_adam_step launch with deferred partial sums")
        d = self._desc(group, mode, step_dev, ticket, bump)
        d.n_tensors = n
        for k, p in enumerate(ps):
            t = d.t[k]
            g = p.grad if grads is None else grads[p.data_ptr()]
            t.param, t.grad = p.data_ptr(), g.data_ptr()
            t.exp_avg, t.exp_avg_sq = self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr()
            t.numel, t.is_adam = p.numel(), 1
            e = reg.by_param.get(p.data_ptr())
            if e is not None:
                if e[4] != p.numel():
                    raise L.PgError("deferred partial sums: registered size differs from the parameter's")
                t.partials, t.part_chunks, t.part_len, t.part_off = e[0].data_ptr(), e[1], e[2], e[3]
                e2 = reg.second.get(p.data_ptr())
                if e2 is not None:             # applied twice this step: sum(first) + sum(second)
                    if e2[4] != p.numel():
                        raise L.PgError("deferred partial sums: registered size differs from the parameter's")
                    t.partials2, t.part2_chunks, t.part2_len, t.part2_off = e2[0].data_ptr(), e2[1], e2[2], e2[3]
        for j, (dst, part, ch, rl, of) in enumerate(reg.extra):
            t = d.t[len(ps) + j]
            t.grad, t.partials, t.numel = dst.data_ptr(), part.data_ptr(), dst.numel()
            t.part_chunks, t.part_len, t.part_off, t.is_adam = ch, rl, of, 0
        return d, dev, gi

    @torch.no_grad()
    def _step_deferred(self, reg, bump=None):
        """bump: optional device int64[1] advanced by the same launch (the model's dropout step counter)"""
        d, dev, gi = self._deferred_desc(reg, L.PG_ADAM_FULL, bump)
        with torch.cuda.device(dev):
            L.check(self._lib.pg_adam_step(ctypes.byref(d), L.stream_ptr()), "pg_adam_step (deferred partial sums)")
        if gi == 0 and not torch.cuda.is_current_stream_capturing():
            self._issued += 1
        return None

    @torch.no_grad()
    def reduce_deferred(self, reg, grads):
        """The N > 1 step (GraphedTrainer): ONE launch adds `reg`'s partial rows up — pg_sum_partials' order, the same
        kernel as the fused step — and writes every parameter's summed gradient to grads[parameter data_ptr] (views of the
        flat buffer the all-reduce works on) and the reduce-only outputs (the fused head's loss) to their destinations.
        No update, no counter moves: step() behind the collective does that. Every parameter with a gradient must have
        deferred rows (the trainer checks the model's `deferrable_parameters`)."""
        if reg.mixed() or reg.conflict:
            raise L.PgError("deferred partial sums: a parameter took a non-deferring path or more than two contributions")
        missing = [p for g in self.param_groups for p in g['params'] if p.grad is not None and p.data_ptr() not in reg.by_param]
        if missing:
            raise L.PgError(f"reduce_deferred: {len(missing)} parameter(s) with a gradient have no deferred partial rows")
        d, dev, _ = self._deferred_desc(reg, L.PG_ADAM_REDUCE_ONLY, None, grads=grads)
        with torch.cuda.device(dev):
            L.check(self._lib.pg_adam_step(ctypes.byref(d), L.stream_ptr()), "pg_adam_step (reduce only)")
