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

    @torch.no_grad()
    def step(self, closure=None, deferred=None, bump=None):
        """deferred: an ops.DeferredPartials registry — gradients whose per-chunk partial rows were left un-summed by
        the weight-gradient kernels are added up (pg_sum_partials' order) inside this step's single launch"""
        if deferred is not None and deferred.mixed():
            raise L.PgError("deferred partial sums: a parameter was used by a deferring dense step AND by a library "
                            "nn.Linear in the same step; its gradient would be incomplete (run without fuse_partials)")
        if deferred is not None and not deferred.conflict and (deferred.by_param or deferred.extra):
            return self._step_deferred(deferred, bump)
        if bump is not None:
            bump.add_(1)           # nothing was deferred (library fall-backs ran): the counter still advances once per step
        if deferred is not None and deferred.conflict:
            raise L.PgError("deferred partial sums: a parameter received more than two gradient contributions in one step")
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if not ps[0].is_cuda:
                raise L.PgError("pagraph_amd.optim.Adam runs on the GPU only (no CPU fallback)")
            step_dev, ticket = self._group_state(gi, dev)
            for p in ps:
                st = self.state[p]
                if not st:
                    st['step'] = step_dev          # shared by the group's tensors (torch keeps one per tensor)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise L.PgError("pagraph_amd.optim.Adam needs contiguous fp32 parameters and gradients")
            b1, b2 = group['betas']
            for c0 in range(0, len(ps), L.PG_ADAM_MAX_TENSORS):
                chunk = ps[c0:c0 + L.PG_ADAM_MAX_TENSORS]
                n = len(chunk)
                arr = lambda xs: (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
                numel = (ctypes.c_int64 * n)(*[p.numel() for p in chunk])
                last = c0 + L.PG_ADAM_MAX_TENSORS >= len(ps)
                # only the last chunk of a group advances the shared step counter
                sd = step_dev if last else step_dev.clone()
                with torch.cuda.device(dev):
                    L.check(self._lib.pg_adam_step(n, arr(chunk), arr([p.grad for p in chunk]),
                                                   arr([self.state[p]['exp_avg'] for p in chunk]),
                                                   arr([self.state[p]['exp_avg_sq'] for p in chunk]), numel,
                                                   float(group['lr']), float(b1), float(b2), float(group['eps']),
                                                   float(group['weight_decay']), L.ptr(sd), L.ptr(ticket),
                                                   L.stream_ptr()), "pg_adam_step")
                if last and gi == 0 and not torch.cuda.is_current_stream_capturing():
                    self._issued += 1
        return loss

    @torch.no_grad()
    def _step_deferred(self, reg, bump=None):
        """bump: optional device int64[1] advanced by the same launch (the model's dropout step counter)"""
        groups = [g for g in self.param_groups if any(p.grad is not None for p in g['params'])]
        if len(groups) != 1:
            raise L.PgError("deferred partial sums need exactly one parameter group")
        group = groups[0]
        gi = self.param_groups.index(group)
        ps = [p for p in group['params'] if p.grad is not None]
        dev = ps[0].device
        step_dev, ticket = self._group_state(gi, dev)
        for p in ps:
            st = self.state[p]
            if not st:
                st['step'] = step_dev
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise L.PgError("pagraph_amd.optim.Adam needs contiguous fp32 parameters and gradients")
        n = len(ps) + len(reg.extra)
        if n > L.PG_ADAM_MAX_TENSORS:
            raise L.PgError("too many tensors for one pg_adam_step_partials launch")
        vp, i32 = ctypes.c_void_p, ctypes.c_int32
        P, G, M, V, PT = (vp * n)(), (vp * n)(), (vp * n)(), (vp * n)(), (vp * n)()
        numel = (ctypes.c_int64 * n)()
        chunks, rowlen, off, adam = (i32 * n)(), (i32 * n)(), (i32 * n)(), (i32 * n)()
        PT2, chunks2, rowlen2, off2 = (vp * n)(), (i32 * n)(), (i32 * n)(), (i32 * n)()
        for k, p in enumerate(ps):
            P[k], G[k] = p.data_ptr(), p.grad.data_ptr()
            M[k], V[k] = self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr()
            numel[k], adam[k] = p.numel(), 1
            e = reg.by_param.get(p.data_ptr())
            if e is not None:
                if e[4] != p.numel():
                    raise L.PgError("deferred partial sums: registered size differs from the parameter's")
                PT[k], chunks[k], rowlen[k], off[k] = e[0].data_ptr(), e[1], e[2], e[3]
                e2 = reg.second.get(p.data_ptr())
                if e2 is not None:             # applied twice this step: sum(first) + sum(second)
                    if e2[4] != p.numel():
                        raise L.PgError("deferred partial sums: registered size differs from the parameter's")
                    PT2[k], chunks2[k], rowlen2[k], off2[k] = e2[0].data_ptr(), e2[1], e2[2], e2[3]
        for j, (dst, part, ch, rl, of) in enumerate(reg.extra):
            k = len(ps) + j
            G[k], PT[k], numel[k] = dst.data_ptr(), part.data_ptr(), dst.numel()
            chunks[k], rowlen[k], off[k], adam[k] = ch, rl, of, 0
        b1, b2 = group['betas']
        with torch.cuda.device(dev):
            L.check(self._lib.pg_adam_step_partials2(n, P, G, M, V, numel, PT, chunks, rowlen, off, PT2, chunks2, rowlen2,
                                                     off2, adam, float(group['lr']), float(b1), float(b2),
                                                     float(group['eps']), float(group['weight_decay']), L.ptr(step_dev),
                                                     L.ptr(ticket), L.ptr(bump), L.stream_ptr()), "pg_adam_step_partials")
        if gi == 0 and not torch.cuda.is_current_stream_capturing():
            self._issued += 1
        return None
