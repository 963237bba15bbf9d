"""autograd wrappers over the HIP aggregation kernels (pagraph_amd/csrc/pg_spmm.hip)."""
import ctypes
import os

import torch

from . import _lib as L

_REDUCE = {"mean": L.PG_REDUCE_MEAN, "sum": L.PG_REDUCE_SUM, "max": L.PG_REDUCE_MAX}


class DropoutSpec:
    """nn.Dropout folded into the aggregation that consumes its output (pg_dropout_t, include/pagraph_hip.h):
    p quantised to threshold / 65536, counter-based mask keyed by (seed, tag, *step) — `step` is a device int64
    tensor the owner bumps once per forward, so a replayed hipGraph draws a fresh mask every step."""
    __slots__ = ("threshold", "seed", "tag", "step", "step_value")

    def __init__(self, p, seed, tag, step, step_value=0):
        self.threshold = min(65535, int(round(float(p) * 65536.0)))
        self.seed, self.tag, self.step = int(seed) & 0xFFFFFFFFFFFFFFFF, int(tag) & 0xFFFFFFFF, step
        self.step_value = int(step_value)      # used when step is None: the caller keeps the count on the host

    def struct(self):
        return L.PgDropout(self.threshold, self.tag, self.seed, L.ptr(self.step), self.step_value)

    @staticmethod
    def fusable(h):
        if isinstance(h, RowSource):
            return h.aligned()
        return (h.is_cuda and h.dtype == torch.float32 and h.dim() == 2 and h.size(1) % 4 == 0 and h.stride(1) == 1
                and h.stride(0) % 4 == 0 and h.data_ptr() % 16 == 0)


class RowSource:
    """Rows of a NodeFlow layer that were never materialised (pg_row_source_t): row p lives in the HBM feature cache
    (slots[p] >= 0) or in the block of miss rows the miss path copied to the device (slots[p] <= -3). What
    GraphCacheServer hands the model in place of a dense [rows, dim] frame when the model only aggregates the field
    (SURVEY 8f-2); block_aggregate consumes it with pg_spmm_fwd_rows. Not differentiable (raw features)."""

    def __init__(self, slots, cache, staged_ptr, staged_stride, dim, keep=(), prof=None):
        self.slots = slots                    # device int32 [rows]
        self.cache = cache                    # device fp32 [cached_rows, dim] column view of the fused cache, or None
        self.staged_ptr = int(staged_ptr or 0)
        self.staged_stride = int(staged_stride)
        self.dim = int(dim)
        self.keep = keep                      # whatever owns the staged block
        self.prof = prof                      # (device uint64 [PG_PROF_WORDS * ring], ring[, marker kernel?]) or None: self-timing
        self.is_cuda, self.dtype, self.device = True, torch.float32, slots.device
        self.requires_grad = False

    @property
    def shape(self):
        return (self.slots.numel(), self.dim)

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def dim_(self):
        return 2

    def aligned(self):
        """pg_spmm_fwd_rows' envelope: every row source is padded to whole 16-byte pieces (dim % 4 != 0 — Reddit's 602 —
        is fine as long as the fused cache row / the staged block have room for the last piece)"""
        d4 = (self.dim + 3) & ~3
        ok = self.dim >= 256
        if self.cache is not None:
            ok = ok and self.cache.stride(0) % 4 == 0 and self.cache.stride(0) >= d4 and self.cache.data_ptr() % 16 == 0
        if self.staged_ptr:
            ok = ok and self.staged_stride % 4 == 0 and self.staged_stride >= d4 and self.staged_ptr % 16 == 0
        return ok

    def struct(self):
        if L.BOUNDS:
            L.note(self.slots), L.note(self.cache)
        return L.PgRowSource(self.slots.data_ptr(), self.cache.data_ptr() if self.cache is not None else 0,
                             self.staged_ptr, self.cache.stride(0) if self.cache is not None else self.dim,
                             self.staged_stride)


def aggregate_rows(indptr, src, rows, n_dst, reduce="mean", dropout=None):
    """block_aggregate for a RowSource: gather + (dropout) + aggregate in one kernel, no [rows, dim] frame"""
    lib = L.load()
    # rows padded to a multiple of 8 floats: whole 16-byte pieces for this kernel, whole octets for ops.linear's MFMA
    # kernel behind it (the padding columns of a ragged dim — 602 — are written as zeros); the caller sees [n_dst, dim]
    pad = (rows.dim + 7) & ~7
    out = torch.empty((int(n_dst), pad), dtype=torch.float32, device=rows.device)[:, :rows.dim]
    rs = rows.struct()
    d = dropout.struct() if dropout is not None else None
    prof, ring, marker = (tuple(rows.prof) + (None,))[:3] if rows.prof is not None else (None, 0, None)
    with torch.cuda.device(rows.device):
        L.check(lib.pg_spmm_fwd_rows(L.ptr(indptr), L.ptr(src), ctypes.byref(rs), int(n_dst), rows.dim, _REDUCE[reduce],
                                     L.ptr(out), out.stride(0), ctypes.byref(d) if d is not None else None,
                                     L.ptr(prof), ring, L.stream_ptr()), "pg_spmm_fwd_rows")
        if marker:                   # profiling runs only: a one-thread marker kernel right behind it (tools/join_stamps_trace.py)
            L.check(lib.pg_prof_stamp(L.ptr(prof), ring, L.ptr(dropout.step) if dropout is not None else None,
                                      L.stream_ptr()), "pg_prof_stamp")
    return out


def spmm_bwd_call(lib, grad_out, grad_h, n_src, reduce, indptr=None, src=None, transposed=None, drop_struct=None, h=None, out=None,
                  act_out=None, dz=None, stream=None):
    """pg_spmm_bwd through its descriptor (pg_spmm_bwd_desc_t): gather form when `transposed` = (tptr, tdst, heavy) is given,
    else the scatter form over (indptr, src); reduce 'max' needs the forward's input h and output out. Returns the C code."""
    d = L.PgSpmmBwdDesc()
    d.indptr, d.src = L.ptr(indptr).value, L.ptr(src).value
    if transposed is not None:
        tptr, tdst, heavy = transposed
        d.tptr, d.tdst, d.heavy = L.ptr(tptr).value, L.ptr(tdst).value, L.ptr(heavy).value
        d.heavy_cap = heavy.numel() - 1 if heavy is not None else 0
    d.grad_out, d.go_stride = L.ptr(grad_out).value, grad_out.stride(0)
    d.grad_h, d.gh_stride = L.ptr(grad_h).value, (grad_h.stride(0) if grad_h is not None else grad_out.size(1))
    d.n_dst, d.n_src, d.dim = grad_out.size(0), int(n_src), grad_out.size(1)
    d.reduce = reduce if isinstance(reduce, int) else _REDUCE[reduce]
    if h is not None:
        d.h, d.h_stride = L.ptr(h).value, h.stride(0)
    if out is not None:
        d.out, d.out_stride = L.ptr(out).value, out.stride(0)
    if act_out is not None:
        d.act_out, d.act_stride = L.ptr(act_out).value, act_out.stride(0)
    d.dz = L.ptr(dz).value
    if drop_struct is not None:
        d.has_drop, d.drop = 1, drop_struct
    return lib.pg_spmm_bwd(ctypes.byref(d), stream if stream is not None else L.stream_ptr())


class _BlockAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, indptr, src, h, n_dst, reduce, drop, tptr, tdst, heavy, dz_n=0):
        lib = L.load()
        h = h.contiguous()
        out = torch.empty((n_dst, h.size(1)), dtype=torch.float32, device=h.device)
        with torch.cuda.device(h.device):
            if drop is None:
                L.check(lib.pg_spmm_fwd(L.ptr(indptr), L.ptr(src), L.ptr(h), h.stride(0), n_dst, h.size(1),
                                        _REDUCE[reduce], L.ptr(out), out.stride(0), L.stream_ptr()), "pg_spmm_fwd")
            else:
                d = drop.struct()
                L.check(lib.pg_spmm_fwd_drop(L.ptr(indptr), L.ptr(src), L.ptr(h), h.stride(0), n_dst, h.size(1),
                                             _REDUCE[reduce], L.ptr(out), out.stride(0), ctypes.byref(d),
                                             L.stream_ptr()), "pg_spmm_fwd_drop")
        ctx.dz_n = 0
        ctx.n_src, ctx.reduce, ctx.drop = h.size(0), reduce, drop
        use_t = tptr is not None and tptr.numel() == h.size(0) + 1
        if reduce == "max":      # its backward compares every message with the maximum: keeps input and output
            ctx.use_t = use_t
            ctx.dz_n = dz_n if use_t and _dz_fusable(h, dz_n) else 0
            ctx.save_for_backward(indptr, src, h, out, *((tptr, tdst, heavy) if use_t else ()))
            return out
        if use_t:
            if _dz_fusable(h, dz_n):
                ctx.dz_n = dz_n
                ctx.save_for_backward(indptr, src, tptr, tdst, heavy, h)
            else:
                ctx.save_for_backward(indptr, src, tptr, tdst, heavy)
        else:
            ctx.save_for_backward(indptr, src)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.needs_input_grad[2]:
            return (None,) * 10
        lib = L.load()
        go = grad_out.contiguous()
        if ctx.reduce == "max":
            return (None, None, _BlockAggregate._backward_max(ctx, lib, go)) + (None,) * 7
        if len(ctx.saved_tensors) >= 5:          # gather form over the block's source-major copy
            indptr, src, tptr, tdst, heavy = ctx.saved_tensors[:5]
            gh = torch.empty((ctx.n_src, go.size(1)), dtype=torch.float32, device=go.device)
            d = ctx.drop.struct() if ctx.drop is not None else None
            y = ctx.saved_tensors[5] if ctx.dz_n else None
            dz = torch.empty((ctx.n_src, ctx.dz_n), dtype=torch.float32, device=go.device) if ctx.dz_n else None
            with torch.cuda.device(go.device):
                L.check(spmm_bwd_call(lib, go, gh, ctx.n_src, ctx.reduce, indptr=indptr, transposed=(tptr, tdst, heavy), drop_struct=d,
                                      act_out=y, dz=dz), "pg_spmm_bwd (gather form)")
            if dz is not None:
                _stash_dz(gh, dz)
            return (None, None, gh) + (None,) * 7
        indptr, src = ctx.saved_tensors
        gh = torch.zeros((ctx.n_src, go.size(1)), dtype=torch.float32, device=go.device)
        with torch.cuda.device(go.device):
            L.check(spmm_bwd_call(lib, go, gh, ctx.n_src, ctx.reduce, indptr=indptr, src=src,
                                  drop_struct=ctx.drop.struct() if ctx.drop is not None else None), "pg_spmm_bwd (scatter form)")
        return (None, None, gh) + (None,) * 7


    @staticmethod
    def _backward_max(ctx, lib, go):
        """pg_spmm_bwd (max, gather form) over the source-major copy when the sampler built one, else the scatter form"""
        indptr, src, h, out = ctx.saved_tensors[:4]
        d = ctx.drop.struct() if ctx.drop is not None else None
        with torch.cuda.device(go.device):
            if ctx.use_t:
                tptr, tdst, heavy = ctx.saved_tensors[4:7]
                gh = torch.empty((ctx.n_src, go.size(1)), dtype=torch.float32, device=go.device)
                dz = torch.empty((ctx.n_src, ctx.dz_n), dtype=torch.float32, device=go.device) if ctx.dz_n else None
                L.check(spmm_bwd_call(lib, go, gh, ctx.n_src, "max", transposed=(tptr, tdst, heavy), drop_struct=d, h=h, out=out,
                                      dz=dz), "pg_spmm_bwd (max, gather form)")
                if dz is not None:
                    _stash_dz(gh, dz)
                return gh
            gh = torch.zeros((ctx.n_src, go.size(1)), dtype=torch.float32, device=go.device)
            L.check(spmm_bwd_call(lib, go, gh, ctx.n_src, "max", indptr=indptr, src=src, drop_struct=d, h=h, out=out),
                    "pg_spmm_bwd (max, scatter form)")
        return gh


def block_aggregate(indptr, src, h, n_dst, reduce="mean", dropout=None, transpose=None):
    """out[v] = reduce_{e in block, dst(e)=v} dropout(h)[src(e)]  (DGL copy_src + mean|sum|max; `dropout` is a
    DropoutSpec or None; `transpose` = (tptr, tdst[, heavy]), the block's source-major copy (and hub list) from
    the sampler, lets the backward run as a gather instead of fp32 atomics)"""
    if isinstance(h, RowSource):
        if dropout is not None and dropout.threshold == 0:
            dropout = None
        return aggregate_rows(indptr, src, h, n_dst, reduce, dropout)
    if h.dtype != torch.float32 or not h.is_cuda:
        raise L.PgError("block_aggregate needs fp32 CUDA tensors (no CPU fallback)")
    if dropout is not None and (dropout.threshold == 0 or not DropoutSpec.fusable(h)):
        if dropout.threshold:
            raise L.PgError("fused dropout needs a row-aligned fp32 input with dim % 4 == 0 (check DropoutSpec.fusable)")
        dropout = None
    tptr, tdst, heavy = (tuple(transpose) + (None,))[:3] if transpose is not None else (None, None, None)
    return _BlockAggregate.apply(indptr, src, h, int(n_dst), reduce, dropout, tptr, tdst, heavy,
                                 int(getattr(h, '_pg_concat_n', 0)))


ACT_NONE, ACT_RELU, ACT_CONCAT = 0, 1, 2

# dZ side channel: when the rows an aggregation consumed were a skip-concat NodeUpdate's output y = [z | relu(z)]
# (ops.linear tags such a y with `_pg_concat_n`), the aggregation's backward (pg_spmm_bwd (gather form + dZ)) writes
# dZ = g[:, :N] + g[:, N:] * (z > 0) next to its grad_h and parks it here; the NodeUpdate's backward, called by autograd with
# that very grad_h, picks it up and skips pg_linear_bwd_w's own dZ launch. The entry holds grad_h itself, so no other live
# tensor can have its address; a summed gradient (y consumed twice) is a different tensor and simply misses.
_DZ_STASH = []


def _stash_dz(gh, dz):
    del _DZ_STASH[:-3]
    _DZ_STASH.append((gh, dz))


def _take_dz(gy, rows, N):
    for i, (gh, dz) in enumerate(_DZ_STASH):
        if gh.data_ptr() == gy.data_ptr() and gh.shape == gy.shape and dz.shape == (rows, N):
            del _DZ_STASH[i]
            return dz
    return None


FUSE_DZ = True          # (a module attribute: the tests that compare the fused dZ with k_dz flip it)


def _dz_fusable(h, dz_n):
    """pg_spmm_bwd (gather form + dZ)'s envelope: [rows, 2 N] fp32 rows of 16-byte pieces, a whole row inside one lane group"""
    return FUSE_DZ and bool(dz_n) and h.size(1) == 2 * dz_n and h.size(1) % 8 == 0 and h.size(1) <= 256 and h.stride(0) % 4 == 0 \
        and h.data_ptr() % 16 == 0


class DeferredPartials:
    """While one of these is active (`with ops.defer_partials() as reg:`), the weight-gradient kernels leave their
    per-chunk partial rows un-summed and register them here, keyed by the parameter's storage; the optimiser
    (pagraph_amd.optim.Adam.step(deferred=reg)) adds them up — in pg_sum_partials' exact order — inside its own single
    launch (pg_adam_step). Two k_sum_partials launches of the replayed GCN step disappear. Only valid when
    every parameter receives exactly ONE gradient contribution per step and nothing reads the gradients (or the fused
    head's loss value) before the optimiser has run."""

    def __init__(self):
        self.by_param = {}     # parameter data_ptr -> (partials tensor, chunks, row length, offset, numel)
        self.second = {}       # parameter data_ptr -> the same for a parameter's SECOND contribution of the step
        self.extra = []        # reduce-only outputs: (destination tensor, partials, chunks, row length, offset)
        self.conflict = False
        self.plain = set()     # parameters that ALSO went through a non-deferring path this step (library nn.Linear)

    def saw_plain(self, *params):
        """a dense step that does not defer (the library fall-back of ops.linear / linear2) used these parameters: autograd
        accumulates their gradient into p.grad, which the deferred optimiser launch would overwrite"""
        for p in params:
            if p is not None:
                self.plain.add(p.data_ptr())

    def mixed(self):
        """True when some parameter has both kinds of use in one step (its deferred sum would drop the other part)"""
        return any(k in self.plain for k in self.by_param)

    def add(self, param, part, chunks, rowlen, off):
        """register one contribution; returns True for a parameter's first contribution of the step — the caller hands
        autograd its (still unsummed) gradient buffer only then and None for a later one, so that AccumulateGrad has
        nothing to add: the optimiser's launch forms sum(first) + sum(second) (pg_adam_step)"""
        e = (part, int(chunks), int(rowlen), int(off), param.numel())
        k = param.data_ptr()
        if k not in self.by_param:
            self.by_param[k] = e
            return True
        if k in self.second:
            self.conflict = True            # a third contribution to the same parameter: cannot be deferred
        self.second[k] = e
        return False


_DEFER = None


class defer_partials:
    def __enter__(self):
        global _DEFER
        self.prev = _DEFER
        _DEFER = DeferredPartials()
        return _DEFER

    def __exit__(self, *exc):
        global _DEFER
        _DEFER = self.prev
        return False


def _apply_act(z, act):
    if act == ACT_RELU:
        return torch.relu(z)
    if act == ACT_CONCAT:
        return torch.cat((z, torch.relu(z)), dim=1)
    return z


def linear_fwd_call(lib, x1, w1, b1, y, n, N, act, x2=None, w2=None, b2=None, stream=None):
    """pg_linear_fwd through its descriptor (include/pagraph_hip.h pg_linear_fwd_desc_t): x1 a [n, K1] tensor or an
    ops.RowSource (rows read in place); optional second operand pair. Returns the C return code."""
    d = L.PgLinearFwdDesc()
    keep = None
    if isinstance(x1, RowSource):
        keep = x1.struct()
        d.X1rows = ctypes.addressof(keep)
        d.K1 = w1.size(1)
    else:
        d.X1, d.x1_stride, d.K1 = L.ptr(x1).value, x1.stride(0), w1.size(1)
    d.W1, d.bias1 = L.ptr(w1).value, L.ptr(b1).value
    if x2 is not None:
        d.X2, d.x2_stride, d.K2 = L.ptr(x2).value, x2.stride(0), w2.size(1)
        d.W2, d.bias2 = L.ptr(w2).value, L.ptr(b2).value
    d.Y, d.y_stride, d.n, d.N, d.act = L.ptr(y).value, y.stride(0), int(n), int(N), int(act)
    return lib.pg_linear_fwd(ctypes.byref(d), stream if stream is not None else L.stream_ptr())


def linear_bwd_call(lib, g, x1, K1, N, dW1, db1, part1, sum_partials, y=None, act=0, dz=None, x2=None, K2=0, dW2=None, db2=None,
                    part2=None, stream=None):
    """pg_linear_bwd_w through its descriptor (pg_linear_bwd_desc_t): one weight gradient, or with x2 GraphSAGE's two over the
    same dZ in one launch; x1 a tensor or an ops.RowSource. Returns the C return code."""
    d = L.PgLinearBwdDesc()
    keep = None
    if isinstance(x1, RowSource):
        keep = x1.struct()
        d.X1rows = ctypes.addressof(keep)
    else:
        d.X1, d.x1_stride = L.ptr(x1).value, x1.stride(0)
    d.dY, d.dy_stride, d.K1, d.N, d.n = L.ptr(g).value, g.stride(0), int(K1), int(N), x1.size(0)
    if x2 is not None:
        d.X2, d.x2_stride, d.K2 = L.ptr(x2).value, x2.stride(0), int(K2)
        d.dW2, d.db2, d.partials2 = L.ptr(dW2).value, L.ptr(db2).value, L.ptr(part2).value
    d.Yout, d.yo_stride, d.act = L.ptr(y).value, (y.stride(0) if y is not None else 0), int(act)
    d.dW1, d.db1, d.dz_scratch, d.partials1 = L.ptr(dW1).value, L.ptr(db1).value, L.ptr(dz).value, L.ptr(part1).value
    d.sum_partials = int(sum_partials)
    return lib.pg_linear_bwd_w(ctypes.byref(d), stream if stream is not None else L.stream_ptr())


class _SkinnyLinear(torch.autograd.Function):
    """NodeUpdate's dense step y = act(x @ W.T + b) with the tall-skinny pieces on the fp32-MFMA kernels
    of pg_dense.hip: forward (bias + activation / skip-concat fused in the epilogue) when out_features <= 64
    and K % 8 == 0; weight / bias gradient (a reduction over all rows into a tiny [N, K] matrix, the
    activation's derivative applied on the fly) always."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        lib = L.load()
        n, K = x.shape
        N = weight.size(0)
        if N <= 64 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and weight.is_contiguous():
            y = torch.empty((n, 2 * N if act == ACT_CONCAT else N), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                L.check(linear_fwd_call(lib, x, weight, bias, y, n, N, act), "pg_linear_fwd")
        else:
            y = _apply_act(torch.nn.functional.linear(x, weight, bias), act)
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        ctx.has_bias = bias is not None
        ctx.bias_ref = bias
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        return _skinny_backward(ctx, gy, x, weight, y, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                ctx.needs_input_grad[2]) + (None,)


def _skinny_backward(ctx, gy, x, weight, y, need_x, need_w, need_b):
    """(gx, gw, gb) of y = act(x @ W.T + b): weight / bias gradient on pg_linear_bwd_w (dZ derived on the fly, or taken from
    the consumer's backward when it left one: _DZ_STASH), gx = dZ @ W through the library"""
    lib = L.load()
    gy = gy.contiguous()
    act = ctx.act
    gx = gw = gb = None
    N, K = weight.shape
    gz = gy if act == ACT_NONE else None
    if need_w or (ctx.has_bias and need_b):
        buf = torch.empty(N * K + N, dtype=torch.float32, device=x.device)
        part = torch.empty(lib.pg_linear_bwd_w_scratch(x.size(0), K, N), dtype=torch.float32, device=x.device)
        gw = buf[:N * K].view(N, K)
        gb = buf[N * K:] if ctx.has_bias else None
        defer = _DEFER is not None and ctx.has_bias
        ready = _take_dz(gy, x.size(0), N) if act == ACT_CONCAT else None
        with torch.cuda.device(x.device):
            if ready is not None:        # dZ came with the gradient (pg_spmm_bwd (gather form + dZ)): plain dY = dZ, no act
                dz = ready
                L.check(linear_bwd_call(lib, dz, x, K, N, gw, gb, part, 0 if defer else 1), "pg_linear_bwd_w")
            else:
                dz = torch.empty((x.size(0), N), dtype=torch.float32, device=x.device) if act != ACT_NONE else None
                L.check(linear_bwd_call(lib, gy, x, K, N, gw, gb, part, 0 if defer else 1, y=y, act=act, dz=dz),
                        "pg_linear_bwd_w")
        if defer:
            rowlen = N * K + N
            first_w = _DEFER.add(weight, part, part.numel() // rowlen, rowlen, 0)
            first_b = _DEFER.add(ctx.bias_ref, part, part.numel() // rowlen, rowlen, N * K)
            gw = gw if first_w else None
            gb = gb if first_b else None
        if dz is not None:
            gz = dz
    if need_x:
        if gz is None:
            gz = gy * (y > 0) if act == ACT_RELU else gy[:, :N] + gy[:, N:] * (y[:, :N] > 0)
        gx = gz @ weight
    return gx, gw, gb


def _bwd_w(lib, g, x, K, N, y, act, want_bias, weight=None, bias=None):
    """(dW [N, K], db [N] or None, dZ) through pg_linear_bwd_w. With an active ops.defer_partials() registry and the
    parameters given, the per-chunk partial rows are left un-summed for the optimiser's launch; dW / db are then only
    placeholders for autograd (None for a parameter's second contribution of the step)."""
    buf = torch.empty(N * K + N, dtype=torch.float32, device=x.device)
    part = torch.empty(lib.pg_linear_bwd_w_scratch(x.size(0), K, N), dtype=torch.float32, device=x.device)
    gw = buf[:N * K].view(N, K)
    gb = buf[N * K:] if want_bias else None
    dz = torch.empty((x.size(0), N), dtype=torch.float32, device=x.device) if act != ACT_NONE else None
    defer = _DEFER is not None and weight is not None and want_bias and bias is not None
    with torch.cuda.device(x.device):
        # (a RowSource: the rows were never gathered — read where they live)
        L.check(linear_bwd_call(lib, g, x, K, N, gw, gb, part, 0 if defer else 1, y=y, act=act, dz=dz), "pg_linear_bwd_w")
    if defer:
        rowlen = N * K + N
        if not _DEFER.add(weight, part, part.numel() // rowlen, rowlen, 0):
            gw = None
        if not _DEFER.add(bias, part, part.numel() // rowlen, rowlen, N * K):
            gb = None
    return gw, gb, (dz if dz is not None else g)


def _bwd_w_pair(lib, g, x1, x2, w1, w2, y, act, has_bias, bias_refs):
    """_bwd_w for both operands of _DualLinear in ONE launch (pg_linear_bwd_w with K2 > 0): (dW1, db1, dW2, db2, dZ) — the same
    partial rows, sums and deferral as two _bwd_w calls"""
    n, N = x1.size(0), w1.size(0)
    K1, K2 = w1.size(1), w2.size(1)
    dev = x2.device
    buf1 = torch.empty(N * K1 + N, dtype=torch.float32, device=dev)
    buf2 = torch.empty(N * K2 + N, dtype=torch.float32, device=dev)
    part1 = torch.empty(lib.pg_linear_bwd_w_scratch(n, K1, N), dtype=torch.float32, device=dev)
    part2 = torch.empty(lib.pg_linear_bwd_w_scratch(n, K2, N), dtype=torch.float32, device=dev)
    gw1, gb1 = buf1[:N * K1].view(N, K1), (buf1[N * K1:] if has_bias[0] else None)
    gw2, gb2 = buf2[:N * K2].view(N, K2), (buf2[N * K2:] if has_bias[1] else None)
    dz = torch.empty((n, N), dtype=torch.float32, device=dev) if act != ACT_NONE else None
    d1 = _DEFER is not None and has_bias[0] and bias_refs[0] is not None
    d2 = _DEFER is not None and has_bias[1] and bias_refs[1] is not None
    defer = d1 and d2
    with torch.cuda.device(dev):
        L.check(linear_bwd_call(lib, g, x1, K1, N, gw1, gb1, part1, 0 if defer else 1, y=y, act=act, dz=dz, x2=x2, K2=K2, dW2=gw2,
                                db2=gb2, part2=part2), "pg_linear_bwd_w (two operands)")
    if defer:
        for (w, b, part, K) in ((w1, bias_refs[0], part1, K1), (w2, bias_refs[1], part2, K2)):
            rowlen = N * K + N
            okw = _DEFER.add(w, part, part.numel() // rowlen, rowlen, 0)
            okb = _DEFER.add(b, part, part.numel() // rowlen, rowlen, N * K)
            if w is w1:
                gw1, gb1 = (gw1 if okw else None), (gb1 if okb else None)
            else:
                gw2, gb2 = (gw2 if okw else None), (gb2 if okb else None)
    return gw1, gb1, gw2, gb2, (dz if dz is not None else g)


class _DualLinear(torch.autograd.Function):
    """GraphSAGE's NodeUpdate dense step y = act(x1 @ W1.T + b1 + x2 @ W2.T + b2) (graphsage_nssc.py:24-29) in
    one MFMA pass (pg_linear_fwd with two operands) instead of two GEMMs, an add, and the activation / concat kernels; the
    backward derives dZ once and runs the weight-gradient kernel per operand."""

    @staticmethod
    def forward(ctx, x1, w1, b1, x2, w2, b2, act):
        lib = L.load()
        n, K1 = x1.shape
        K2 = x2.size(1)
        N = w1.size(0)
        y = torch.empty((n, 2 * N if act == ACT_CONCAT else N), dtype=torch.float32, device=x1.device)
        rows = x1 if isinstance(x1, RowSource) else None
        with torch.cuda.device(x1.device):
            # (a RowSource: fc_self(h) of a layer whose features stayed in the cache / the staged miss block, graphsage_nssc.py:24
            # — the gather is the dense kernel's own LDS fill)
            L.check(linear_fwd_call(lib, x1, w1, b1, y, n, N, act, x2=x2, w2=w2, b2=b2), "pg_linear_fwd (two operands)")
        ctx.rows = rows
        ctx.save_for_backward(x1 if rows is None else None, w1, x2, w2, y if act != ACT_NONE else None)
        ctx.bias = (b1 is not None, b2 is not None)
        ctx.bias_refs = (b1, b2)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        x1, w1, x2, w2, y = ctx.saved_tensors
        if ctx.rows is not None:
            x1 = ctx.rows
        lib = L.load()
        gy = gy.contiguous()
        N = w1.size(0)
        need = ctx.needs_input_grad
        both_or_none = _DEFER is None or all(ctx.bias[i] and ctx.bias_refs[i] is not None for i in (0, 1)) \
            or not any(ctx.bias[i] and ctx.bias_refs[i] is not None for i in (0, 1))
        if (both_or_none and x2.stride(1) == 1 and gy.stride(1) == 1
                and (isinstance(x1, RowSource) or x1.stride(1) == 1)):
            gw1, gb1, gw2, gb2, dz = _bwd_w_pair(lib, gy, x1, x2, w1, w2, y, ctx.act, ctx.bias, ctx.bias_refs)
        else:
            gw1, gb1, dz = _bwd_w(lib, gy, x1, w1.size(1), N, y, ctx.act, ctx.bias[0], w1, ctx.bias_refs[0])
            gw2, gb2, _ = _bwd_w(lib, dz, x2, w2.size(1), N, None, ACT_NONE, ctx.bias[1], w2, ctx.bias_refs[1])
        gx1 = dz @ w1 if need[0] else None
        gx2 = dz @ w2 if need[3] else None
        return gx1, gw1, gb1, gx2, gw2, gb2, None


def _rows_ok(x):
    """tall inputs only (below ~1 K rows the library GEMM is as fast) — unless partial sums are being deferred: a
    parameter applied twice per step must take the SAME path both times (a placeholder gradient from the deferring
    path plus a real one from the library path would be mis-accumulated), whatever the two blocks' row counts are"""
    return x.size(0) >= 1024 or _DEFER is not None


def _skinny_ok(x, w, any_rows=False):
    if isinstance(x, RowSource):
        return x.aligned() and w.size(0) <= 64 and w.is_contiguous() and w.size(1) == x.dim
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and w.size(0) <= 64 and (any_rows or _rows_ok(x))
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and w.is_contiguous())


def linear2(x1, mod1, x2, mod2, act=ACT_NONE):
    """act(mod1(x1) + mod2(x2)) — GraphSAGE's NodeUpdate — on one MFMA pass when both operands fit the
    skinny envelope, else through the modules"""
    virt = isinstance(x1, RowSource)      # rows that exist nowhere as a tensor: the kernel is the only way, however few
    if x1.size(0) == x2.size(0) and mod1.weight.size(0) == mod2.weight.size(0) and _skinny_ok(x1, mod1.weight) \
            and _skinny_ok(x2, mod2.weight, any_rows=virt):
        if _DEFER is not None and (mod1.bias is None or mod2.bias is None):
            _DEFER.saw_plain(mod1.weight, mod1.bias, mod2.weight, mod2.bias)     # _bwd_w does not defer without a bias
        return _DualLinear.apply(x1, mod1.weight, mod1.bias, x2, mod2.weight, mod2.bias, act)
    if isinstance(x1, RowSource) or isinstance(x2, RowSource):
        raise L.PgError("linear2: an un-materialised operand (ops.RowSource) needs the skinny dense kernels: first operand "
                        "only, 16-byte aligned rows, at most 64 outputs (a model's virtual_inputs() promised that)")
    return _apply_act(linear(x1, mod1) + linear(x2, mod2), act)


def linear(x, module, act=ACT_NONE):
    """NodeUpdate's nn.Linear (+ activation): tall inputs (thousands of rows, <= 64 outputs) go through
    _SkinnyLinear, anything else through the module itself"""
    w, b = module.weight, module.bias
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and w.size(0) <= 64 and _rows_ok(x)
            and x.stride(1) == 1):
        if _DEFER is not None and b is None:
            _DEFER.saw_plain(w)                # _SkinnyLinear defers only with a bias
        y = _SkinnyLinear.apply(x, w, b, act)
        if act == ACT_CONCAT:
            y._pg_concat_n = w.size(0)         # lets the consumer's backward produce this layer's dZ (see _DZ_STASH)
        return y
    if _DEFER is not None:
        _DEFER.saw_plain(w, b)
    return _apply_act(module(x), act)


class _CrossEntropy(torch.autograd.Function):
    """loss head: row pass + one-block reduction forward, one scale backward (pg_loss.hip)"""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        lib = L.load()
        n, C = logits.shape
        if logits.stride(1) != 1:
            logits = logits.contiguous()
        labels = labels.contiguous()
        need_grad = ctx.needs_input_grad[0]
        dl = torch.empty((n, C), dtype=torch.float32, device=logits.device) if need_grad else None
        scratch = torch.empty(n + 2, dtype=torch.float32, device=logits.device)
        meta = scratch[n:]
        with torch.cuda.device(logits.device):
            L.check(lib.pg_xent_fwd(L.ptr(logits), logits.stride(0), L.ptr(labels), n, C, int(ignore_index),
                                    L.ptr(dl), C, L.ptr(scratch), L.ptr(meta), L.stream_ptr()), "pg_xent_fwd")
        if need_grad:
            ctx.save_for_backward(dl, meta)
        return meta[0]

    @staticmethod
    def backward(ctx, g):
        dl, meta = ctx.saved_tensors
        lib = L.load()
        gx = torch.empty_like(dl)
        g = g.contiguous().to(torch.float32)
        with torch.cuda.device(dl.device):
            L.check(lib.pg_xent_bwd(L.ptr(dl), dl.stride(0), dl.size(0), dl.size(1), L.ptr(meta), L.ptr(g), L.ptr(gx),
                                    gx.stride(0), L.stream_ptr()), "pg_xent_bwd")
        return gx, None, None


def cross_entropy(logits, labels, ignore_index=-100):
    """torch.nn.functional.cross_entropy(logits, labels) (mean, no class weights) on the HIP loss head"""
    if (not logits.is_cuda or logits.dtype != torch.float32 or logits.dim() != 2 or labels.dtype != torch.int64
            or labels.dim() != 1 or labels.size(0) != logits.size(0) or logits.size(0) == 0):
        raise L.PgError("cross_entropy needs fp32 CUDA logits [n, C] and int64 labels [n] (no CPU fallback)")
    return _CrossEntropy.apply(logits, labels, ignore_index)


class CrossEntropyLoss(torch.nn.Module):
    """drop-in for the trainer scripts' `torch.nn.CrossEntropyLoss()` (pa_gcn.py:80)"""

    def __init__(self, ignore_index=-100):
        super().__init__()
        self.ignore_index = ignore_index

    def forward(self, logits, labels):
        return cross_entropy(logits, labels, self.ignore_index)


def fused_loss(loss_fcn):
    """the HIP loss head when `loss_fcn` is a plain torch.nn.CrossEntropyLoss (mean, unweighted, no label
    smoothing) — what pa_gcn.py / pa_gs.py construct — otherwise `loss_fcn` itself"""
    if (type(loss_fcn) is torch.nn.CrossEntropyLoss and loss_fcn.weight is None and loss_fcn.reduction == 'mean'
            and getattr(loss_fcn, 'label_smoothing', 0.0) == 0.0):
        return CrossEntropyLoss(loss_fcn.ignore_index)
    return loss_fcn


def head_desc(indptr, src, h, weight, bias, labels, n_valid, grad_seed, ignore_index, reduce, drop_struct, logits, dagg, part, dW,
              db_loss, flags, h_self=None, w_self=None, b_self=None, dself=None):
    """pg_head_desc_t of one output-head launch (include/pagraph_hip.h); the tensors must stay alive until it is enqueued"""
    d = L.PgHeadDesc()
    d.indptr, d.src, d.h, d.W, d.bias = L.ptr(indptr).value, L.ptr(src).value, L.ptr(h).value, L.ptr(weight).value, L.ptr(bias).value
    d.labels, d.n_valid_dev, d.grad_scale_dev = L.ptr(labels).value, L.ptr(n_valid).value, L.ptr(grad_seed).value
    d.ignore_index, d.n_dst = int(ignore_index), labels.numel()
    d.h_stride, d.K, d.C, d.reduce, d.flags = h.stride(0), weight.size(1), weight.size(0), _REDUCE[reduce], int(flags)
    d.has_drop = 1 if drop_struct is not None else 0
    if drop_struct is not None:
        d.drop = drop_struct
    d.logits, d.dagg, d.partials = L.ptr(logits).value, L.ptr(dagg).value, L.ptr(part).value
    d.dW, d.db_loss = L.ptr(dW).value, L.ptr(db_loss).value
    if h_self is not None:
        d.h_self, d.W_self, d.bias_self, d.dself = L.ptr(h_self).value, L.ptr(w_self).value, L.ptr(b_self).value, L.ptr(dself).value
        d.hs_stride, d.Ks = h_self.stride(0), w_self.size(1)
    return d


class _GCNHead(torch.autograd.Function):
    """last aggregation (+ dropout) -> output linear layer -> CrossEntropyLoss and all their gradients in one
    pass (pg_head). The gradients are computed in the forward, already multiplied by `grad_seed` (a device
    scalar: d objective / d loss); backward hands them out when it is called with that very tensor and
    rescales otherwise."""

    @staticmethod
    def forward(ctx, indptr, src, h, weight, bias, labels, n_valid, grad_seed, ignore_index, reduce, drop, tptr, tdst,
                heavy, want_logits, dz_n=0):
        lib = L.load()
        h = h.contiguous()
        n_dst = labels.numel()
        C, K = weight.shape
        buf = torch.empty(C * K + C + 1, dtype=torch.float32, device=h.device)
        gw, gbl = buf[:C * K].view(C, K), buf[C * K:]
        dagg = torch.empty((n_dst, K), dtype=torch.float32, device=h.device)
        logits = torch.empty((n_dst, C), dtype=torch.float32, device=h.device) if want_logits else None
        part = torch.empty(lib.pg_gcn_head_scratch(n_dst, K, C), dtype=torch.float32, device=h.device)
        d = drop.struct() if drop is not None and drop.threshold else None
        # deferral needs the registered gradient seed (the gradients leave the kernel already scaled) and a bias
        defer = _DEFER is not None and bias is not None and grad_seed is not None
        hd = head_desc(indptr, src, h, weight, bias, labels, n_valid, grad_seed, ignore_index, reduce, d, logits, dagg, part, gw,
                       gbl, (0 if defer else L.PG_HEAD_SUM_PARTIALS) | L.PG_HEAD_DAGG_PER_EDGE)
        with torch.cuda.device(h.device):
            L.check(lib.pg_head(ctypes.byref(hd), L.stream_ptr()), "pg_head")
        if defer:
            rowlen = lib.pg_gcn_head_row_len(K, C)      # C * K + C + 1 padded to whole 16-byte pieces
            chunks = part.numel() // rowlen
            _DEFER.add(weight, part, chunks, rowlen, 0)
            _DEFER.add(bias, part, chunks, rowlen, C * K)
            _DEFER.extra.append((gbl[C:C + 1], part, chunks, rowlen, C * K + C))     # the loss value
        use_t = tptr is not None and tptr.numel() == h.size(0) + 1
        ctx.dz_n = dz_n if use_t and _dz_fusable(h, dz_n) else 0
        ctx.save_for_backward(indptr, src, dagg, gw, gbl, grad_seed, *((tptr, tdst, heavy) if use_t else ()),
                              *((h,) if ctx.dz_n else ()))
        ctx.use_t, ctx.n_src, ctx.reduce, ctx.drop = use_t, h.size(0), reduce, (drop if d is not None else None)
        ctx.has_bias = bias is not None
        loss = gbl[C]
        if want_logits:
            ctx.mark_non_differentiable(logits)
            return loss, logits
        return loss

    @staticmethod
    def backward(ctx, g, *unused):
        saved = ctx.saved_tensors
        indptr, src, dagg, gw, gbl, seed = saved[:6]
        if seed is None or g.data_ptr() != seed.data_ptr():
            k = g if seed is None else g / seed             # not the registered seed: rescale (extra launches)
            dagg, gw, gbl = dagg * k, gw * k, gbl * k
        lib = L.load()
        gh = None
        if ctx.needs_input_grad[2]:
            K = dagg.size(1)
            d = ctx.drop.struct() if ctx.drop is not None else None
            with torch.cuda.device(dagg.device):
                if ctx.use_t:
                    tptr, tdst, heavy = saved[6:9]
                    gh = torch.empty((ctx.n_src, K), dtype=torch.float32, device=dagg.device)
                    y = saved[9] if ctx.dz_n else None
                    dz = torch.empty((ctx.n_src, ctx.dz_n), dtype=torch.float32, device=dagg.device) if ctx.dz_n else None
                    # dagg left the head per edge (PG_HEAD_DAGG_PER_EDGE): the mean's backward is a plain sum of it
                    L.check(spmm_bwd_call(lib, dagg, gh, ctx.n_src, L.PG_REDUCE_SUM, indptr=indptr, transposed=(tptr, tdst, heavy),
                                          drop_struct=d, act_out=y, dz=dz), "pg_spmm_bwd (gather form)")
                    if dz is not None:
                        _stash_dz(gh, dz)
                else:
                    gh = torch.zeros((ctx.n_src, K), dtype=torch.float32, device=dagg.device)
                    L.check(spmm_bwd_call(lib, dagg, gh, ctx.n_src, L.PG_REDUCE_SUM, indptr=indptr, src=src, drop_struct=d),
                            "pg_spmm_bwd (scatter form)")
        C = gw.size(0)
        return (None, None, gh, gw, gbl[:C] if ctx.has_bias else None) + (None,) * 11


class _SageHead(torch.autograd.Function):
    """_GCNHead for GraphSAGE's output NodeUpdate z = fc_neigh(aggregate(dropout(h))) + fc_self(h_self)
    (graphsage_nssc.py:24, pg_head): loss and every gradient — dAgg (scattered back over the block in backward),
    dSelf, both weight matrices, both biases — in one pass, already multiplied by `grad_seed`."""

    @staticmethod
    def forward(ctx, indptr, src, h, w_n, b_n, h_self, w_s, b_s, labels, n_valid, grad_seed, ignore_index, reduce, drop,
                tptr, tdst, heavy):
        lib = L.load()
        h = h.contiguous()
        n_dst = labels.numel()
        C, K = w_n.shape
        Ks = w_s.size(1)
        Kt = K + Ks
        buf = torch.empty(C * Kt + C + 1, dtype=torch.float32, device=h.device)
        gwn, gws, gbl = buf[:C * K].view(C, K), buf[C * K:C * Kt].view(C, Ks), buf[C * Kt:]
        dagg = torch.empty((n_dst, K), dtype=torch.float32, device=h.device)
        dself = torch.empty((n_dst, Ks), dtype=torch.float32, device=h.device)
        part = torch.empty(lib.pg_gcn_head_scratch(n_dst, Kt, C), dtype=torch.float32, device=h.device)
        d = drop.struct() if drop is not None and drop.threshold else None
        defer = _DEFER is not None and b_n is not None and b_s is not None and grad_seed is not None
        hd = head_desc(indptr, src, h, w_n, b_n, labels, n_valid, grad_seed, ignore_index, reduce, d, None, dagg, part, buf, gbl,
                       (0 if defer else L.PG_HEAD_SUM_PARTIALS) | L.PG_HEAD_DAGG_PER_EDGE, h_self=h_self, w_self=w_s, b_self=b_s,
                       dself=dself)
        with torch.cuda.device(h.device):
            L.check(lib.pg_head(ctypes.byref(hd), L.stream_ptr()), "pg_head (GraphSAGE)")
        ok = [True] * 4
        if defer:
            rowlen = lib.pg_gcn_head_row_len(Kt, C)
            chunks = part.numel() // rowlen
            ok[0] = _DEFER.add(w_n, part, chunks, rowlen, 0)
            ok[1] = _DEFER.add(w_s, part, chunks, rowlen, C * K)
            ok[2] = _DEFER.add(b_n, part, chunks, rowlen, C * Kt)
            ok[3] = _DEFER.add(b_s, part, chunks, rowlen, C * Kt)
            _DEFER.extra.append((gbl[C:C + 1], part, chunks, rowlen, C * Kt + C))     # the loss value
        use_t = tptr is not None and tptr.numel() == h.size(0) + 1
        ctx.save_for_backward(indptr, src, dagg, dself, gwn, gws, gbl, grad_seed, *((tptr, tdst, heavy) if use_t else ()))
        ctx.use_t, ctx.n_src, ctx.reduce, ctx.drop = use_t, h.size(0), reduce, (drop if d is not None else None)
        ctx.has_bias = (b_n is not None, b_s is not None)
        ctx.deferred_ok = ok
        # both biases have the gradient db. Handing autograd the SAME tensor for two parameters makes AccumulateGrad copy it
        # (one more launch per replayed step); when the optimiser adds the partial rows up itself the tensors autograd sees
        # are placeholders anyway: the second bias gets one of its own
        # (made in backward: a tensor this context also held could not be stolen by AccumulateGrad either)
        ctx.gb_self_placeholder = bool(defer)
        return gbl[C]

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        indptr, src, dagg, dself, gwn, gws, gbl, seed = saved[:8]
        if seed is None or g.data_ptr() != seed.data_ptr():
            k = g if seed is None else g / seed
            dagg, dself, gwn, gws, gbl = dagg * k, dself * k, gwn * k, gws * k, gbl * k
        lib = L.load()
        gh = None
        if ctx.needs_input_grad[2]:
            K = dagg.size(1)
            d = ctx.drop.struct() if ctx.drop is not None else None
            with torch.cuda.device(dagg.device):
                if ctx.use_t:
                    tptr, tdst, heavy = saved[8:11]
                    gh = torch.empty((ctx.n_src, K), dtype=torch.float32, device=dagg.device)
                    L.check(spmm_bwd_call(lib, dagg, gh, ctx.n_src, L.PG_REDUCE_SUM, indptr=indptr, transposed=(tptr, tdst, heavy),
                                          drop_struct=d), "pg_spmm_bwd (gather form)")
                else:
                    gh = torch.zeros((ctx.n_src, K), dtype=torch.float32, device=dagg.device)
                    L.check(spmm_bwd_call(lib, dagg, gh, ctx.n_src, L.PG_REDUCE_SUM, indptr=indptr, src=src, drop_struct=d),
                            "pg_spmm_bwd (scatter form)")
        C = gwn.size(0)
        ok = ctx.deferred_ok
        gb = gbl[:C]
        gbs = torch.empty_like(gb) if ctx.gb_self_placeholder else gb
        return (None, None, gh, gwn if ok[0] else None, (gb if ok[2] else None) if ctx.has_bias[0] else None,
                dself if ctx.needs_input_grad[5] else None, gws if ok[1] else None,
                (gbs if ok[3] else None) if ctx.has_bias[1] else None) + (None,) * 9


def head_fits(n_agg_cols, n_self_cols, n_classes, weights, dropout_active):
    """the STATIC part of pg_head / pg_head's envelope, from the module alone: a model asks BEFORE it runs (and
    consumes) the layers below the head — a head that declines afterwards leaves a NodeFlow whose frames were popped
    (`pa_gs.py --n-hidden 32` raised KeyError('features') in the fall-back forward)."""
    if n_agg_cols + n_self_cols > 64 or n_classes > 64:
        return False
    if dropout_active and n_agg_cols % 4:
        return False
    return all(w.is_cuda and w.dtype == torch.float32 and w.is_contiguous() for w in weights)


def sage_head(indptr, src, h, fc_neigh, h_self, fc_self, labels, n_valid, grad_seed=None, ignore_index=-100, reduce="mean",
              dropout=None, transpose=None):
    """loss of GraphSAGE's output layer over the last NodeFlow block: CrossEntropyLoss(fc_neigh(aggregate(dropout(h))) +
    fc_self(h_self)) with every gradient produced in the same pass (pg_head). None when the shapes are outside the
    kernel's envelope (the caller then runs the unfused path)."""
    wn, bn, ws, bs = fc_neigh.weight, fc_neigh.bias, fc_self.weight, fc_self.bias
    ok = lambda t: torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1
    if not (ok(h) and ok(h_self) and wn.size(1) == h.size(1) and ws.size(1) == h_self.size(1) and ws.size(0) == wn.size(0)
            and wn.size(1) + ws.size(1) <= 64 and wn.size(0) <= 64 and wn.is_contiguous() and ws.is_contiguous()
            and labels.dtype == torch.int64 and labels.numel() > 0 and h_self.size(0) == labels.numel()
            and reduce in ("mean", "sum")):
        return None
    if dropout is not None and dropout.threshold and h.size(1) % 4:
        return None
    tptr, tdst, heavy = (tuple(transpose) + (None,))[:3] if transpose is not None else (None, None, None)
    return _SageHead.apply(indptr, src, h, wn, bn, h_self, ws, bs, labels.contiguous(), n_valid, grad_seed, ignore_index,
                           reduce, dropout, tptr, tdst, heavy)


def gcn_head(indptr, src, h, linear, labels, n_valid, grad_seed=None, ignore_index=-100, reduce="mean", dropout=None,
             transpose=None, want_logits=False):
    """loss (and optionally logits) of the sampled GCN's output layer over the last NodeFlow block:
    CrossEntropyLoss(linear(aggregate(dropout(h)))) with every gradient produced in the same pass.
    n_valid: device int32[1], number of labels != ignore_index (pg_gather_labels). Returns None when the
    shapes are outside the kernel's envelope (callers then run the unfused path)."""
    w, b = linear.weight, linear.bias
    if not (h.is_cuda and h.dtype == torch.float32 and h.dim() == 2 and h.stride(1) == 1 and w.size(1) == h.size(1)
            and w.size(1) <= 64 and w.size(0) <= 64 and w.is_contiguous() and labels.dtype == torch.int64
            and labels.numel() > 0):
        return None
    if dropout is not None and dropout.threshold and h.size(1) % 4:
        return None
    tptr, tdst, heavy = (tuple(transpose) + (None,))[:3] if transpose is not None else (None, None, None)
    return _GCNHead.apply(indptr, src, h, w, b, labels.contiguous(), n_valid, grad_seed, ignore_index, reduce, dropout,
                          tptr, tdst, heavy, want_logits, int(getattr(h, '_pg_concat_n', 0)))
