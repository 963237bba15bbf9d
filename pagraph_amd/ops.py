"""autograd wrappers over the HIP aggregation kernels (pagraph_amd/csrc/pg_spmm.hip)."""
import torch

from . import _lib as L

_REDUCE = {"mean": L.PG_REDUCE_MEAN, "sum": L.PG_REDUCE_SUM}


class _BlockAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, indptr, src, h, n_dst, reduce):
        lib = L.load()
        h = h.contiguous()
        out = torch.empty((n_dst, h.size(1)), dtype=torch.float32, device=h.device)
        with torch.cuda.device(h.device):
            L.check(lib.pg_spmm_fwd(L.ptr(indptr), L.ptr(src), L.ptr(h), h.stride(0), n_dst, h.size(1),
                                    _REDUCE[reduce], L.ptr(out), out.stride(0), L.stream_ptr()), "pg_spmm_fwd")
        ctx.save_for_backward(indptr, src)
        ctx.n_src, ctx.reduce = h.size(0), reduce
        return out

    @staticmethod
    def backward(ctx, grad_out):
        indptr, src = ctx.saved_tensors
        if not ctx.needs_input_grad[2]:
            return None, None, None, None, None
        lib = L.load()
        go = grad_out.contiguous()
        gh = torch.zeros((ctx.n_src, go.size(1)), dtype=torch.float32, device=go.device)
        with torch.cuda.device(go.device):
            L.check(lib.pg_spmm_bwd(L.ptr(indptr), L.ptr(src), L.ptr(go), go.stride(0), go.size(0), go.size(1),
                                    _REDUCE[ctx.reduce], L.ptr(gh), gh.stride(0), L.stream_ptr()), "pg_spmm_bwd")
        return None, None, gh, None, None


def block_aggregate(indptr, src, h, n_dst, reduce="mean"):
    """out[v] = reduce_{e in block, dst(e)=v} h[src(e)]  (DGL copy_src + mean|sum)"""
    if h.dtype != torch.float32 or not h.is_cuda:
        raise L.PgError("block_aggregate needs fp32 CUDA tensors (no CPU fallback)")
    return _BlockAggregate.apply(indptr, src, h, int(n_dst), reduce)


ACT_NONE, ACT_RELU, ACT_CONCAT = 0, 1, 2


def _apply_act(z, act):
    if act == ACT_RELU:
        return torch.relu(z)
    if act == ACT_CONCAT:
        return torch.cat((z, torch.relu(z)), dim=1)
    return z


class _SkinnyLinear(torch.autograd.Function):
    """NodeUpdate's dense step y = act(x @ W.T + b) with the tall-skinny pieces on the fp32-MFMA kernels
    of pg_dense.hip: forward (bias + activation / skip-concat fused in the epilogue) when out_features <= 32
    and K % 8 == 0; weight / bias gradient (a reduction over all rows into a tiny [N, K] matrix, the
    activation's derivative applied on the fly) always."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        lib = L.load()
        n, K = x.shape
        N = weight.size(0)
        if (N <= 32 and K % 8 == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and weight.is_contiguous()
                and weight.data_ptr() % 16 == 0):
            y = torch.empty((n, 2 * N if act == ACT_CONCAT else N), dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                L.check(lib.pg_linear_fwd(L.ptr(x), x.stride(0), L.ptr(weight), L.ptr(bias), L.ptr(y), y.stride(0), n, K,
                                          N, act, L.stream_ptr()), "pg_linear_fwd")
        else:
            y = _apply_act(torch.nn.functional.linear(x, weight, bias), act)
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        ctx.has_bias = bias is not None
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        lib = L.load()
        gy = gy.contiguous()
        act = ctx.act
        gx = gw = gb = None
        N, K = weight.shape
        gz = gy if act == ACT_NONE else None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            buf = torch.zeros(N * K + N, dtype=torch.float32, device=x.device)   # one fill for dW and db
            gw = buf[:N * K].view(N, K)
            gb = buf[N * K:] if ctx.has_bias else None
            dz = torch.empty((x.size(0), N), dtype=torch.float32, device=x.device) if act != ACT_NONE else None
            with torch.cuda.device(x.device):
                L.check(lib.pg_linear_bwd_w(L.ptr(gy), gy.stride(0), L.ptr(x), x.stride(0), x.size(0), K, N,
                                            L.ptr(gw), L.ptr(gb), L.ptr(y), y.stride(0) if y is not None else 0, act,
                                            L.ptr(dz), L.stream_ptr()), "pg_linear_bwd_w")
            if dz is not None:
                gz = dz
        if ctx.needs_input_grad[0]:
            if gz is None:
                gz = gy * (y > 0) if act == ACT_RELU else gy[:, :N] + gy[:, N:] * (y[:, :N] > 0)
            gx = gz @ weight
        return gx, gw, gb, None


def linear(x, module, act=ACT_NONE):
    """NodeUpdate's nn.Linear (+ activation): tall inputs (thousands of rows, <= 64 outputs) go through
    _SkinnyLinear, anything else through the module itself"""
    w, b = module.weight, module.bias
    if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and w.size(0) <= 64 and x.size(0) >= 1024
            and x.stride(1) == 1):
        return _SkinnyLinear.apply(x, w, b, act)
    return _apply_act(module(x), act)
