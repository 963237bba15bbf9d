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
