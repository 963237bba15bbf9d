"""Feature provider — what server/pa_server.py:15-61 publishes into the DGL shared-memory
store, built in-process: `features` (feat.npy or U[0,1) fallback) and, for GCN,
`norm = 1/in_degree` (pa_server.py:43, inf for isolated vertices exactly as there) plus the
optional one-hop preprocessing X' = norm * (A^T X) (pa_server.py:45-52)."""
import ctypes

import numpy as np
import scipy.sparse as spsp
import torch

from . import data
from .storage import HostFeatureStore


def preprocess_features(csc, features, norm, chunk_rows=1 << 20):
    """pa_server.py:45-52: X'[v] = norm[v] * sum_{u->v} X[u] (update_all copy_src/sum, then * norm) as a
    one-off SpMM on the GPU: the full graph's CSC *is* a NodeFlow block (destinations = all vertices,
    sources = all vertices), so the aggregation kernel of the training path (pg_spmm_fwd, reduce = sum)
    does it, destination chunk by chunk."""
    from . import _lib as L
    lib = L.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    V, Fdim = features.shape
    x = features.to(dev).contiguous()
    # the graph's offsets are 64-bit (scipy gives int64 indptr beyond 2^31 entries); a CHUNK of destinations is a
    # NodeFlow block with 32-bit offsets relative to its first edge: the chunk's indptr is rebased and the kernel gets
    # `indices` advanced to that edge, so nothing limits the graph's total edge count
    indptr = torch.as_tensor(np.ascontiguousarray(csc.indptr, dtype=np.int64)).to(dev)
    indices = csc.indices if isinstance(csc.indices, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(csc.indices, dtype=np.int32))
    indices = indices.to(dev, torch.int32)
    nrm = torch.as_tensor(norm, dtype=torch.float32).to(dev)
    out = torch.empty((V, Fdim), dtype=torch.float32)
    lo = 0
    while lo < V:
        hi = min(V, lo + chunk_rows)
        e0 = int(indptr[lo])
        while int(indptr[hi]) - e0 >= 2 ** 31 - 1:       # (a chunk of hub destinations: shrink until its edges fit)
            if hi - lo == 1:
                raise L.PgError("preprocess_features: one vertex with >= 2^31 in-edges")
            hi = lo + max(1, (hi - lo) // 2)
        rel = (indptr[lo:hi + 1] - e0).to(torch.int32)
        agg = torch.empty((hi - lo, Fdim), dtype=torch.float32, device=dev)
        L.check(lib.pg_spmm_fwd(L.ptr(rel), ctypes.c_void_p(indices.data_ptr() + 4 * e0), L.ptr(x), x.stride(0), hi - lo, Fdim,
                                L.PG_REDUCE_SUM, L.ptr(agg), agg.stride(0), L.stream_ptr()), "pg_spmm_fwd")
        out[lo:hi] = (agg * nrm[lo:hi]).cpu()
        lo = hi
    return out


def _build_fields(dataset, model, preprocess):
    coo_adj, feat = data.get_graph_data(dataset, mmap=True)
    if isinstance(feat, torch.Tensor):
        features = feat
    else:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")      # read-only memmap: the table is only ever read from here on
            features = torch.as_tensor(np.asarray(feat), dtype=torch.float32)
    fields = {}
    if model == 'gcn':
        csc = spsp.csc_matrix(coo_adj)
        csc.sum_duplicates()
        in_deg = torch.from_numpy(np.diff(csc.indptr).astype(np.float32))
        norm = (1. / in_deg).unsqueeze(1)
        if preprocess:
            print('Preprocessing features...')
            features = preprocess_features(csc, features, norm)
        fields['norm'] = norm
        fields['features'] = features
    elif model == 'graphsage':
        if preprocess:
            print('preprocessing: warning: jusy copy')
            fields['neigh'] = features          # the same tensor: HostFeatureStore keeps ONE host copy for both names
        fields['features'] = features
    else:
        raise ValueError(model)
    return fields


def load_store(dataset, model='gcn', preprocess=False, pin=True, shared=None, local_rank=None):
    """the feature tables of pa_server.py:38-54. `shared` (default: whenever a process group with more than one rank
    is up): the tables exist ONCE per node in shared memory — local rank 0 loads / preprocesses and publishes,
    the other ranks attach (HostFeatureStore.shared) — instead of one pinned copy per rank."""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if shared is None:
        shared = multi
    if shared and multi:
        if local_rank is None:
            import os
            local_rank = int(os.environ.get("LOCAL_RANK", dist.get_rank()))
        return HostFeatureStore.shared(lambda: _build_fields(dataset, model, preprocess), local_rank,
                                       tag="store", register=pin)
    return HostFeatureStore(_build_fields(dataset, model, preprocess), pin=pin)
