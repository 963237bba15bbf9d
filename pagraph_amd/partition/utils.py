"""L-hop closure sub-graph of a partition's train vertices — counterpart of
PaGraph/partition/utils.py:9-52 (get_sub_graph).  The reference runs DGL's
NeighborSampler with fan-out = |V| (full neighbours) and post-processes the block
edges with numpy; here the frontier expansion is the HIP kernel
pg_frontier_mark_neighbors + bitmap compaction, and the relabelling works on
sorted id lists in HBM.

Sub-graph definition (what utils.py:25-44 computes): with F_0 = train vertices and
F_{h+1} = in-neighbours(F_h) (per-layer dedup), D = F_0 u ... u F_{hops-1}:
  edges     = every in-edge (u -> v) of every v in D          (union of block edges, deduped)
  sub2full  = sorted unique endpoints of those edges           (utils.py:33)
  subtrain  = full2sub[unique(train)] with the clamp of utils.py:48-51 and full2sub's
              zero default for vertices that are not endpoints (utils.py:34)
"""
import numpy as np
import scipy.sparse as spsp
import torch

from .. import _lib as L


def _bitmap_ids(lib, bitmap, V, device):
    n_words = bitmap.numel()
    out = torch.empty(V, dtype=torch.int64, device=device)
    cnt = torch.zeros(1, dtype=torch.int64, device=device)
    scratch = torch.empty(n_words // 1024 + 8, dtype=torch.int32, device=device)
    L.check(lib.pg_bitmap_to_ids(L.ptr(bitmap), n_words, L.ptr(out), V, L.ptr(cnt), None, L.ptr(scratch),
                                 L.stream_ptr()), "pg_bitmap_to_ids")
    return out[:int(cnt.item())]


def closure_device(g, train_nid, num_hops):
    """-> (sub_indptr int64 [Vs+1], sub_indices int32 [nnz_s]  (CSC of the sub-graph: column =
    destination, entries = source sub-ids ascending), sub2full int64 [Vs], subtrainid int64)"""
    lib = L.load()
    dev = g.device
    V = g.number_of_nodes()
    n_words = (V + 63) // 64
    train = torch.as_tensor(train_nid).to(dev, torch.int64).contiguous()
    with torch.cuda.device(dev):
        sp = L.stream_ptr()
        dmask = torch.zeros(V, dtype=torch.bool, device=dev)       # D: destinations of some block
        frontier = torch.unique(train)
        for h in range(num_hops):
            dmask[frontier] = True
            if h + 1 < num_hops:
                bm = torch.zeros(n_words, dtype=torch.int64, device=dev)
                L.check(lib.pg_frontier_mark_neighbors(L.ptr(g.indptr), L.ptr(g.indices), L.ptr(frontier),
                                                       frontier.numel(), L.ptr(bm), 0, sp),
                        "pg_frontier_mark_neighbors")
                frontier = _bitmap_ids(lib, bm, V, dev)
        deg = g.indptr[1:] - g.indptr[:-1]
        dst_ids = torch.nonzero(dmask & (deg > 0)).squeeze(1)       # destinations that own >= 1 edge
        # endpoints = those destinations + all their in-neighbours
        bm = torch.zeros(n_words, dtype=torch.int64, device=dev)
        L.check(lib.pg_frontier_mark_neighbors(L.ptr(g.indptr), L.ptr(g.indices), L.ptr(dst_ids), dst_ids.numel(),
                                               L.ptr(bm), 1, sp), "pg_frontier_mark_neighbors")
        sub2full = _bitmap_ids(lib, bm, V, dev)
        Vs = sub2full.numel()
        full2sub = torch.zeros(V, dtype=torch.int64, device=dev)    # utils.py:34 (zeros default)
        full2sub[sub2full] = torch.arange(Vs, device=dev)
        # sub CSC: column j keeps the whole in-list of sub2full[j] when it is a destination
        is_dst = dmask[sub2full]
        sdeg = torch.where(is_dst, deg[sub2full], torch.zeros_like(sub2full))
        sub_indptr = torch.zeros(Vs + 1, dtype=torch.int64, device=dev)
        sub_indptr[1:] = torch.cumsum(sdeg, 0)
        nnz = int(sub_indptr[-1].item())
        # expand: edge e of column j reads indices[indptr[full] + (e - sub_indptr[j])]
        col = torch.repeat_interleave(torch.arange(Vs, device=dev), sdeg, output_size=nnz)
        within = torch.arange(nnz, device=dev) - sub_indptr[col]
        full_src = g.indices[g.indptr[sub2full[col]] + within].long()
        sub_indices = full2sub[full_src].to(torch.int32)
        # utils.py:47-52
        tnid = train
        valid_t_max = sub2full.max()
        valid_t_min = tnid.min()
        tnid = torch.where(tnid <= valid_t_max, tnid, valid_t_min)
        subtrainid = full2sub[torch.unique(tnid)]
    return sub_indptr, sub_indices, sub2full, subtrainid


def get_sub_graph(g, train_nid, num_hops):
    """reference-shaped return: (scipy CSR adj with row = src, col = dst, uint8 ones;
    sub2full ndarray; subtrainid ndarray)  — utils.py:52"""
    ip, ix, sub2full, subtrain = closure_device(g, train_nid, num_hops)
    Vs = sub2full.numel()
    csc = spsp.csc_matrix((np.ones(ix.numel(), dtype=np.uint8), ix.cpu().numpy(), ip.cpu().numpy()), shape=(Vs, Vs))
    csr = csc.tocsr()
    print('vertex#: {} edge#: {}'.format(Vs, csr.data.shape[0]))
    return csr, sub2full.cpu().numpy(), subtrain.cpu().numpy()
