"""Full-neighbour NodeFlow — what the reference's evaluation builds with
`NeighborSampler(graph, len(test_nid), graph.number_of_nodes(), neighbor_type='in', num_hops=...)`
(examples/eval.py:20-26: expand_factor = V, i.e. every in-neighbour). Not a hot path: one NodeFlow per
evaluation, built with device-wide tensor ops (the frontier expansion itself is the closure kernel,
pg_frontier_mark_neighbors, that get_sub_graph uses)."""
import torch

from .. import _lib as L
from .nodeflow import NodeFlow


def full_neighbor_nodeflow(g, seed_nodes, num_hops):
    """NodeFlow with num_hops + 1 layers whose top layer is `seed_nodes` (in order) and whose block i holds EVERY
    in-edge of layer i+1's vertices; lower layers are de-duplicated and ascending (the sampler spec's rules 4-5,
    DESIGN.md), block sources are positions in the layer below, in adjacency (ascending-id) order."""
    dev = g.device
    lib = L.load()
    V = g.number_of_nodes()
    layers = [torch.as_tensor(seed_nodes).to(dev, torch.int64).contiguous()]
    blocks = []
    n_words = (V + 63) // 64
    with torch.cuda.device(dev):
        sp = L.stream_ptr()
        for _ in range(num_hops):
            dst = layers[0]
            beg, end = g.indptr[dst], g.indptr[dst + 1]
            deg = end - beg
            ip = torch.zeros(dst.numel() + 1, dtype=torch.int64, device=dev)
            ip[1:] = torch.cumsum(deg, 0)
            total = int(ip[-1])
            if total >= 2 ** 31:
                raise L.PgError("full-neighbour block with more than 2^31 edges")
            # edge e of destination j reads indices[beg[j] + (e - ip[j])]
            owner = torch.repeat_interleave(torch.arange(dst.numel(), device=dev), deg)
            src_ids = g.indices[beg[owner] + (torch.arange(total, device=dev) - ip[owner])].long()
            # the layer below = distinct sources, ascending: mark them in a bitmap, list the set bits
            bm = torch.zeros(n_words, dtype=torch.int64, device=dev)
            uniq = torch.unique(dst)
            L.check(lib.pg_frontier_mark_neighbors(L.ptr(g.indptr), L.ptr(g.indices), L.ptr(uniq), uniq.numel(),
                                                   L.ptr(bm), 0, sp), "pg_frontier_mark_neighbors")
            from ..partition.utils import _bitmap_ids
            below = _bitmap_ids(lib, bm, V, dev)
            pos = torch.searchsorted(below, src_ids)
            layers.insert(0, below)
            blocks.insert(0, (ip.to(torch.int32), pos.to(torch.int32)))
    offs = [0]
    for l in layers:
        offs.append(offs[-1] + int(l.numel()))
    return NodeFlow(torch.cat(layers), offs, [b[0] for b in blocks], [b[1] for b in blocks])
