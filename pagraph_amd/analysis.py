"""Cache-policy analysis — counterparts of examples/opt_cache_hit.py:16-31 and
examples/count_vnum.py:16-20: how often each vertex's features are requested in an epoch, the
hit rate an ORACLE cache of a given size would reach (most-frequently-accessed vertices cached),
and the hit rate of PaGraph's static top-out-degree policy (storage.py:97-104) on the same trace."""
import torch


def access_frequency(sampler, num_nodes=None, max_batches=None, layers=None):
    """freq[v] = number of NodeFlow rows that reference vertex v over one pass of `sampler`
    (count_vertex_freq, opt_cache_hit.py:22-24). `layers`: restrict to these NodeFlow layers.
    Also returns the number of vertices loaded (count_nf_vnum, count_vnum.py:16-20)."""
    n = num_nodes if num_nodes is not None else sampler.g.number_of_nodes()
    freq = torch.zeros(n, dtype=torch.int64, device=sampler.device)
    loaded = 0
    for b, nf in enumerate(sampler):
        if max_batches is not None and b >= max_batches:
            break
        for lid in range(nf.num_layers):
            if layers is not None and lid not in layers:
                continue
            ids = nf.layer_parent_nid(lid)
            ids = ids[ids >= 0]
            # `freq[ids] += 1` (opt_cache_hit.py:24) counts a vertex ONCE per layer however often the layer repeats
            # it (numpy fancy-index add does not accumulate); count_nf_vnum (count_vnum.py:19) counts every row
            freq[torch.unique(ids)] += 1
            loaded += int(ids.numel())
    return freq, loaded


def optimal_cache_hit(freq, cached):
    """opt_cache_hit.py:26-31: cache the int(V * cached) most frequently accessed vertices"""
    num = int(freq.numel() * cached)
    total = freq.sum()
    if num <= 0 or int(total) == 0:
        return 0.0
    top = torch.topk(freq, num).values.sum()
    return float(top) / float(total)


def degree_cache_hit(freq, out_degrees, cached):
    """hit rate of the reference policy: the int(V * cached) highest out-degree vertices
    (ties: lower id first), evaluated on the same access trace"""
    num = int(freq.numel() * cached)
    total = freq.sum()
    if num <= 0 or int(total) == 0:
        return 0.0
    order = torch.argsort(out_degrees.to(freq.device), descending=True, stable=True)[:num]
    return float(freq[order].sum()) / float(total)
