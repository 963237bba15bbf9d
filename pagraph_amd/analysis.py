"""Cache-policy analysis — counterparts of examples/opt_cache_hit.py:16-31 and
examples/count_vnum.py:16-20: how often each vertex's features are requested in an epoch, the
hit rate an ORACLE cache of a given size would reach (most-frequently-accessed vertices cached),
and the hit rate of PaGraph's static top-out-degree policy (storage.py:97-104) on the same trace."""
import torch


def access_frequency(sampler, num_nodes=None, max_batches=None, layers=None, epochs=1):
    """freq[v] = number of NodeFlow rows that reference vertex v over one pass of `sampler`
    (count_vertex_freq, opt_cache_hit.py:22-24). `layers`: restrict to these NodeFlow layers.
    Also returns the number of vertices loaded (count_nf_vnum, count_vnum.py:16-20).
    `epochs` > 1 accumulates that many passes (each pass draws new neighbours): what the presample cache policy uses —
    one epoch's counts are mostly 0 / 1 / 2 and rank the vertices worse than their degree does."""
    n = num_nodes if num_nodes is not None else sampler.g.number_of_nodes()
    freq = torch.zeros(n, dtype=torch.int64, device=sampler.device)
    loaded = 0
    for _ in range(max(1, int(epochs))):
        for b, nf in enumerate(sampler):
            if max_batches is not None and b >= max_batches:
                break
            for lid in range(nf.num_layers):
                if layers is not None and lid not in layers:
                    continue
                ids = nf.layer_parent_nid(lid)
                ids = ids[ids >= 0]
                # `freq[ids] += 1` (opt_cache_hit.py:24) counts a vertex ONCE per layer however often the layer repeats
                # it (numpy fancy-index add does not accumulate); count_nf_vnum (count_vnum.py:19) counts every row.
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


def presample_order(freq, out_degrees):
    """the order auto_cache(policy='presample') caches in: most looked-up first, ties in the reference's degree order"""
    order = torch.argsort(out_degrees.to(freq.device), descending=True, stable=True)
    return order[torch.argsort(freq[order], descending=True, stable=True)]


def presample_cache_hit(freq, freq_presampled, out_degrees, cached):
    """hit rate on the trace `freq` of a cache filled from ANOTHER epoch's counts (`freq_presampled`: a different
    sampler seed) — what the presample policy delivers, as opposed to optimal_cache_hit's bound (which knows the trace)"""
    num = int(freq.numel() * cached)
    total = freq.sum()
    if num <= 0 or int(total) == 0:
        return 0.0
    order = presample_order(freq_presampled.to(freq.device), out_degrees)[:num]
    return float(freq[order].sum()) / float(total)
