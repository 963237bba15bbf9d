"""Synthetic datasets generated on the GPU (the GPU box has no dataset files).

Stands in for the reference's offline recipe (README.md:36-47):
  PaRMAT -noDuplicateEdges -undirected  ->  pp.txt
  PaGraph/data/preprocess.py --gen-feature --gen-label --gen-set:
    pp2adj        symmetrise every undirected edge           (preprocess.py:10-46)
    random_feature U[0,1) fp32 [V, F]                        (preprocess.py:50-63)
    random_label   randint(class_num)                        (preprocess.py:66-80)
    split_dataset  shuffled ids, 65 / 10 / 25                (preprocess.py:83-114)
The reference draws from unseeded numpy; here every stream is seeded.
"""
import math

import torch

from .. import _lib as L

SEED_GRAPH, SEED_FEAT, SEED_LABEL, SEED_SPLIT = 0x5EED0001, 0x5EED0002, 0x5EED0003, 0x5EED0004


def _q32(x):
    return int(round(x * 2 ** 32)) & 0xFFFFFFFF


def rmat_candidates(seed, scale, first, n, device, a=0.45, b=0.22, c=0.22):
    lib = L.load()
    src = torch.empty(n, dtype=torch.int64, device=device)
    dst = torch.empty(n, dtype=torch.int64, device=device)
    with torch.cuda.device(device):
        L.check(lib.pg_rmat_edges(seed, scale, _q32(a), _q32(b), _q32(c), first, n, L.ptr(src), L.ptr(dst),
                                  L.stream_ptr()), "pg_rmat_edges")
    return src, dst


def select_unique_undirected(src, dst, V, E):
    """The first E distinct undirected, loop-free, in-range edges in candidate order
    (sequential 'generate until E unique' semantics, evaluated with sorts). Returns
    (u, v) with u < v, or None when the candidates hold fewer than E such edges."""
    u = torch.minimum(src, dst)
    v = torch.maximum(src, dst)
    ok = (u != v) & (v < V)
    key = torch.where(ok, u * V + v, torch.full_like(u, -1))
    skey, order = torch.sort(key, stable=True)
    first = torch.ones_like(skey, dtype=torch.bool)
    first[1:] = skey[1:] != skey[:-1]
    first &= skey >= 0
    first_idx = order[first]
    if first_idx.numel() < E:
        return None
    if first_idx.numel() > E:
        # (the E-th smallest candidate index; a device sort takes tens of ms where torch.kthvalue took 2.5 s at 10^8 entries)
        thr = torch.sort(first_idx).values[E - 1]
        keep = first_idx <= thr
        keys = skey[first][keep]
    else:
        keys = skey[first]
    return keys // V, keys % V


def build_csc(u, v, V):
    """symmetrise (preprocess.py:36-38) and build the CSC (in-neighbour lists, ascending)"""
    src = torch.cat([u, v])
    dst = torch.cat([v, u])
    key = dst * V + src
    key, _ = torch.sort(key)
    indices = (key % V).to(torch.int32)
    counts = torch.bincount(key // V, minlength=V)
    indptr = torch.zeros(V + 1, dtype=torch.int64, device=u.device)
    indptr[1:] = torch.cumsum(counts, 0)
    return indptr, indices


def rmat_graph(V, E, seed=SEED_GRAPH, device="cuda", oversample=1.35):
    """symmetric RMAT graph with exactly E undirected edges -> (indptr int64 [V+1], indices int32 [2E])"""
    scale = max(1, math.ceil(math.log2(V)))
    factor = oversample
    while True:
        n = int(E * factor) + 1024
        src, dst = rmat_candidates(seed, scale, 0, n, device)
        sel = select_unique_undirected(src, dst, V, E)
        del src, dst
        if sel is not None:
            break
        factor *= 1.5
        if factor > 64:
            raise L.PgError("RMAT: cannot reach the requested number of distinct edges")
    return build_csc(sel[0], sel[1], V)


def fill_random_features(table, seed=SEED_FEAT, device="cuda", chunk_rows=1 << 18):
    """table: host fp32 [V, F] (ideally pinned): generated on the GPU in chunks, copied down."""
    lib = L.load()
    V, F = table.shape
    buf = torch.empty((min(chunk_rows, V), F), dtype=torch.float32, device=device)
    for lo in range(0, V, chunk_rows):
        hi = min(V, lo + chunk_rows)
        with torch.cuda.device(device):
            L.check(lib.pg_random_features(seed, lo, hi - lo, F, L.ptr(buf), buf.stride(0), L.stream_ptr()),
                    "pg_random_features")
        table[lo:hi].copy_(buf[:hi - lo])
    return table


def random_features_device(row_ids_or_count, F, seed=SEED_FEAT, device="cuda", row0=0):
    """fp32 [rows, F] for rows row0..row0+rows-1 directly in HBM"""
    lib = L.load()
    rows = int(row_ids_or_count)
    out = torch.empty((rows, F), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        L.check(lib.pg_random_features(seed, row0, rows, F, L.ptr(out), out.stride(0), L.stream_ptr()),
                "pg_random_features")
    return out


def random_labels(V, n_classes, seed=SEED_LABEL):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, n_classes, (V,), generator=g, dtype=torch.int64)


def split_dataset(V, seed=SEED_SPLIT):
    """(train_mask, val_mask, test_mask) int64 0/1, sizes int(0.65V), int(0.1V), rest"""
    g = torch.Generator().manual_seed(seed)
    nids = torch.randperm(V, generator=g)
    train_len, val_len = int(V * 0.65), int(V * 0.1)
    test_len = V - train_len - val_len
    masks = []
    for sel in (nids[:train_len], nids[train_len:train_len + val_len], nids[V - test_len:]):
        m = torch.zeros(V, dtype=torch.int64)
        m[sel] = 1
        masks.append(m)
    return tuple(masks)
