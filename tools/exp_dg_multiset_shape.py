"""Where do the entries of dg's two-hop multisets come from? Share of sum(w) by the size of the vertex's multiset w(v) and by
the length of the adjacency list an entry is read from — what decides whether a range-major walk (a bitmap slice kept in LDS /
L2) would pay. usage: exp_dg_multiset_shape.py [10M|100M]"""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pagraph_amd.data import synthetic as syn

size = sys.argv[1] if len(sys.argv) > 1 else "10M"
V, E = {"1M": (1_000_000, 10_000_000), "10M": (10_000_000, 100_000_000), "100M": (100_000_000, 1_000_000_000)}[size]
dev = torch.device("cuda", 0)
indptr, indices = syn.rmat_graph(V, E, device=dev)
train_mask, _, _ = syn.split_dataset(V)
train = torch.nonzero(train_mask).squeeze(1).to(dev)
deg = indptr[1:] - indptr[:-1]
# w(v) = deg(v) + sum of deg(u) over u in in(v)
cs = torch.zeros(indices.numel() + 1, dtype=torch.int64, device=dev)
step = 1 << 28
for lo in range(0, indices.numel(), step):
    hi = min(indices.numel(), lo + step)
    cs[lo + 1:hi + 1] = deg[indices[lo:hi].long()]
cs = torch.cumsum(cs, 0)
w = (cs[indptr[1:]] - cs[indptr[:-1]] + deg)[train].double()
del cs
tot = float(w.sum())
lines = V / 8 / 128
rec = {"graph": f"RMAT {V} / {E}", "sum_w_train": tot, "mean_w": tot / train.numel(), "median_w": float(w.median()),
       "bitmap_lines": lines, "share_of_entries_by_w": {}, "share_of_entries_by_list_length": {}, "vertices_by_w": {}}
for thr in (1e3, 1e4, 1e5, 1e6, 1e7, 1e8):
    m = w >= thr
    rec["share_of_entries_by_w"][f">={thr:g}"] = round(float(w[m].sum()) / tot, 4)
    rec["vertices_by_w"][f">={thr:g}"] = int(m.sum())
# an upper bound on the lines a vertex's multiset touches: min(w, lines) -> HBM line fetches if each line were fetched once
rec["line_fetches_if_once_per_vertex_over_entries"] = round(float(torch.clamp(w, max=lines).sum()) / tot, 4)
d = deg.double()
# entries read from list u over all train vertices: deg(u) x (train vertices that have u as an in-neighbour) ~ deg(u)^2 x 0.65
d2 = d * d
for thr in (64, 256, 1024, 4096, 16384, 65536):
    rec["share_of_entries_by_list_length"][f">={thr}"] = round(float(d2[d >= thr].sum() / d2.sum()), 4)
rec["max_degree"] = int(deg.max())
print(json.dumps(rec))
