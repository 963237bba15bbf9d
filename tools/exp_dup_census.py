"""Duplicate census of the rows fetch_data looks up (north star: "LDS-staged index dedup"): how many of the rows of one
minibatch's NodeFlow are repeats — within a layer, across layers, among the misses only, and how many of the fused
kernel's edges re-read a source row that another edge of the same 4-destination block already read.
Default = the benchmark's workload (RMAT 10 M / 100 M, B = 6000, fan-out 2, 30 % hot-degree cache, 1naive)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pagraph_amd import _lib as L
from pagraph_amd.data import synthetic as syn
from pagraph_amd.partition.utils import closure_device
from pagraph_amd.sampling import DeviceGraph, NeighborSampler

ap = argparse.ArgumentParser()
ap.add_argument("--vertices", type=int, default=10_000_000)
ap.add_argument("--edges", type=int, default=100_000_000)
ap.add_argument("--batches", type=int, default=50)
ap.add_argument("--cache-ratio", type=float, default=0.30)
a = ap.parse_args()
L.load(); dev = torch.device("cuda", 0); torch.cuda.set_device(0)
indptr, indices = syn.rmat_graph(a.vertices, a.edges, device=dev)
train_mask, _, _ = syn.split_dataset(a.vertices)
g_full = DeviceGraph.from_csc(indptr, indices, a.vertices)
sub_indptr, sub_indices, sub2full, subtrain = closure_device(g_full, torch.nonzero(train_mask).squeeze(1), 2)
Vs = sub2full.numel()
g = DeviceGraph.from_csc(sub_indptr, sub_indices, Vs)
order = torch.argsort(g.out_degrees().to(torch.int64), descending=True, stable=True)
cached = torch.zeros(Vs, dtype=torch.bool, device=dev)
cached[order[:int(Vs * a.cache_ratio)]] = True
cached = cached.cpu().numpy()
sampler = NeighborSampler(g, 6000, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=subtrain, prefetch=True, seed=0)
acc = {}
def add(k, v):
    acc[k] = acc.get(k, 0.0) + float(v)
n = 0
for nf in sampler:
    ids = [nf.layer_parent_nid(i).cpu().numpy() for i in range(3)]
    allr = np.concatenate(ids)
    add("rows_all_layers", len(allr)); add("unique_all_layers", len(np.unique(allr)))
    for i, x in enumerate(ids):
        add(f"rows_layer{i}", len(x)); add(f"unique_layer{i}", len(np.unique(x)))
    miss = allr[~cached[allr]]
    add("miss_rows_all_layers", len(miss)); add("unique_miss_rows_all_layers", len(np.unique(miss)))
    m0 = ids[0][~cached[ids[0]]]
    add("miss_rows_layer0", len(m0)); add("unique_miss_rows_layer0", len(np.unique(m0)))
    # fused layer-0 aggregation: 4 destinations (waves) per 256-thread block; edges whose source another edge of the
    # same block already read = what an LDS dedup inside the block could save (L2 hits, not HBM)
    ip = nf.blk_indptr[0].cpu().numpy().astype(np.int64); src = nf.blk_src[0].cpu().numpy()
    n_dst = len(ids[1]); ip = ip[:n_dst + 1]; src = src[:ip[-1]]
    add("edges_block0", len(src)); add("unique_sources_block0", len(np.unique(src)))
    blk = np.repeat(np.arange(n_dst) // 4, np.diff(ip))
    key = blk * (1 << 32) + src
    add("edges_repeating_a_source_within_a_4_destination_block", len(key) - len(np.unique(key)))
    n += 1
    if n >= a.batches:
        break
out = {k: v / n for k, v in acc.items()}
out["batches"] = n
out["dup_frac_all_layers"] = 1 - out["unique_all_layers"] / out["rows_all_layers"]
out["dup_frac_misses_all_layers"] = 1 - out["unique_miss_rows_all_layers"] / max(1.0, out["miss_rows_all_layers"])
out["dup_frac_edges_block0_vs_sources"] = 1 - out["unique_sources_block0"] / out["edges_block0"]
out["dup_frac_within_4dst_block"] = out["edges_repeating_a_source_within_a_4_destination_block"] / out["edges_block0"]
print(json.dumps(out, indent=1))
