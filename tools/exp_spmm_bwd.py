"""scatter-form backward aggregation: uniform sources vs the hub-heavy block a real NodeFlow has"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pagraph_amd import ops
from pagraph_amd.data import synthetic as syn
from pagraph_amd.sampling import DeviceGraph, NeighborSampler
dev = torch.device("cuda", 0)

def bench(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

V, E = 10_000_000, 100_000_000
ip, ix = syn.rmat_graph(V, E, seed=0x5EED0001, device=dev)
g = DeviceGraph.from_csc(ip, ix, V)
smp = NeighborSampler(g, 6000, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=torch.arange(0, V, 3)[:60000], prefetch=True, seed=1, transpose=None)
nf = next(iter(smp)); torch.cuda.synchronize()
ipb, srb = nf.blk_indptr[1], nf.blk_src[1]
n_dst, n_src = nf.layer_size(2), nf.layer_size(1)
cnt = torch.bincount(srb.long(), minlength=n_src)
print("block1: dst", n_dst, "src", n_src, "edges", srb.numel(), "max multiplicity", int(cnt.max()), "rows>32:", int((cnt > 32).sum()), "edges on rows>32:", int(cnt[cnt > 32].sum()))
lib = ops.L.load()
go = torch.rand((n_dst, 64), device=dev)
gh = torch.zeros((n_src, 64), device=dev)
def run(src):
    ops.L.check(ops.spmm_bwd_call(lib, go, gh, n_src, "mean", indptr=ipb, src=src), "bwd")
print("real block      : %.1f us" % bench(lambda: run(srb)))
uni = torch.randint(0, n_src, (srb.numel(),), device=dev, dtype=torch.int32)
print("uniform sources : %.1f us" % bench(lambda: run(uni)))
perm = torch.randperm(n_src, device=dev)[:srb.numel()].to(torch.int32) if n_src >= srb.numel() else uni
print("distinct sources: %.1f us" % bench(lambda: run(perm)))
print("zero fill       : %.1f us" % bench(lambda: gh.zero_()))
