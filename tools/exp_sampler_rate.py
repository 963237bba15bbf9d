"""how fast can the sampler alone produce minibatches (hipGraph chain per ring slot)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pagraph_amd.data import synthetic as syn
from pagraph_amd.sampling import DeviceGraph, NeighborSampler
dev = torch.device("cuda", 0)
V, E = 10_000_000, 100_000_000
ip, ix = syn.rmat_graph(V, E, device=dev)
g = DeviceGraph.from_csc(ip, ix, V)
train_mask, _, _ = syn.split_dataset(V)
train = torch.nonzero(torch.as_tensor(train_mask)).squeeze(1)
for transpose in ('auto', None):
    smp = NeighborSampler(g, 6000, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True, seed=0,
                          static=True, transpose=transpose)
    smp.manual_release = True
    it = iter(smp)
    for _ in range(12):
        nf = next(it); smp.release(nf)
    torch.cuda.synchronize()
    t0 = time.time(); n = 400
    for _ in range(n):
        nf = next(it); smp.release(nf)
    torch.cuda.synchronize()
    print(f"transpose={transpose}: {(time.time() - t0) / n * 1e6:.1f} us per minibatch (sampler alone, ring {len(smp.slots)})")
