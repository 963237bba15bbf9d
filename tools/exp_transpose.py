import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pagraph_amd.data import synthetic as syn
from pagraph_amd.sampling import DeviceGraph, NeighborSampler
dev = torch.device("cuda", 0)
V, E = 10_000_000, 100_000_000
ip, ix = syn.rmat_graph(V, E, seed=0x5EED0001, device=dev)
g = DeviceGraph.from_csc(ip, ix, V)
train_mask, _, _ = syn.split_dataset(V)
train = torch.nonzero(torch.as_tensor(train_mask)).squeeze(1)
print("train", train.numel())
for static, sd in ((False, 1), (True, 0)):
    smp = NeighborSampler(g, 6000, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True, seed=sd, static=static)
    it = iter(smp)
    for i in range(3):
        t0 = time.time(); nf = next(it); torch.cuda.synchronize(); dt = time.time() - t0
        sizes, edges = nf.actual_sizes() if static else ([nf.layer_size(i) for i in range(3)], [nf.block_size(i) for i in range(2)])
        tp = nf.blk_tptr[1].cpu().numpy()
        seg = np.diff(tp)
        srb = nf.blk_src[1][:edges[1]].cpu().numpy()
        ipb = nf.blk_indptr[1].cpu().numpy()[:sizes[2]+1]
        d = np.diff(ipb)
        hub = int(np.argmax(np.bincount(srb)))
        print('   dst deg hist', np.bincount(d).tolist(), 'hub local', hub, 'hub id', int(nf.layer_parent_nid(1)[hub]))
        cnt = np.bincount(srb, minlength=sizes[1])
        print(f"seed={sd} static={static} batch {i}: {dt*1e3:.2f} ms sizes {sizes} edges {edges} max seg {seg.max()} (bincount max {cnt.max()}) tptr[-1]={tp[-1]} len {len(tp)} rows>32 {(seg>32).sum()}")
