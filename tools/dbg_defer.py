import sys, numpy as np, torch, scipy.sparse as sp, faulthandler
sys.path.insert(0, "/root/repo"); faulthandler.enable()
import torch.nn.functional as Fn
from pagraph_amd.model import GCNSampling
from pagraph_amd.optim import Adam
from pagraph_amd.sampling import DeviceGraph, NeighborSampler
from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
from pagraph_amd.trainer import GraphedTrainer, cycle_batches
dev = torch.device("cuda", 0)
Fd, p, ratio, pre = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(1)
V, C, B = 20000, 41, 2000
src = rng.integers(0, V, 200000); dst = rng.integers(0, V, 200000)
adj = sp.csc_matrix((np.ones(200000, np.float32), (src, dst)), shape=(V, V)); adj.sum_duplicates(); adj.sort_indices()
g = DeviceGraph(adj)
feats = torch.from_numpy(rng.random((V, Fd), dtype=np.float32))
labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
c = GraphCacheServer(HostFeatureStore({"features": feats}), V, torch.arange(V), 0, miss_mode="async")
c.init_field(["features"]); c.auto_cache(g, ["features"], cache_ratio=ratio)
model = GCNSampling(Fd, 32, C, 1, Fn.relu, p).to(dev).train()
need = model.required_inputs(3)
if pre:   # an eager forward on the same model / cacher first, like the test
    s0 = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=np.arange(0, V, 2), seed=0)
    nf = next(iter(s0)); c.fetch_data(nf, need=need, slot=0, virtual=model.virtual_inputs(3)); c.wait_misses(0); y = model(nf); torch.cuda.synchronize()
smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=np.arange(0, V, 2), seed=1, static=True, defer_transpose=True)
tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), Adam(model.parameters(), lr=3e-2), c, smp, labels, dev, need=need, keep_losses=True)
tr.run_steps(cycle_batches(smp, 40), 24); tr.synchronize(); torch.cuda.synchronize()
print("ok", Fd, p, ratio, pre, float(tr.last_loss))
