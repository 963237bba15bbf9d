import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import test_gpu_parity as T
import torch.nn.functional as Fn
from pagraph_amd.model import GCNSampling
from pagraph_amd.sampling import DeviceGraph, NeighborSampler
from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
from pagraph_amd.trainer import GraphedTrainer, cycle_batches
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
rng = np.random.default_rng(5)
V, Fdim, C, B = 5000, 64, 5, 500
adj = T._rand_csc(rng, V, 30000); g = DeviceGraph(adj)
feats = rng.standard_normal((V, Fdim)).astype(np.float32)
labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
train = np.arange(0, V, 2, dtype=np.int64)
res = {}
for mode in ("zerocopy", "async"):
    store = HostFeatureStore({"features": torch.from_numpy(feats)})
    c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode=mode)
    c.init_field(["features"]); c.auto_cache(g, ["features"], cache_ratio=0.4)
    torch.manual_seed(0)
    model = GCNSampling(Fdim, 16, C, 1, Fn.relu, 0.0).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=True)
    smp = NeighborSampler(g, B, 2, neighbor_type="in", shuffle=True, num_hops=2, seed_nodes=train, prefetch=True, seed=9, static=True)
    tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=model.required_inputs(3))
    log = []
    orig = tr.compute
    def comp(s, orig=orig, log=log, c=c, tr=tr):
        c.wait_misses(s.slot_index, tr.compute_stream)
        torch.cuda.synchronize()
        nm = s.nf_cur._node_mapping.tousertensor()
        n0 = s.nf_cur._layer_offsets[1]
        ids = nm[:n0]; valid = ids >= 0
        want = torch.from_numpy(feats).to(dev)[ids.clamp(min=0)]
        got = s.out["features"][:n0]
        bad = int(((got != want).any(1) & valid).sum())
        log.append((int(valid.sum()), bad, int(nm[nm >= 0].sum()), int(s.label.sum())))
        return orig(s)
    tr.compute = comp
    tr.run_steps(cycle_batches(smp, 8), 8); torch.cuda.synchronize()
    res[mode] = log
for i in range(8):
    print(i, "zerocopy", res["zerocopy"][i], "async", res["async"][i])
