"""early layer-0 aggregation vs the in-step one over many steps (dropout off: must be bit-identical), then with dropout"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.sparse as spsp
import torch.nn.functional as Fn
from pagraph_amd.model import GCNSampling
from pagraph_amd.optim import Adam
from pagraph_amd.sampling import DeviceGraph, NeighborSampler
from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
from pagraph_amd.trainer import GraphedTrainer, cycle_batches
dev = torch.device("cuda", 0)
rng = np.random.default_rng(77)
V, Fdim, C, B, E = 60000, 600, 60, 6000, 600000
s_, d_ = rng.integers(0, V, E), rng.integers(0, V, E)
adj = spsp.coo_matrix((np.ones(2 * E, np.int8), (np.concatenate([s_, d_]), np.concatenate([d_, s_]))), shape=(V, V)).tocsc()
adj.sum_duplicates(); adj.sort_indices()
g = DeviceGraph(adj)
feats = rng.standard_normal((V, Fdim)).astype(np.float32)
labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
train = np.arange(0, V, 2, dtype=np.int64)
def run(early, p_drop, steps, prof=False):
    store = HostFeatureStore({"features": torch.from_numpy(feats)})
    c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode="async")
    c.init_field(["features"]); c.auto_cache(g, ["features"], cache_ratio=1.0)
    if prof:
        from pagraph_amd import _lib as L
        c.rows_prof = (torch.zeros(L.PG_PROF_WORDS * 4096, dtype=torch.int64, device=dev), 4096)
    torch.manual_seed(0)
    model = GCNSampling(Fdim, 32, C, 1, Fn.relu, p_drop).to(dev)
    opt = Adam(model.parameters(), lr=3e-3)
    smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True, seed=9, static=True, defer_transpose=True)
    tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=model.required_inputs(3), keep_losses=False)
    tr.early_aggregate = early
    tr.keep_primed = True
    out = []
    tr.on_step = lambda step, loss: out.append(loss.detach().clone())
    it = cycle_batches(smp, steps + 64)
    done = 0
    while done < steps:
        done += tr.run_steps(it, min(200, steps - done))
    tr.synchronize()
    return torch.stack(out).cpu().numpy()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for prof in (False, True):
    a = run("0", 0.0, steps, prof); b = run("1", 0.0, steps, prof)
    bad = np.flatnonzero(a != b)
    print(f"prof={prof} dropout 0: identical {np.array_equal(a, b)}; first mismatch {bad[:5]}, max loss {a.max():.3f} / {b.max():.3f}")
a = run("0", 0.2, steps); b = run("1", 0.2, steps)
print("dropout 0.2: in-step means", [round(float(x.mean()), 3) for x in np.array_split(a, 10)], "max", a.max())
print("dropout 0.2: early   means", [round(float(x.mean()), 3) for x in np.array_split(b, 10)], "max", b.max())
