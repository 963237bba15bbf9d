import sys, numpy as np, torch, scipy.sparse as sp
sys.path.insert(0, "/root/repo")
import torch.nn.functional as Fn
from pagraph_amd import ops
from pagraph_amd.model import GraphSageSampling
from pagraph_amd.optim import Adam
from pagraph_amd.sampling import DeviceGraph, NeighborSampler
dev = torch.device("cuda", 0)
rng = np.random.default_rng(33)
V, Fd, C, B = 8000, 600, 41, 3000
src = rng.integers(0, V, 80000); dst = rng.integers(0, V, 80000)
adj = sp.csc_matrix((np.ones(80000, np.float32), (src, dst)), shape=(V, V)); adj.sum_duplicates(); adj.sort_indices()
g = DeviceGraph(adj)
feats = torch.from_numpy(rng.random((V, Fd), dtype=np.float32)).to(dev)
labels_all = torch.from_numpy(rng.integers(0, C, V)).to(dev)
smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=False, num_hops=2, seed_nodes=np.arange(0, V, 2), seed=2)
nf = next(iter(smp))
loss_fcn = ops.fused_loss(torch.nn.CrossEntropyLoss())
model = GraphSageSampling(Fd, 16, C, 1, Fn.relu, 0.2, 'mean').to(dev).train()
ids = nf._node_mapping.tousertensor(); o = nf._layer_offsets
print("layer sizes", [o[i + 1] - o[i] for i in range(nf.num_layers)])
for i in range(nf.num_layers):
    nf._node_frames[i] = {"features": feats[ids[o[i]:o[i + 1]]]}
lab = labels_all[ids[o[-2]:o[-1]]].contiguous()
def walk(fn, seen, out):
    if fn is None or fn in seen: return
    seen.add(fn); out.append(type(fn).__name__)
    for nxt, _ in fn.next_functions: walk(nxt, seen, out)
with ops.defer_partials() as reg:
    pred = model(nf)
    names = []; walk(pred.grad_fn, set(), names); print("graph:", names)
    loss = loss_fcn(pred, lab)
    loss.backward()
    print("by_param", len(reg.by_param), "second", len(reg.second), "conflict", reg.conflict)
print([None if p.grad is None else tuple(p.grad.shape) for p in model.parameters()])
