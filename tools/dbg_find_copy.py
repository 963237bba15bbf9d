import os, sys
sys.path.insert(0, "/root/repo")
import torch
from pagraph_amd import ops
from pagraph_amd.data import synthetic as syn
from pagraph_amd.model import GraphSageSampling
from pagraph_amd.sampling import DeviceGraph, NeighborSampler
dev = torch.device("cuda", 0)
V, E, B, Fd, C = 30000, 300000, 1500, 64, 11
ip, ix = syn.rmat_graph(V, E, seed=8, device=dev)
g = DeviceGraph.from_csc(ip, ix, V)
smp = NeighborSampler(g, B, 2, neighbor_type='in', num_hops=2, seed_nodes=torch.arange(2 * B, device=dev), seed=1)
nf = next(iter(smp))
feats = syn.random_features_device(V, Fd, seed=2, device=dev)
labels = torch.randint(0, C, (nf.layer_size(-1),), device=dev)
n_valid = (labels != -100).sum().to(torch.int32).reshape(1)
model = GraphSageSampling(Fd, 16, C, 1, torch.relu, 0.2, 'mean').to(dev).train()
seed = torch.ones((), device=dev)
for it in range(2):
    for i in range(nf.num_layers):
        nf.layers[i].data.clear(); nf.layers[i].data['features'] = feats[nf.layer_parent_nid(i)]
    model.zero_grad(set_to_none=True)
    with torch.autograd.profiler.profile(use_cuda=True, with_stack=True) as prof:
        with ops.defer_partials() as reg:
            loss = model.forward_loss(nf, labels, n_valid, seed, -100)
            loss.backward(seed)
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=25, max_src_column_width=120))
