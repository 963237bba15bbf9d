"""which stage of the all-virtual GraphSAGE forward differs from the materialised one?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as spsp, torch
import torch.nn.functional as Fn
from pagraph_amd import ops
from pagraph_amd.model import GraphSageSampling
from pagraph_amd.sampling import DeviceGraph, NeighborSampler
from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
dev = torch.device("cuda", 0)
rng = np.random.default_rng(12)
V, Fd, C, B, k = 5000, 600, 11, 1100, 2
w = 1.0 / np.arange(1, V + 1) ** 0.9; w /= w.sum()
s = rng.choice(V, 40000, p=w); d = rng.choice(V, 40000, p=w)
a = spsp.coo_matrix((np.ones(80000, np.int8), (np.concatenate([s, d]), np.concatenate([d, s]))), shape=(V, V)).tocsr(); a.data[:] = 1
g = DeviceGraph(a)
feats = rng.random((V, Fd), dtype=np.float32)
c = GraphCacheServer(HostFeatureStore({"features": torch.from_numpy(feats)}), V, torch.arange(V), 0, miss_mode="async")
c.init_field(["features"])
c.auto_cache(g, ["features"], cache_ratio=1.0 if len(sys.argv) < 2 else float(sys.argv[1]))
torch.manual_seed(5)
model = GraphSageSampling(Fd, 16, C, 1, Fn.relu, 0.3, 'mean').to(dev).train()
need = model.required_inputs(3); virt = model.virtual_inputs(3)
smp = NeighborSampler(g, B, k, neighbor_type='in', shuffle=False, num_hops=2, seed_nodes=np.arange(0, V, 2), seed=1)
nf = next(iter(smp))
frames = {}
for name, v in (("dense", None), ("virt", virt)):
    c.fetch_data(nf, need=need, slot=0, virtual=v); c.wait_misses(0); torch.cuda.synchronize()
    frames[name] = [dict(f) for f in nf._node_frames]
ids = [nf.layer_parent_nid(i).cpu().numpy() for i in range(3)]
for i in range(3):
    dn = frames["dense"][i]["features"]
    vr = frames["virt"][i]["features"]
    n = vr.shape[0]
    ip = torch.arange(n + 1, dtype=torch.int32, device=dev); sr = torch.arange(n, dtype=torch.int32, device=dev)
    mat = ops.aggregate_rows(ip, sr, vr, n, "sum")
    print(f"layer {i}: rows {n}; dense == table {np.array_equal(dn.cpu().numpy(), feats[ids[i]])}; virtual rows == dense {torch.equal(mat, dn)}")
L0 = model.layers[0]
for i in (1, 2):
    n = frames["dense"][i]["features"].shape[0]
    neigh = torch.from_numpy(rng.random((n, Fd), dtype=np.float32)).to(dev)
    ya = ops.linear2(frames["dense"][i]["features"], L0.fc_self, neigh, L0.fc_neigh, ops.ACT_CONCAT)
    yb = ops.linear2(frames["virt"][i]["features"], L0.fc_self, neigh, L0.fc_neigh, ops.ACT_CONCAT)
    print(f"linear2 on layer {i} ({n} rows): equal {torch.equal(ya, yb)} max diff {(ya - yb).abs().max().item():.3e}")
for blk in (0, 1):
    for dp in (None, ops.DropoutSpec(0.3, 1234, blk, model._drop_step)):
        aa = ops.block_aggregate(nf.blk_indptr[blk], nf.blk_src[blk], frames["dense"][blk]["features"], nf.layer_size(blk + 1), "mean", dropout=dp)
        ab = ops.block_aggregate(nf.blk_indptr[blk], nf.blk_src[blk], frames["virt"][blk]["features"], nf.layer_size(blk + 1), "mean", dropout=dp)
        print(f"aggregate block {blk} dropout {dp is not None}: equal {torch.equal(aa, ab)} max diff {(aa - ab).abs().max().item():.3e}")
print("---- whole model")
caps = {}
def mk(tag):
    def hook(mod, inp, out):
        caps.setdefault(tag, []).append((inp[0].data.get('neigh').detach().clone(), out['activation'].detach().clone()))
    return hook
model.layers[0].register_forward_hook(mk("L0")); model.layers[1].register_forward_hook(mk("L1"))
outs = {}
for name, v in (("dense", None), ("virt", virt)):
    caps.clear()
    model._drop_step.fill_(7)
    model.zero_grad(set_to_none=True)
    c.fetch_data(nf, need=need, slot=0, virtual=v); c.wait_misses(0)
    y = model(nf)
    torch.cuda.synchronize()
    outs[name] = (y.detach().clone(), {k: list(vv) for k, vv in caps.items()})
print("logits equal", torch.equal(outs["dense"][0], outs["virt"][0]), (outs["dense"][0] - outs["virt"][0]).abs().max().item())
for tag in ("L0", "L1"):
    for j, (a_, b_) in enumerate(zip(outs["dense"][1][tag], outs["virt"][1][tag])):
        print(tag, "call", j, "neigh equal", torch.equal(a_[0], b_[0]), "act equal", torch.equal(a_[1], b_[1]),
              "neigh maxdiff %.3e act maxdiff %.3e" % ((a_[0] - b_[0]).abs().max().item(), (a_[1] - b_[1]).abs().max().item()))
