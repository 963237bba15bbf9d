import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pagraph_amd.data import synthetic as syn
from pagraph_amd.sampling import DeviceGraph
from pagraph_amd.partition.utils import closure_device
dev = torch.device("cuda", 0)
V, E = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
ip, ix = syn.rmat_graph(V, E, device=dev)
g = DeviceGraph.from_csc(ip, ix, V)
train_mask, _, _ = syn.split_dataset(V)
train = torch.nonzero(torch.as_tensor(train_mask)).squeeze(1)
sip, six, sub2full, subtrain = closure_device(g, train, 2)
Vs = sub2full.numel()
print("V_sub", Vs, "nnz_sub", six.numel(), "nnz_full", ix.numel(), "train", train.numel())
deg_full = (ip[1:] - ip[:-1])
deg_sub = (sip[1:] - sip[:-1])
s2f = sub2full.to(dev)
st = subtrain.to(dev)
df = deg_full[s2f[st]]; ds = deg_sub[st]
print("train vertices: deg equal:", bool((df == ds).all()), "mismatch count", int((df != ds).sum()), "zero-deg in full", int((df == 0).sum()), "zero-deg in sub", int((ds == 0).sum()))
bad = torch.nonzero(df != ds).squeeze(1)[:5]
for b in bad.tolist():
    v = int(st[b]); f = int(s2f[v])
    print(" sub", v, "full", f, "deg_sub", int(ds[b]), "deg_full", int(df[b]), "sub nbrs->full", s2f[six[sip[v]:sip[v+1]].long()].tolist()[:6], "full nbrs", ix[ip[f]:ip[f+1]].tolist()[:6])
# are the sub-graph's neighbour lists the full ones (mapped)?
cnt = torch.bincount(six.long(), minlength=Vs)
top = torch.topk(cnt, 3)
print("top out-degree in sub:", top.values.tolist(), "sub ids", top.indices.tolist(), "full ids", s2f[top.indices].tolist())
cf = torch.bincount(ix.long(), minlength=V)
print("their out-degree in full:", cf[s2f[top.indices]].tolist(), " top in full:", torch.topk(cf, 3).values.tolist())
print("subtrain", subtrain.numel(), "deg_full[train]==0:", int((deg_full[train.to(dev)] == 0).sum()))
tr_full_from_sub = s2f[st]
print("sub2full[subtrain] == train (sorted)?", bool(torch.equal(torch.sort(tr_full_from_sub).values, torch.sort(train.to(dev)).values)))
print("sub2full sorted ascending?", bool((s2f[1:] > s2f[:-1]).all()), "unique", int(torch.unique(s2f).numel()))
iso = train.to(dev)[deg_full[train.to(dev)] == 0][:5]
print("isolated train (full ids)", iso.tolist())
# where are they in the sub graph
pos = torch.searchsorted(s2f, iso)
print("pos", pos.tolist(), "s2f[pos]", s2f[pos.clamp(max=Vs-1)].tolist(), "deg_sub there", deg_sub[pos.clamp(max=Vs-1)].tolist())
