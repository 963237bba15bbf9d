"""dev experiment: wall time of pg_dg_partition on the full 10M/100M RMAT graph"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pagraph_amd.data import synthetic as syn
from pagraph_amd.partition.dg import dg_raw
V, E = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
ip, ix = syn.rmat_graph(V, E, device=dev)
tm, _, _ = syn.split_dataset(V)
train = torch.nonzero(tm).squeeze(1).numpy()
iph, ixh = ip.cpu().numpy(), ix.cpu().numpy()
deg = np.diff(iph)
print("max deg", deg.max(), "sum deg^2 %.3e" % float((deg.astype(np.float64) ** 2).sum()), flush=True)
for hops, P in ((1, 4), (2, 2), (2, 8)):
    t0 = time.time()
    b, _, pv, rv = dg_raw(P, iph, ixh, V, train, hops)
    print(f"dg hops={hops} P={P}: {time.time()-t0:.1f}s p_vnum={pv.tolist()} r_vnum={rv.tolist()}", flush=True)
