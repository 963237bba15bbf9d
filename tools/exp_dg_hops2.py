"""time pg_dg_partition_mt (hops 2, P = 4) on the 10M/100M RMAT graph of the benchmark (host threads: all the process may use)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pagraph_amd.data import synthetic as syn
from pagraph_amd.partition.dg import dg_raw, default_threads
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
V, E = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
indptr, indices = syn.rmat_graph(V, E, device=dev)
train = torch.nonzero(syn.split_dataset(V)[0]).squeeze(1).numpy()
ip, ix = indptr.cpu().numpy(), indices.cpu().numpy()
for hops, thr in ((1, 1), (2, default_threads())):
    t0 = time.time()
    b, r, pv, rv = dg_raw(4, ip, ix, V, train, hops, threads=thr)
    print(f"dg P=4 hops={hops} threads={thr}: {time.time()-t0:.1f}s p_vnum={pv.tolist()} r_vnum={rv.tolist()}", flush=True)
