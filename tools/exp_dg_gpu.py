"""pg_dg_partition_gpu at BASELINE's sizes (round 6): time, statistics, and equality with the host code where that finishes
in about a minute. usage: exp_dg_gpu.py [10M|100M] [P] [hops] [--check]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pagraph_amd.data import synthetic as syn
import importlib
dgmod = importlib.import_module("pagraph_amd.partition.dg")

size = sys.argv[1] if len(sys.argv) > 1 else "10M"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
hops = int(sys.argv[3]) if len(sys.argv) > 3 else 2
check = "--check" in sys.argv
V, E = {"1M": (1_000_000, 10_000_000), "10M": (10_000_000, 100_000_000), "100M": (100_000_000, 1_000_000_000)}[size]
dev = torch.device("cuda", 0)
t0 = time.time()
indptr, indices = syn.rmat_graph(V, E, device=dev)
torch.cuda.synchronize()
train_mask, _, _ = syn.split_dataset(V)
train = torch.nonzero(train_mask).squeeze(1).numpy()
deg = (indptr[1:] - indptr[:-1]).double()
rec = {"graph": f"RMAT {V} / {E}", "P": P, "hops": hops, "train_vertices": int(len(train)), "graph_seconds": round(time.time() - t0, 1),
       "sum_deg": float(deg.sum().item()), "sum_deg_squared": float((deg * deg).sum().item())}
t0 = time.time()
a = dgmod.dg_raw(P, indptr, indices, V, train, hops, device="cuda", want_r_mask=False)
rec["gpu_seconds"] = round(time.time() - t0, 2)
rec["gpu_stats"] = dict(dgmod.LAST_GPU_STATS)
rec["p_vnum"], rec["r_vnum"] = a[2].tolist(), a[3].tolist()
if check:
    t0 = time.time()
    b = dgmod.dg_raw(P, indptr, indices, V, train, hops, device="cpu", want_r_mask=False)
    rec["host_seconds"] = round(time.time() - t0, 2)
    rec["identical_to_host_code"] = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]))
print(json.dumps(rec))
