"""Where k_t_block spends its launch (the one-workgroup block transpose of the sampler chain): needs the library built with
-DPG_T_STAMPS (tools/exp_t_stamps.sh builds it beside the product library and puts it in its place ON THE GPU BOX ONLY).
Samples minibatches of the headline's shape (10M / 100M RMAT, B 6000, fan-out 2, 2 hops) and prints the phase stamps of
the last k_t_block<12> launch of each, in microseconds from the kernel's first instruction."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pagraph_amd import _lib as L
from pagraph_amd.data import synthetic as syn
from pagraph_amd.sampling import DeviceGraph, NeighborSampler

dev = torch.device("cuda", 0)
lib = L.load()
stamped = hasattr(lib, "pg_debug_t_stamps")      # (the product library: the same loop, for rocprofv3 --kernel-trace --stats)
if stamped:
    lib.pg_debug_t_stamps.restype = ctypes.c_int
    lib.pg_debug_t_stamps.argtypes = [ctypes.c_void_p]
V, E = 10_000_000, 100_000_000
indptr, indices = syn.rmat_graph(V, E, device=dev)
g = DeviceGraph.from_csc(indptr, indices, V, device=dev)
train_mask, _, _ = syn.split_dataset(V)
train = torch.nonzero(train_mask).squeeze(1).numpy()
smp = NeighborSampler(g, 6000, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True, seed=1,
                      static=True, defer_transpose=True)
names = ["start", "loads+zero", "dest+hist", "scan", "tptr out", "placement", "rank", "hubs", "tdst out"]
st = torch.cuda.Stream(device=dev)
rows = []
for k, nf in enumerate(smp):
    st.wait_event(nf._slot.ready)
    smp.transpose_blocks(nf, st)
    st.synchronize()
    if stamped:
        out = (ctypes.c_longlong * 16)()
        L.check(lib.pg_debug_t_stamps(out), "pg_debug_t_stamps")
        t = np.array(list(out)[:9], dtype=np.float64)
        rows.append((t - t[0]) / 100.0)
    smp.release(nf) if hasattr(smp, "release") else None
    if k >= 40:
        break
if not stamped:
    print("no stamps in this library (product build): ran %d minibatches" % (k + 1))
    sys.exit(0)
r = np.array(rows[5:])
print("phase ends, us from the first instruction (median over %d launches; 100 MHz clock):" % len(r))
med = np.median(r, axis=0)
for n, a, b in zip(names[1:], med[:-1], med[1:]):
    print("  %-12s %6.2f  (+%.2f)" % (n, b, b - a))
