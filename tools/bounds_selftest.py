"""Self-test of the debug build that bounds-checks ids (PG_BOUNDS=1 -> libpagraph_hip_bounds.so; include/pagraph_hip.h
pg_bounds_*): a clean pipeline leaves no record; an id beyond the partition, a cache slot beyond the cache and a block edge
beyond the source layer are each named with kernel, site, value and bound — and none of them faults.
usage: PG_BOUNDS=1 python tools/bounds_selftest.py   (prints one JSON line; exit code 0 = as expected)"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pagraph_amd import _lib as L  # noqa: E402


def main():
    lib = L.load()
    assert L.BOUNDS, "not the debug build: set PG_BOUNDS=1 and build `make -C pagraph_amd/csrc bounds`"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = {}
    # ---- a clean training pipeline leaves no record ------------------------------------------------------------
    import scipy.sparse as spsp
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, cycle_batches
    rng = np.random.default_rng(3)
    V, Fd, C, B = 5000, 600, 5, 400
    s_, d_ = rng.integers(0, V, 30000), rng.integers(0, V, 30000)
    adj = spsp.coo_matrix((np.ones(60000, np.int8), (np.concatenate([s_, d_]), np.concatenate([d_, s_]))), shape=(V, V)).tocsr()
    adj.data[:] = 1
    g = DeviceGraph(adj)
    feats = torch.from_numpy(rng.standard_normal((V, Fd)).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    c = GraphCacheServer(HostFeatureStore({"features": feats}), V, torch.arange(V), 0, miss_mode="async")
    c.init_field(["features"])
    c.auto_cache(g, ["features"], cache_ratio=0.4)
    model = GCNSampling(Fd, 32, C, 1, Fn.relu, 0.2).to(dev)
    smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=np.arange(0, V, 2), prefetch=True,
                          seed=1, static=True, defer_transpose=True)
    tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), Adam(model.parameters(), lr=1e-2), c, smp, labels, dev,
                        need=model.required_inputs(3))
    tr.run_steps(cycle_batches(smp, 24), 24)
    tr.synchronize()
    out["clean_pipeline"] = L.bounds_report()
    assert out["clean_pipeline"] is None, out["clean_pipeline"]
    tr.close(); smp.close(); c.close()
    # ---- an id beyond the partition: k_split ---------------------------------------------------------------------
    n = 1000
    ids = torch.arange(n, device=dev, dtype=torch.int64)
    ids[123] = V + 5
    slot_map = torch.full((V,), -1, dtype=torch.int32, device=dev)
    nid_map = torch.arange(V, device=dev)
    mpos = torch.empty(n, dtype=torch.int32, device=dev)
    mfull = torch.empty(n, dtype=torch.int64, device=dev)
    mcnt = torch.zeros(1, dtype=torch.int32, device=dev)
    slots = torch.empty(n, dtype=torch.int32, device=dev)
    sp = L.stream_ptr()
    ml = L.miss_list(mpos, mfull, mcnt)
    L.check(lib.pg_split_rows(L.ptr(ids), n, L.ptr(slot_map), L.ptr(nid_map), ctypes.byref(ml), L.ptr(slots), None, None, sp))
    r = L.bounds_report()
    out["id_beyond_partition"] = r
    assert r and r["kernel"].startswith("k_split") and r["value"] == V + 5 and r["bound"] == V and r["offenders"] == 1, r
    # ---- a cache slot beyond the cache, a block edge beyond the source layer: k_spmm_fwd_rows ---------------------
    rows_cached, n_src, n_dst = 300, 500, 64
    cache = torch.rand((rows_cached, 608), device=dev)
    sl = torch.randint(0, rows_cached, (n_src,), dtype=torch.int32, device=dev)
    sl[77] = rows_cached + 9
    indptr = torch.arange(0, 2 * n_dst + 1, 2, dtype=torch.int32, device=dev)
    src = torch.randint(0, n_src, (2 * n_dst,), dtype=torch.int32, device=dev)
    src[5] = 77
    o = torch.empty((n_dst, 600), device=dev)
    L.note(sl), L.note(cache)
    rs = L.PgRowSource(sl.data_ptr(), cache.data_ptr(), 0, 608, 600, 0)
    L.check(lib.pg_spmm_fwd_rows(L.ptr(indptr), L.ptr(src), ctypes.byref(rs), n_dst, 600, 0, L.ptr(o), 600, None, None, 0, sp))
    r = L.bounds_report()
    out["slot_beyond_cache"] = r
    assert r and r["kernel"].startswith("k_spmm_fwd_rows") and r["value"] == rows_cached + 9 and r["bound"] == rows_cached, r
    sl[77] = 0
    src[9] = n_src + 1000
    L.check(lib.pg_spmm_fwd_rows(L.ptr(indptr), L.ptr(src), ctypes.byref(rs), n_dst, 600, 0, L.ptr(o), 600, None, None, 0, sp))
    r = L.bounds_report()
    out["edge_beyond_layer"] = r
    assert r and r["kernel"].startswith("k_spmm_fwd_rows") and r["value"] == n_src + 1000 and r["bound"] == n_src, r
    torch.cuda.synchronize()              # ... and nothing faulted
    print(json.dumps(out))


if __name__ == "__main__":
    main()
