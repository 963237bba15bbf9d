"""GPU (-m gpu): the HIP path, called through the C-ABI, against the oracle and the
golden vectors from the reference's own Python. Integer/byte/index work: bit exact.
Floating point (aggregation, model): tolerance 1e-4 (BASELINE.json north_star)."""
import ctypes
import glob
import os

import numpy as np
import pytest
import scipy.sparse as spsp
import torch

from _helpers import ROOT, free_port as _free_port, rand_csc as _rand_csc  # noqa: F401

pytestmark = pytest.mark.gpu
TOL = 1e-4


class FakeNF:
    """the NodeFlow surface fetch_data touches (storage.py:171-173,202)"""

    def __init__(self, layers, device):
        from pagraph_amd.sampling.nodeflow import _UserTensor
        cat = torch.from_numpy(np.concatenate([np.asarray(l, np.int64) for l in layers])).to(device)
        self._node_mapping = _UserTensor(cat)
        self._layer_offsets = [0]
        for l in layers:
            self._layer_offsets.append(self._layer_offsets[-1] + len(l))
        self.num_layers = len(layers)
        self._node_frames = [None] * len(layers)


def _cacher(z, dev, mode="staged"):
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    store = HostFeatureStore({"features": torch.from_numpy(z["features_table"]), "norm": torch.from_numpy(z["norm_table"])})
    c = GraphCacheServer(store, len(z["nid_map"]), torch.from_numpy(z["nid_map"]), 0, miss_mode=mode)
    c.init_field(["features", "norm"])
    return c


@pytest.mark.parametrize("mode", ["staged", "zerocopy", "async", "async-split", "async-device-only"])
@pytest.mark.parametrize("F", [8, 600, 602])
def test_fetch_data_vs_reference_golden(dev, hiplib, golden_dir, F, mode):
    """G1/G2: GraphCacheServer.fetch_data == the reference's outputs, bit for bit (every miss path; 'async-split'
    = the worker thread moves the head of each miss list, the device reads the tail over PCIe)"""
    z = np.load(os.path.join(golden_dir, f"g1_fetch_data_F{F}.npz"))
    c = _cacher(z, dev, "async" if mode.startswith("async") else mode)
    c.cpu_share = {"async-split": 0.4, "async-device-only": 0.0}.get(mode, 1.0)
    c.log = True
    nids = torch.from_numpy(z["cached_nids"]).to(dev)
    c.cache_fix_data(nids, c.get_feat_from_server(nids, ["features", "norm"], to_gpu=True), is_full=False)
    assert np.array_equal(c.localid2cacheid.cpu().numpy(), z["state_localid2cacheid"])
    assert np.array_equal(c.gpu_flag.cpu().numpy(), z["state_gpu_flag"])
    assert c.cached_num == int(z["state_cached_num"])
    layers = [z[f"layer{i}_nids"] for i in range(int(z["num_layers"]))]
    for rep in range(3):                                  # async: the slots are reused
        nf = FakeNF(layers, dev)
        c.fetch_data(nf, slot=rep % 2)
        c.wait_misses(rep % 2)
        torch.cuda.synchronize()
        for i in range(len(layers)):
            for name in ("features", "norm"):
                got = nf._node_frames[i][name].cpu().numpy()
                assert got.shape == z[f"layer{i}_{name}"].shape
                assert np.array_equal(got, z[f"layer{i}_{name}"]), (i, name)
        assert c.get_miss_rate() == float(z["miss_rate"])


@pytest.mark.parametrize("F", [8, 600, 602])
def test_fetch_from_cache_vs_reference_golden(dev, hiplib, golden_dir, F):
    """G3: the full-cache path"""
    z = np.load(os.path.join(golden_dir, f"g3_fetch_from_cache_F{F}.npz"))
    c = _cacher(z, dev)
    full = torch.arange(len(z["nid_map"]), device=dev)
    c.cache_fix_data(full, c.get_feat_from_server(full, ["features", "norm"], to_gpu=True), is_full=True)
    layers = [z[f"layer{i}_nids"] for i in range(int(z["num_layers"]))]
    nf = FakeNF(layers, dev)
    c.fetch_data(nf)
    for i in range(len(layers)):
        for name in ("features", "norm"):
            assert np.array_equal(nf._node_frames[i][name].cpu().numpy(), z[f"layer{i}_{name}"])


@pytest.mark.parametrize("tag", ["partial", "full"])
def test_auto_cache_vs_reference_golden(dev, hiplib, golden_dir, tag, monkeypatch):
    """G5: capability rule + top-out-degree selection (storage.py:78-104)"""
    import types
    z = np.load(os.path.join(golden_dir, f"g5_auto_cache_{tag}.npz"))
    c = _cacher(z, dev)
    monkeypatch.setattr(torch.cuda, "max_memory_allocated", lambda device=None: int(z["peak_allocated"]))
    monkeypatch.setattr(torch.cuda, "max_memory_reserved", lambda device=None: int(z["peak_cached"]))
    monkeypatch.setattr(torch.cuda, "get_device_properties",
                        lambda d: types.SimpleNamespace(total_memory=int(z["total_memory"])))
    g = types.SimpleNamespace(out_degrees=lambda: torch.from_numpy(z["out_degrees"]))
    c.auto_cache(g, ["features", "norm"])
    assert c.capability == int(z["capability"]) and c.cached_num == int(z["cached_num"])
    assert c.full_cached == bool(z["full_cached"])
    assert np.array_equal(c.gpu_flag.cpu().numpy(), z["gpu_flag"])
    assert np.array_equal(c.localid2cacheid.cpu().numpy(), z["localid2cacheid"])
    assert np.array_equal(c.gpu_fix_cache["features"].cpu().numpy(), z["cache_features"])
    assert np.array_equal(c.gpu_fix_cache["norm"].cpu().numpy(), z["cache_norm"])


def test_auto_cache_presample_policy(dev, hiplib, golden_dir, monkeypatch):
    """auto_cache(policy='presample'): the `capability` most looked-up vertices, ties in the reference's degree order; fetches
    stay bit-exact; the default policy is untouched (the G5 test above); on its own trace it reaches opt_cache_hit.py's bound"""
    import types
    from pagraph_amd import analysis
    z = np.load(os.path.join(golden_dir, "g5_auto_cache_partial.npz"))
    c = _cacher(z, dev)
    monkeypatch.setattr(torch.cuda, "max_memory_allocated", lambda device=None: int(z["peak_allocated"]))
    monkeypatch.setattr(torch.cuda, "max_memory_reserved", lambda device=None: int(z["peak_cached"]))
    monkeypatch.setattr(torch.cuda, "get_device_properties",
                        lambda d: types.SimpleNamespace(total_memory=int(z["total_memory"])))
    deg = z["out_degrees"]
    V = len(deg)
    g = types.SimpleNamespace(out_degrees=lambda: torch.from_numpy(deg))
    rng = np.random.default_rng(5)
    freq = np.zeros(V, np.int64)
    seen = rng.choice(V, V // 3, replace=False)
    freq[seen] = rng.integers(1, 6, len(seen))               # many ties, two thirds never seen
    with pytest.raises(ValueError):
        c.auto_cache(g, ["features", "norm"], policy="presample")
    c.auto_cache(g, ["features", "norm"], policy="presample", freq=torch.from_numpy(freq))
    cap = int(z["capability"])
    assert c.cached_num == cap and not c.full_cached
    by_deg = np.argsort(-deg, kind="stable")
    want = by_deg[np.argsort(-freq[by_deg], kind="stable")][:cap]
    l2c = c.localid2cacheid.cpu().numpy()
    assert np.array_equal(np.flatnonzero(c.gpu_flag.cpu().numpy()), np.sort(want))
    assert np.array_equal(l2c[want], np.arange(cap))
    assert np.array_equal(c.gpu_fix_cache["features"].cpu().numpy(), z["features_table"][z["nid_map"][want]])
    ids = rng.integers(0, V, 500).astype(np.int64)
    nf = FakeNF([ids], dev)
    c.fetch_data(nf)
    torch.cuda.synchronize()
    assert np.array_equal(nf._node_frames[0]["features"].cpu().numpy(), z["features_table"][z["nid_map"][ids]])
    f = torch.from_numpy(freq).to(dev)
    d = torch.from_numpy(deg).to(dev)
    ratio = cap / V
    assert abs(analysis.presample_cache_hit(f, f, d, ratio) - analysis.optimal_cache_hit(f, ratio)) < 1e-12
    assert analysis.presample_cache_hit(f, f, d, ratio) >= analysis.degree_cache_hit(f, d, ratio)


@pytest.mark.parametrize("n", [1, 63, 1000, 6000, 16384, 16385, 50000])
def test_gather_labels_and_valid_count(dev, hiplib, n):
    """pg_gather_labels == labels[batch_nids] (pa_gcn.py:99-100) with the padding ids of a fixed-shape batch mapped to the
    loss's ignore_index, + the number of rows the loss counts — written, not accumulated: the count word holds garbage before
    the call (one workgroup up to 16 K ids, the multi-block kernel behind a zero fill above)"""
    from pagraph_amd import _lib as L
    rng = np.random.default_rng(n)
    V = 5000
    labels = torch.from_numpy(rng.integers(0, 41, V)).to(dev)
    ids = rng.integers(0, V, n)
    ids[rng.random(n) < 0.2] = -1                                 # padding
    d_ids = torch.from_numpy(ids).to(dev)
    out = torch.full((n,), 7777, dtype=torch.int64, device=dev)
    cnt = torch.full((1,), 123456, dtype=torch.int32, device=dev)
    for _ in range(2):                                            # the second call must not add to the first one's count
        L.check(hiplib.pg_gather_labels(L.ptr(d_ids), n, L.ptr(labels), V, -100, L.ptr(out), L.ptr(cnt), L.stream_ptr()))
    torch.cuda.synchronize()
    want = np.where(ids >= 0, labels.cpu().numpy()[np.maximum(ids, 0)], -100)
    assert np.array_equal(out.cpu().numpy(), want)
    assert int(cnt.item()) == int((ids >= 0).sum())
    # the self-cleaning variant (no zero fill in front): same outputs, call after call, its two scratch words back at zero
    out2 = torch.full((n,), 7777, dtype=torch.int64, device=dev)
    cnt2 = torch.full((1,), 123456, dtype=torch.int32, device=dev)
    scr = torch.zeros(2, dtype=torch.int32, device=dev)
    for _ in range(3):
        L.check(hiplib.pg_gather_labels_sc(L.ptr(d_ids), n, L.ptr(labels), V, -100, L.ptr(out2), L.ptr(cnt2), L.ptr(scr),
                                           L.stream_ptr()))
        torch.cuda.synchronize()
        assert int(cnt2.item()) == int((ids >= 0).sum()) and scr.tolist() == [0, 0]
    assert np.array_equal(out2.cpu().numpy(), want)


@pytest.mark.parametrize("n,F,ratio", [(1, 600, 0.5), (63, 600, 0.0), (64, 602, 1.0), (65, 600, 0.3), (4097, 128, 0.3),
                                        (50000, 600, 0.3), (600000, 64, 0.7), (1000, 7, 0.5), (1000, 33, 0.5)])
def test_gather_vs_oracle_random(dev, hiplib, oracle, n, F, ratio):
    """pg_gather_rows vs the C oracle on seeded ids: ragged tails, dims that force the
    dwordx4 / dwordx2 / dword / lane-per-row paths, empty cache, both launch shapes"""
    from pagraph_amd import _lib as L
    rng = np.random.default_rng(n + F)
    V, N = 3000, 5000
    table = rng.random((N, F), dtype=np.float32)
    nid_map = np.sort(rng.choice(N, V, replace=False)).astype(np.int64)
    st = oracle.CacheState(V, nid_map)
    cached = rng.permutation(V)[:int(V * ratio)].astype(np.int64)
    st.cache_fix_data(cached, {"f": table}, ratio == 1.0)
    ids = rng.integers(0, V, n).astype(np.int64)
    want = st.fetch_layer(ids, {"f": table})["f"]
    # device side through the raw C-ABI
    d_ids = torch.from_numpy(ids).to(dev)
    slot = torch.empty(V, dtype=torch.int32, device=dev)
    sp = L.stream_ptr()
    L.check(hiplib.pg_slot_map_reset(L.ptr(slot), V, sp))
    d_cached = torch.from_numpy(cached).to(dev)
    L.check(hiplib.pg_slot_map_assign(L.ptr(slot), L.ptr(d_cached), len(cached), sp))
    cache = torch.from_numpy(st.cache["f"]).to(dev) if len(cached) else None
    out = torch.full((n, F), -1.0, device=dev)
    mpos = torch.empty(n, dtype=torch.int32, device=dev)
    mfull = torch.empty(n, dtype=torch.int64, device=dev)
    mcnt = torch.zeros(1, dtype=torch.int32, device=dev)
    scratch = torch.empty(n, dtype=torch.int32, device=dev) if n % 2 else None   # both code paths
    stats = torch.tensor([5, 7], dtype=torch.int64, device=dev)                    # accumulated, not reset
    fields, nf = L.make_fields([(cache, out, F, F, F)])
    ml = L.miss_list(mpos, mfull, mcnt)
    L.check(hiplib.pg_gather_rows(L.ptr(d_ids), n, L.ptr(slot), L.ptr(torch.from_numpy(nid_map).to(dev)), fields, nf,
                                  ctypes.byref(ml), L.ptr(scratch), L.ptr(stats), None, None, sp))
    m = int(mcnt.item())
    assert m == st.miss_num
    assert stats.tolist() == [5 + n, 7 + m]
    pos = mpos[:m].cpu().numpy(); full = mfull[:m].cpu().numpy()
    hit = st.gpu_flag[ids].astype(bool)
    assert np.array_equal(np.sort(pos), np.nonzero(~hit)[0])             # exactly the missing rows
    assert np.array_equal(full, nid_map[ids[pos]])                       # with their full-graph ids
    got = out.cpu().numpy()
    assert np.array_equal(got[hit], want[hit])
    assert np.all(got[~hit] == -1.0)                                     # misses untouched by the gather
    # finish the misses with the scatter kernel
    staged = torch.from_numpy(table[full]).to(dev)
    L.check(hiplib.pg_scatter_rows(L.ptr(staged), L.ptr(mpos), m, None, F, L.ptr(out), F, sp))
    assert np.array_equal(out.cpu().numpy(), want)


def test_miss_gather_stragglers_are_rescued(dev, hiplib, monkeypatch):
    """the CPU row gather of the async miss path survives pool threads that lose their CPU with a claimed chunk in hand
    (here: every third chunk a pool thread claims sleeps 4 ms first — PG_MISSQ_TEST_STALL): the worker re-executes the
    overdue chunks, the rows are bit-exact, and a job takes a fraction of one stall instead of several in a row"""
    import time
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    monkeypatch.setenv("PG_MISSQ_TEST_STALL", "3,4000")
    rng = np.random.default_rng(11)
    V, Fd, n = 60000, 600, 6000
    table = torch.from_numpy(rng.random((V, Fd), dtype=np.float32))
    c = GraphCacheServer(HostFeatureStore({"features": table}), V, torch.arange(V), 0, miss_mode="async", host_threads=6)
    c.init_field(["features"])
    c.missq_slots = 3
    took = []
    for rep in range(9):
        ids = rng.choice(V, n, replace=False).astype(np.int64)
        nf = FakeNF([ids], dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        c.fetch_data(nf, slot=rep % 3)
        c.wait_misses(rep % 3)
        torch.cuda.synchronize()
        took.append(time.perf_counter() - t0)
        assert np.array_equal(nf._node_frames[0]["features"].cpu().numpy(), table.numpy()[ids])
    st = c.miss_queue_stats()
    assert st["rescued_chunks"] > 0, st
    # 188 chunks per job, a third of those the five pool threads claim would sleep 4 ms each, one after the other per thread:
    # tens of milliseconds without the rescue
    assert np.median(took) < 0.012, took
    c.shutdown_miss_queue()


@pytest.mark.parametrize("V,E,B,k,hops", [(2000, 12000, 256, 2, 2), (5000, 60000, 1000, 2, 2), (800, 9000, 100, 5, 3),
                                           (3000, 20000, 333, 1, 1), (4000, 50000, 512, 64, 1), (100000, 900000, 6000, 2, 2),
                                           # 64 lanes per destination and more than 1024 * 4 of them: two destinations per group
                                           (9000, 700000, 5000, 40, 1),
                                           # a two-word bitmap, five seeds per batch, three hops: every launch is one block
                                           (70, 300, 5, 2, 3),
                                           # fan-out above one wave (k_sample_wide; the reference takes any --num-neighbors)
                                           (3000, 400000, 128, 65, 1), (3000, 500000, 64, 100, 2), (2500, 900000, 50, 300, 1)])
def test_sampler_vs_oracle_bit_exact(dev, hiplib, oracle, V, E, B, k, hops):
    """NodeFlow node-id sets, layer offsets and block CSRs equal the CPU restatement under a fixed seed"""
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    rng = np.random.default_rng(V + k)
    adj = _rand_csc(rng, V, E)
    g = DeviceGraph(adj)
    train = np.sort(rng.choice(V, int(V * 0.65), replace=False)).astype(np.int64)
    smp = NeighborSampler(g, B, k, neighbor_type='in', shuffle=True, num_hops=hops, seed_nodes=train, prefetch=True, seed=42)
    seeds_order = smp.seeds.cpu().numpy()
    assert np.array_equal(np.sort(seeds_order), train)
    csc = spsp.csc_matrix(adj); csc.sort_indices()
    for epoch in range(2):
        nb = 0
        for b, nf in enumerate(smp):
            if b > 3 and b < len(smp) - 1:
                continue                                  # first batches + the short last batch
            s = seeds_order[b * B:(b + 1) * B]
            ref = oracle.sample_nodeflow(csc.indptr, csc.indices, s, k, hops, 42, epoch, b)
            assert nf._layer_offsets == [int(x) for x in ref["layer_offsets"][:hops + 2]]
            assert np.array_equal(nf._node_mapping.tousertensor().cpu().numpy(), ref["node_mapping"])
            for i in range(hops):
                assert np.array_equal(nf.blk_indptr[i].cpu().numpy(), ref["blocks"][i][0])
                assert np.array_equal(nf.blk_src[i].cpu().numpy(), ref["blocks"][i][1])
            nb += 1
        assert nb >= 1 and b == len(smp) - 1


@pytest.mark.parametrize("clear_by_ids", [False, True])
@pytest.mark.parametrize("V,E,B,k,hops,lookback", [(70000, 400000, 700, 2, 2, 3), (70000, 400000, 300, 3, 3, 2),
                                                    (5000, 60000, 900, 33, 1, 7), (140000, 300000, 2000, 2, 2, 1)])
def test_sampler_multi_round_lookback_paths(dev, hiplib, oracle, monkeypatch, V, E, B, k, hops, lookback, clear_by_ids):
    """the five-launch chain's decoupled look-backs with the per-launch block limit shrunk (PG_SAMPLER_LOOKBACK, read at
    sampler creation): a group samples several destinations (iters > 1) and a rank block makes several rounds over its
    bitmap words (m > 1) — the shapes a 2^31-vertex graph or a 64-wide fan-out over a large batch takes with the real
    limit of 1024 blocks. Static (fixed-shape, -1 padding) and DGL layouts, bit-exact vs the oracle.
    clear_by_ids: the rank launch clears the other bitmap through the previous layer's id list instead of word by word (what
    samplers over more than 64 M vertices do: PG_SAMPLER_CLEAR_IDS_ABOVE) — every batch of the pass is checked then, a mark
    left behind would show up as a foreign vertex in a later batch's layers."""
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    monkeypatch.setenv("PG_SAMPLER_LOOKBACK", str(lookback))
    monkeypatch.setenv("PG_SAMPLER_CLEAR_IDS_ABOVE", "0" if clear_by_ids else str(1 << 40))
    rng = np.random.default_rng(V + k + lookback)
    adj = _rand_csc(rng, V, E)
    g = DeviceGraph(adj)
    train = np.sort(rng.choice(V, int(V * 0.3), replace=False)).astype(np.int64)
    csc = spsp.csc_matrix(adj); csc.sort_indices()
    for static in (False, True):
        smp = NeighborSampler(g, B, k, neighbor_type='in', shuffle=False, num_hops=hops, seed_nodes=train, seed=9,
                              static=static)
        for b, nf in enumerate(smp):
            if b > 2 and b < len(smp) - 1 and not (clear_by_ids and b < 12):
                continue
            ref = oracle.sample_nodeflow(csc.indptr, csc.indices, train[b * B:(b + 1) * B], k, hops, 9, 0, b)
            torch.cuda.synchronize()
            nm = nf._node_mapping.tousertensor().cpu().numpy()
            if not static:
                assert nf._layer_offsets == [int(x) for x in ref["layer_offsets"][:hops + 2]]
                assert np.array_equal(nm, ref["node_mapping"])
            offs = ref["layer_offsets"]
            for l in range(hops + 1):
                want = ref["node_mapping"][offs[l]:offs[l + 1]]
                o0, o1 = nf._layer_offsets[l], nf._layer_offsets[l + 1]
                assert np.array_equal(nm[o0:o0 + len(want)], want)
                assert (nm[o0 + len(want):o1] == -1).all()
            for i in range(hops):
                ip, sr = ref["blocks"][i]
                got_ip = nf.blk_indptr[i].cpu().numpy()
                assert np.array_equal(got_ip[:len(ip)], ip) and (got_ip[len(ip):] == ip[-1]).all()
                assert np.array_equal(nf.blk_src[i].cpu().numpy()[:len(sr)], sr)
        smp.check()                                       # no look-back poll gave up (pg_sampler_status)


@pytest.mark.timeout(600)
def test_graph_offsets_beyond_2_to_31(dev, hiplib, oracle):
    """the partition's CSC has 64-bit offsets on every path that walks it (scipy's own int64 CSR/CSC,
    PaGraph/partition/utils.py:36-45): a graph whose in-lists start past entry 2^31 samples, expands and preprocesses
    exactly like the same lists at the front of the array. Vertex 0 owns a 2^31-entry dummy in-list that nothing ever
    visits (no vertex lists it as a neighbour, it is not a seed), so the oracle can run on the compact graph."""
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    free, _ = torch.cuda.mem_get_info()
    if free < 24 << 30:
        pytest.skip("needs ~10 GB of device memory")
    rng = np.random.default_rng(31)
    V, E, B, k, hops = 3000, 40000, 200, 3, 2
    adj = _rand_csc(rng, V, E).tolil()
    adj[0, :] = 0; adj[:, 0] = 0                    # vertex 0: isolated in the real graph
    csc = spsp.csc_matrix(adj.tocsr()); csc.eliminate_zeros(); csc.sort_indices()
    big = (1 << 31) + 12345
    indptr = torch.from_numpy(csc.indptr.astype(np.int64))
    indptr[1:] += big                               # vertex 0's "in-list" = the first `big` entries
    indices = torch.empty(big + csc.nnz, dtype=torch.int32, device=dev)
    indices[:4096].fill_(1)
    indices[big:] = torch.from_numpy(csc.indices.astype(np.int32)).to(dev)
    g = DeviceGraph.from_csc(indptr.to(dev), indices, V)
    assert int(g.indptr[-1]) > 2 ** 31
    train = np.sort(rng.choice(np.arange(1, V), 1500, replace=False)).astype(np.int64)
    smp = NeighborSampler(g, B, k, neighbor_type='in', shuffle=False, num_hops=hops, seed_nodes=train, seed=7)
    for b, nf in enumerate(smp):
        ref = oracle.sample_nodeflow(csc.indptr, csc.indices, train[b * B:(b + 1) * B], k, hops, 7, 0, b)
        assert np.array_equal(nf._node_mapping.tousertensor().cpu().numpy(), ref["node_mapping"])
        for i in range(hops):
            assert np.array_equal(nf.blk_indptr[i].cpu().numpy(), ref["blocks"][i][0])
            assert np.array_equal(nf.blk_src[i].cpu().numpy(), ref["blocks"][i][1])
        if b >= 2:
            break
    # the L-hop closure's frontier expansion over the same arrays
    from pagraph_amd import _lib as L
    bm = torch.zeros((V + 63) // 64, dtype=torch.int64, device=dev)
    fr = torch.from_numpy(train[:300]).to(dev)
    L.check(hiplib.pg_frontier_mark_neighbors(L.ptr(g.indptr), L.ptr(g.indices), L.ptr(fr), fr.numel(), L.ptr(bm), 0,
                                              L.stream_ptr()))
    got = np.unpackbits(bm.cpu().numpy().view(np.uint8), bitorder="little")[:V].nonzero()[0]
    want = np.unique(np.concatenate([csc.indices[csc.indptr[v]:csc.indptr[v + 1]] for v in train[:300]]))
    assert np.array_equal(got, want)
    # pa_server.py:45-52's one-hop preprocessing on the same offsets
    from pagraph_amd.server import preprocess_features

    class _BigCSC:                                   # the attributes preprocess_features reads
        nnz = big + csc.nnz
        shape = (V, V)
    bc = _BigCSC()
    bc.indptr, bc.indices = csc.indptr.astype(np.int64) + big, indices      # every list past 2^31; vertex 0's is empty here
    feats = torch.from_numpy(rng.random((V, 24), dtype=np.float32))
    norm = torch.ones((V, 1))
    got = preprocess_features(bc, feats, norm, chunk_rows=1000).numpy()
    want = (csc.T @ feats.numpy().astype(np.float64)).astype(np.float32)
    assert np.allclose(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("n_dst,n_src,deg,dim,reduce", [(500, 900, 2, 600, "mean"), (6000, 12000, 2, 64, "mean"),
                                                         (100, 50, 7, 602, "sum"), (37, 80, 3, 33, "mean"),
                                                         (1000, 1000, 0, 64, "mean"), (300, 400, 4, 1200, "sum")])
def test_spmm_fwd_bwd_vs_oracle(dev, hiplib, oracle, n_dst, n_src, deg, dim, reduce):
    from pagraph_amd.ops import block_aggregate
    rng = np.random.default_rng(n_dst + dim)
    cnt = rng.integers(0, deg + 1, n_dst) if deg else np.zeros(n_dst, np.int64)
    indptr = np.zeros(n_dst + 1, np.int32); indptr[1:] = np.cumsum(cnt)
    src = rng.integers(0, n_src, int(indptr[-1])).astype(np.int32)
    h = rng.standard_normal((n_src, dim)).astype(np.float32)
    want = oracle.spmm_fwd(indptr, src, h, n_dst, reduce)
    th = torch.from_numpy(h).to(dev).requires_grad_(True)
    out = block_aggregate(torch.from_numpy(indptr).to(dev), torch.from_numpy(src).to(dev), th, n_dst, reduce)
    got = out.detach().cpu().numpy()
    assert np.array_equal(got, want)            # same summation order as the sequential oracle
    go = rng.standard_normal((n_dst, dim)).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    want_g = oracle.spmm_bwd(indptr, src, go, n_src, reduce)
    assert np.allclose(th.grad.cpu().numpy(), want_g, rtol=0, atol=TOL)   # atomics: order differs


@pytest.mark.parametrize("arch", ["gcn", "gcn_pre", "sage", "infer"])
def test_model_forward_vs_oracle(dev, hiplib, oracle, arch):
    """layer outputs within 1e-4 of the CPU restatement (dropout off)"""
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNInfer, GCNSampling, GraphSageSampling
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    rng = np.random.default_rng(9)
    V, Fdim, C, B, k = 3000, 600, 60, 500, 2
    hops = 1 if arch == "gcn_pre" else 2
    adj = _rand_csc(rng, V, 20000)
    g = DeviceGraph(adj)
    feats = rng.random((V, Fdim), dtype=np.float32)
    train = np.arange(0, V, 2, dtype=np.int64)
    smp = NeighborSampler(g, B, k, neighbor_type='in', shuffle=False, num_hops=hops, seed_nodes=train, seed=3)
    nf = next(iter(smp))
    csc = spsp.csc_matrix(adj); csc.sort_indices()
    ref = oracle.sample_nodeflow(csc.indptr, csc.indices, train[:B], k, hops, 3, 0, 0)
    nm = ref["node_mapping"]; o = ref["layer_offsets"]
    norm = (1.0 / np.maximum(1, np.diff(csc.indptr))).astype(np.float32).reshape(-1, 1)
    for i in range(nf.num_layers):
        ids = nm[o[i]:o[i + 1]]
        nf._node_frames[i] = {"features": torch.from_numpy(feats[ids]).to(dev), "norm": torch.from_numpy(norm[ids]).to(dev)}
    torch.manual_seed(1)
    if arch in ("gcn", "gcn_pre"):
        model = GCNSampling(Fdim, 32, C, 1, Fn.relu, 0.0, preprocess=(arch == "gcn_pre")).to(dev)
    elif arch == "infer":
        model = GCNInfer(Fdim, 32, C, 1, Fn.relu).to(dev)
    else:
        model = GraphSageSampling(Fdim, 16, C, 1, Fn.relu, 0.0, 'mean').to(dev)
    got = model(nf).detach().cpu().numpy()
    P = lambda lin: (lin.weight.detach().cpu().numpy(), lin.bias.detach().cpu().numpy())
    if arch == "gcn":
        want, _ = oracle.gcn_forward(ref, feats[nm[o[0]:o[1]]], [P(l.linear) for l in model.layers])
    elif arch == "gcn_pre":
        W, b = P(model.linear)
        z = feats[nm[o[0]:o[1]]] @ W.T + b
        h0 = np.concatenate([z, np.maximum(z, 0)], 1)
        W1, b1 = P(model.layers[0].linear)
        want = oracle.spmm_fwd(*ref["blocks"][0], h0, o[2] - o[1], "mean") @ W1.T + b1
    elif arch == "infer":
        h = feats[nm[o[0]:o[1]]]
        for i, l in enumerate(model.layers):
            W, b = P(l.linear)
            agg = oracle.spmm_fwd(*ref["blocks"][i], h, o[i + 2] - o[i + 1], "sum") * norm[nm[o[i + 1]:o[i + 2]]]
            z = agg @ W.T + b
            h = np.concatenate([z, np.maximum(z, 0)], 1) if i == 0 else z
        want = h
    else:
        params = [P(l.fc_self) + P(l.fc_neigh) for l in model.layers]
        want = oracle.sage_forward(ref, [feats[nm[o[i]:o[i + 1]]] for i in range(3)], params)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "g6_*.npz"))), ids=os.path.basename)
def test_closure_vs_golden(dev, hiplib, path):
    """get_sub_graph on the GPU == the reference's numpy tail on the stand-in sampler (utils.py:25-52)"""
    from pagraph_amd.partition.utils import get_sub_graph
    from pagraph_amd.sampling import DeviceGraph
    z = np.load(path)
    V = int(z["V"])
    csc = spsp.csc_matrix((np.ones(len(z["csc_indices"]), np.int8), z["csc_indices"], z["csc_indptr"]), shape=(V, V))
    g = DeviceGraph(csc)
    csr, sub2full, subtrain = get_sub_graph(g, z["train_nids"], int(z["hops"]))
    csr.sort_indices()
    assert np.array_equal(sub2full, z["sub2full"]) and np.array_equal(subtrain, z["subtrainid"])
    assert np.array_equal(csr.indptr, z["sub_indptr"]) and np.array_equal(csr.indices, z["sub_indices"])
    assert csr.data.dtype == np.uint8 and np.all(csr.data == 1)


def test_closure_vs_oracle_medium(dev, hiplib, oracle):
    from pagraph_amd.partition.utils import closure_device
    from pagraph_amd.sampling import DeviceGraph
    rng = np.random.default_rng(77)
    V = 20000
    adj = _rand_csc(rng, V, 60000)
    g = DeviceGraph(adj)
    csc = spsp.csc_matrix(adj); csc.sort_indices()
    train = np.sort(rng.choice(V, 2000, replace=False)).astype(np.int64)
    for hops in (1, 2, 3):
        ip, ix, s2f, st = closure_device(g, train, hops)
        o_ip, o_ix, o_s2f, o_st = oracle.closure_subgraph(csc.indptr, csc.indices, V, train, hops)
        assert np.array_equal(s2f.cpu().numpy(), o_s2f) and np.array_equal(st.cpu().numpy(), o_st)
        Vs = len(o_s2f)
        mine = spsp.csc_matrix((np.ones(ix.numel(), np.int8), ix.cpu().numpy(), ip.cpu().numpy()), shape=(Vs, Vs)).tocsr()
        mine.sort_indices()
        assert np.array_equal(mine.indptr, o_ip) and np.array_equal(mine.indices, o_ix)


def test_synthetic_generators_vs_oracle(dev, hiplib, oracle):
    from pagraph_amd.data import synthetic as syn
    s, d = syn.rmat_candidates(99, 17, 1000, 200000, dev)
    os_, od = oracle.rmat_edges(99, 17, 1000, 200000)
    assert np.array_equal(s.cpu().numpy(), os_) and np.array_equal(d.cpu().numpy(), od)
    f = syn.random_features_device(5000, 600, seed=5, device=dev, row0=123)
    assert np.array_equal(f.cpu().numpy(), oracle.random_features(5, 123, 5000, 600))
    f2 = syn.random_features_device(100, 602, seed=5, device=dev)
    assert np.array_equal(f2.cpu().numpy(), oracle.random_features(5, 0, 100, 602))
    assert float(f.min()) >= 0.0 and float(f.max()) < 1.0
    # graph builder: exactly E distinct undirected edges, symmetric, sorted columns, no loops
    V, E = 20000, 100000
    ip, ix = syn.rmat_graph(V, E, seed=3, device=dev)
    assert ip[-1].item() == 2 * E and ix.numel() == 2 * E
    a = spsp.csc_matrix((np.ones(2 * E, np.int8), ix.cpu().numpy(), ip.cpu().numpy()), shape=(V, V))
    assert a.has_sorted_indices or (a.sort_indices() is None)
    assert (a != a.T).nnz == 0 and a.diagonal().sum() == 0
    a.sum_duplicates(); assert a.nnz == 2 * E
    # sequential "first E unique candidates" semantics, checked with numpy on the oracle's candidates
    factor = 1.35
    while True:                                  # same candidate-count schedule as rmat_graph
        n = int(E * factor) + 1024
        cs, cd = oracle.rmat_edges(3, 15, 0, n)
        u = np.minimum(cs, cd); v = np.maximum(cs, cd)
        seen, chosen = set(), []
        for a_, b_ in zip(u.tolist(), v.tolist()):
            if a_ != b_ and b_ < V and (a_, b_) not in seen:
                seen.add((a_, b_)); chosen.append((a_, b_))
                if len(chosen) == E:
                    break
        if len(chosen) == E:
            break
        factor *= 1.5
    ref = spsp.coo_matrix((np.ones(E, np.int8), ([c[0] for c in chosen], [c[1] for c in chosen])), shape=(V, V))
    ref = (ref + ref.T).tocsc(); ref.sort_indices()
    assert np.array_equal(ref.indptr, ip.cpu().numpy()) and np.array_equal(ref.indices, ix.cpu().numpy())


def test_full_size_gather_properties(dev, hiplib):
    """BASELINE-size gather (R = 1M rows, F = 600, 30% cache) checked through size-independent
    properties: every output row equals its source row (row checksums match a checksum of the
    table gathered by torch), hits + misses partition the rows, idempotence."""
    from pagraph_amd import _lib as L
    from pagraph_amd.data import synthetic as syn
    V, F, R = 400000, 600, 1 << 20
    table = syn.random_features_device(V, F, seed=1, device=dev)       # the "host" table, kept in HBM for the check
    deg_order = torch.randperm(V, device=dev)
    cached = deg_order[:int(V * 0.3)].contiguous()
    cache = table[cached].contiguous()
    slot = torch.empty(V, dtype=torch.int32, device=dev)
    sp = L.stream_ptr()
    L.check(hiplib.pg_slot_map_reset(L.ptr(slot), V, sp))
    L.check(hiplib.pg_slot_map_assign(L.ptr(slot), L.ptr(cached), cached.numel(), sp))
    ids = torch.randint(0, V, (R,), device=dev)
    nid_map = torch.arange(V, device=dev)
    out = torch.zeros((R, F), device=dev)
    mpos = torch.empty(R, dtype=torch.int32, device=dev); mfull = torch.empty(R, dtype=torch.int64, device=dev)
    mcnt = torch.zeros(1, dtype=torch.int32, device=dev)
    fields, nf = L.make_fields([(cache, out, F, F, F)])
    for _ in range(2):                                                   # idempotent
        ml = L.miss_list(mpos, mfull, mcnt)
        L.check(hiplib.pg_gather_rows(L.ptr(ids), R, L.ptr(slot), L.ptr(nid_map), fields, nf, ctypes.byref(ml), None, None, None,
                                      None, sp))
    m = int(mcnt.item())
    hit = slot[ids] >= 0
    assert m == int((~hit).sum())
    assert torch.equal(torch.sort(mpos[:m].long()).values, torch.nonzero(~hit).squeeze(1))
    assert torch.equal(mfull[:m], ids[mpos[:m].long()])
    L.check(hiplib.pg_scatter_rows(L.ptr(table[mfull[:m]].contiguous()), L.ptr(mpos), m, None, F, L.ptr(out), F, sp))
    assert torch.equal(out, table[ids])


def test_trainer_loop_runs_and_learns(dev, hiplib):
    """a few steps of the pa_gcn.py loop: loss decreases on a learnable synthetic task"""
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    rng = np.random.default_rng(0)
    V, Fdim, C = 6000, 64, 4
    labels = rng.integers(0, C, V)
    # homophilous graph: a GCN without self loops can only learn a vertex's label from its neighbours
    order = np.argsort(labels, kind="stable"); start = np.searchsorted(labels[order], np.arange(C + 1))
    s_ = rng.integers(0, V, 40000)
    d_ = order[start[labels[s_]] + rng.integers(0, 1 << 30, 40000) % (start[labels[s_] + 1] - start[labels[s_]])]
    adj = spsp.coo_matrix((np.ones(80000, np.int8), (np.concatenate([s_, d_]), np.concatenate([d_, s_]))), shape=(V, V)).tocsr()
    g = DeviceGraph(adj)
    feats = (np.eye(C, Fdim, dtype=np.float32)[labels] + 0.1 * rng.standard_normal((V, Fdim))).astype(np.float32)
    store = HostFeatureStore({"features": torch.from_numpy(feats)})
    c = GraphCacheServer(store, V, torch.arange(V), 0)
    c.init_field(["features"])
    c.log = True
    train = np.arange(V, dtype=np.int64)
    smp = NeighborSampler(g, 512, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True, seed=1)
    torch.manual_seed(0)
    model = GCNSampling(Fdim, 32, C, 1, Fn.relu, 0.2).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=3e-2)
    lab = torch.from_numpy(labels).to(dev)
    losses = []
    for epoch in range(3):
        for step, nf in enumerate(smp):
            c.fetch_data(nf)
            y = lab[nf.layer_parent_nid(-1)]
            loss = Fn.cross_entropy(model(nf), y)
            opt.zero_grad(); loss.backward(); opt.step()
            losses.append(loss.item())
            if epoch == 0 and step == 0:
                c.auto_cache(g, ["features"], cache_ratio=0.3)
    assert losses[-1] < 0.7 * losses[0], (losses[0], losses[-1])
    mr = c.get_miss_rate()
    assert 0.0 < mr < 1.0


@pytest.mark.parametrize("mode", ["staged", "zerocopy", "async"])
def test_fetch_only_what_the_model_reads(dev, hiplib, golden_dir, mode):
    """fetch_data(need=...) returns, for the requested layers/fields, exactly the rows
    the reference's full fetch_data returns (golden G1), and touches nothing else"""
    z = np.load(os.path.join(golden_dir, "g1_fetch_data_F600.npz"))
    c = _cacher(z, dev, mode)
    c.log = True
    nids = torch.from_numpy(z["cached_nids"]).to(dev)
    c.cache_fix_data(nids, c.get_feat_from_server(nids, ["features", "norm"], to_gpu=True), is_full=False)
    layers = [z[f"layer{i}_nids"] for i in range(int(z["num_layers"]))]
    for need in ({0: ["features"]}, {3: ["norm"], 4: ["features", "norm"]}, {1: ["features"], 2: [], 3: ["norm"]}):
        nf = FakeNF(layers, dev)
        c.fetch_data(nf, need=need, slot=1)
        c.wait_misses(1)
        torch.cuda.synchronize()
        for i in range(len(layers)):
            got = nf._node_frames[i]
            assert sorted(got) == sorted(need.get(i, []))
            for name in got:
                assert np.array_equal(got[name].cpu().numpy(), z[f"layer{i}_{name}"]), (need, i, name)
    # miss accounting covers only the rows that were looked up
    c.get_miss_rate()
    nf = FakeNF(layers, dev)
    c.fetch_data(nf, need={0: ["features"]}, slot=0)
    c.wait_misses(0)
    t, m = c._stats.tolist()
    assert t == len(layers[0]) and m == int((~z["state_gpu_flag"][layers[0]]).sum())


def test_graphed_trainer_matches_eager(dev, hiplib):
    """the hipGraph-replayed step (padded fixed-shape NodeFlows) follows the same loss trajectory as
    the eager reference-style loop on the same seeds (dropout off => deterministic up to fp32 atomics)"""
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, MinibatchTrainer, cycle_batches
    rng = np.random.default_rng(5)
    V, Fdim, C, B = 5000, 64, 5, 500
    adj = _rand_csc(rng, V, 30000)
    g = DeviceGraph(adj)
    feats = rng.standard_normal((V, Fdim)).astype(np.float32)
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    train = np.arange(0, V, 2, dtype=np.int64)          # 2500 seeds: 5 full batches, no short batch
    losses = {}
    for mode in ("eager", "graph", "graph-async", "graph-hipgraph"):
        # graph-hipgraph: the captured step replayed with hipGraphLaunch (PG_FLAT_REPLAY=0) instead of as plain launches of
        # its kernels (pg_tape_launch, the default on one GPU): same kernels, same arguments, same order
        os.environ.pop("PG_FLAT_REPLAY", None)
        if mode == "graph-hipgraph":
            os.environ["PG_FLAT_REPLAY"] = "0"
        store = HostFeatureStore({"features": torch.from_numpy(feats)})
        c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode="async" if mode == "graph-async" else "zerocopy")
        c.init_field(["features"])
        c.auto_cache(g, ["features"], cache_ratio=0.4)
        torch.manual_seed(0)
        model = GCNSampling(Fdim, 16, C, 1, Fn.relu, 0.0).to(dev)
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, capturable=(mode != "eager"))
        smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True,
                              seed=9, static=(mode != "eager"))
        cls = GraphedTrainer if mode != "eager" else MinibatchTrainer
        tr = cls(model, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=model.required_inputs(3))
        out = []
        tr.on_step = lambda step, loss: out.append(loss.detach())   # valid after a device sync
        tr.run_steps(cycle_batches(smp, 20), 20)
        torch.cuda.synchronize()
        losses[mode] = torch.stack(out).cpu().numpy()
        if mode != "eager":
            # (this test's torch.optim.Adam(capturable=True) puts nodes into the graph that a tape does not take — the trainer
            # then keeps the graph launch; the trainers of bench.py / pa_gcn.py use pagraph_amd.optim.Adam: see the test below)
            taped = [s_.tape is not None for s_ in tr.slots.values() if s_.graph is not None]
            assert taped and (mode != "graph-hipgraph" or not any(taped)), (mode, taped)
        os.environ.pop("PG_FLAT_REPLAY", None)
    assert np.allclose(losses["eager"], losses["graph"], rtol=2e-4, atol=2e-5), (losses["eager"], losses["graph"])
    assert np.allclose(losses["eager"], losses["graph-async"], rtol=2e-4, atol=2e-5)
    assert np.array_equal(losses["graph"], losses["graph-hipgraph"])       # bit for bit: the very same launches
    assert losses["graph"][-1] < losses["graph"][0]


def test_tape_refuses_a_capture_that_consumed_torch_rng(dev, hiplib):
    """torch.cuda.CUDAGraph.replay() refreshes the Philox seed / offset of every torch RNG kernel in
    the graph; the plain-launch tape (pg_tape_launch) cannot. A model that falls back to nn.Dropout (fuse_dropout=False)
    must therefore keep hipGraphLaunch — same loss trajectory with PG_FLAT_REPLAY on and off, no tape built — while the
    same model with the fused mask (no torch RNG in the capture) is taped."""
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, cycle_batches
    rng = np.random.default_rng(11)
    V, Fdim, C, B = 4000, 64, 5, 400
    g = DeviceGraph(_rand_csc(rng, V, 24000))
    feats = rng.standard_normal((V, Fdim)).astype(np.float32)
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    train = np.arange(0, V, 2, dtype=np.int64)
    res = {}
    for mode in ("torch-rng-flat", "torch-rng-graph", "fused-flat"):
        os.environ.pop("PG_FLAT_REPLAY", None)
        if mode == "torch-rng-graph":
            os.environ["PG_FLAT_REPLAY"] = "0"
        try:
            store = HostFeatureStore({"features": torch.from_numpy(feats)})
            c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode="zerocopy")
            c.init_field(["features"])
            c.auto_cache(g, ["features"], cache_ratio=1.0)
            torch.manual_seed(0)
            torch.cuda.manual_seed(1234)
            model = GCNSampling(Fdim, 16, C, 1, Fn.relu, 0.3).to(dev)
            model.fuse_dropout = not mode.startswith("torch-rng")
            opt = Adam(model.parameters(), lr=1e-2)
            smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True, seed=9,
                                  static=True)
            tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=model.required_inputs(3))
            tr.fuse_gather = model.fuse_dropout       # nn.Dropout needs materialised rows
            out = []
            tr.on_step = lambda step, loss: out.append(loss.detach().clone())
            tr.run_steps(cycle_batches(smp, 40), 40)
            tr.synchronize()
            torch.cuda.synchronize()
            res[mode] = (torch.stack(out).cpu().numpy(), [s_.tape is not None for s_ in tr.slots.values() if s_.graph is not None])
            tr.close()
        finally:
            os.environ.pop("PG_FLAT_REPLAY", None)
    assert res["torch-rng-flat"][1] and not any(res["torch-rng-flat"][1])          # RNG in the capture: no tape
    assert res["fused-flat"][1] and all(res["fused-flat"][1])                      # none: taped
    assert np.isfinite(res["torch-rng-flat"][0]).all()
    # the same masks, replay after replay, as hipGraphLaunch draws them (a tape would repeat one mask per ring slot)
    assert np.array_equal(res["torch-rng-flat"][0], res["torch-rng-graph"][0])


@pytest.mark.parametrize("arch,hidden,C,p_drop", [("sage", 32, 5, 0.0), ("sage", 16, 70, 0.0), ("sage", 32, 70, 0.25),
                                                  ("gcn", 64, 5, 0.0), ("gcn", 16, 70, 0.25), ("sage", 8, 5, 0.0)])
def test_graphed_trainer_outside_the_fused_head_envelope(dev, hiplib, arch, hidden, C, p_drop):
    """`pa_gs.py --n-hidden 32`, a dataset with more than 64 classes: the fused output head declines, and it
    must decline BEFORE the layers below it consume the NodeFlow's frames — GraphedTrainer's fall-back `model(nf)` then
    trains as the eager reference-style loop does (same losses, dropout off; finite and falling with dropout on).
    The last case is inside the envelope: the control."""
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling, GraphSageSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, MinibatchTrainer, cycle_batches
    rng = np.random.default_rng(11)
    V, Fdim, B = 5000, 64, 500
    g = DeviceGraph(_rand_csc(rng, V, 30000))
    feats = rng.standard_normal((V, Fdim)).astype(np.float32)
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    train = np.arange(0, V, 2, dtype=np.int64)
    losses = {}
    for mode in ("eager", "graph"):
        c = GraphCacheServer(HostFeatureStore({"features": torch.from_numpy(feats)}), V, torch.arange(V), 0, miss_mode="async")
        c.init_field(["features"])
        c.auto_cache(g, ["features"], cache_ratio=0.4)
        torch.manual_seed(0)
        model = (GraphSageSampling(Fdim, hidden, C, 1, Fn.relu, p_drop, 'mean') if arch == "sage"
                 else GCNSampling(Fdim, hidden, C, 1, Fn.relu, p_drop)).to(dev)
        opt = Adam(model.parameters(), lr=1e-2)
        smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True,
                              seed=9, static=(mode != "eager"))
        cls = GraphedTrainer if mode != "eager" else MinibatchTrainer
        tr = cls(model, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=model.required_inputs(3))
        out = []
        tr.on_step = lambda step, loss: out.append(loss.detach().clone())
        tr.run_steps(cycle_batches(smp, 20), 20)
        torch.cuda.synchronize()
        losses[mode] = torch.stack(out).cpu().numpy()
        c.shutdown_miss_queue()
    for v in losses.values():
        assert np.isfinite(v).all() and v[-5:].mean() < v[:5].mean(), v
    if p_drop == 0.0:
        assert np.allclose(losses["eager"], losses["graph"], rtol=3e-4, atol=3e-5), (losses["eager"], losses["graph"])


@pytest.mark.parametrize("arch", ["gcn", "sage"])
@pytest.mark.parametrize("ratio,miss_mode", [(1.0, "async"), (0.4, "async")])
def test_early_layer0_aggregation_matches_the_in_step_one(dev, hiplib, ratio, miss_mode, arch, monkeypatch):
    """GraphedTrainer.early_aggregate: block 0's aggregation launched from prepare() on the load stream (ahead of its step)
    gives the same loss trajectory, bit for bit, as the aggregation inside the replayed step (dropout off: the same kernel on
    the same rows) with the table cached; a partial cache keeps the aggregation in the step whatever the mode; with dropout on
    the early launch draws the very masks the in-step path draws."""
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling, GraphSageSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, cycle_batches
    rng = np.random.default_rng(77)
    V, Fdim, C, B = 6000, 600, 7, 600
    adj = _rand_csc(rng, V, 40000)
    g = DeviceGraph(adj)
    feats = rng.standard_normal((V, Fdim)).astype(np.float32)
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    train = np.arange(0, V, 2, dtype=np.int64)

    def run(early, p_drop, steps=24):
        store = HostFeatureStore({"features": torch.from_numpy(feats)})
        c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode=miss_mode)
        c.init_field(["features"])
        c.auto_cache(g, ["features"], cache_ratio=ratio)
        torch.manual_seed(0)
        if arch == "gcn":
            model = GCNSampling(Fdim, 32, C, 1, Fn.relu, p_drop).to(dev)
        else:                     # GraphSAGE: BOTH blocks' aggregations of raw rows run ahead (graphsage_nssc.py:92-111)
            model = GraphSageSampling(Fdim, 16, C, 1, Fn.relu, p_drop, 'mean').to(dev)
        opt = Adam(model.parameters(), lr=1e-2)
        smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True,
                              seed=9, static=True, defer_transpose=True)
        tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=model.required_inputs(3))
        tr.early_aggregate = early
        out = []
        tr.on_step = lambda step, loss: out.append(loss.detach().clone())
        tr.run_steps(cycle_batches(smp, steps), steps)
        tr.synchronize()
        used = sum(1 for s_ in tr.slots.values() if s_.early is not None)
        # the captured step (library kernels only with pagraph_amd.optim.Adam) is replayed as plain launches of its kernels
        # (pg_tape_launch) unless PG_FLAT_REPLAY=0
        taped = [s_.tape is not None for s_ in tr.slots.values() if s_.graph is not None]
        assert taped and all(taped) == (os.environ.get("PG_FLAT_REPLAY", "1") != "0"), taped
        # ... and prepare() is ONE C call (pg_batch_prepare) whenever the table is resident, unless PG_NATIVE_PREPARE=0
        native = sum(1 for s_ in tr.slots.values() if s_.batch_plan)
        assert (native == len(tr.slots)) == (ratio == 1.0 and os.environ.get("PG_NATIVE_PREPARE", "1") != "0"), native
        return torch.stack(out).cpu().numpy(), tr.early_ordinal, used

    monkeypatch.setenv("PG_FLAT_REPLAY", "0")                      # the reference run: hipGraphLaunch, call-by-call prepare,
    monkeypatch.setenv("PG_NATIVE_PREPARE", "0")                   # aggregation in the step
    base, n0, used0 = run("0", 0.0)
    monkeypatch.delenv("PG_FLAT_REPLAY")
    monkeypatch.delenv("PG_NATIVE_PREPARE")
    assert n0 == 0 and used0 == 0
    assert np.array_equal(run("0", 0.0)[0], base)                  # the same step as plain launches: bit for bit
    if ratio < 1.0:
        for mode in ("auto", "1"):                                 # a partial cache keeps the aggregation in the step
            got, n1, used1 = run(mode, 0.0)
            assert n1 == 0 and used1 == 0 and np.array_equal(got, base)
        return
    for mode in ("auto", "1"):
        got, n1, used1 = run(mode, 0.0)
        assert n1 >= 24 and used1 > 0, (mode, n1, used1)          # every batch of the run (+ the look-ahead) went early
        assert np.array_equal(got, base), (mode, got, base)
    # dropout on: the early launch is keyed by the value the model's step counter will hold when the batch is computed —
    # the masks, and with them every loss, are the in-step path's
    drop, n2, _ = run("1", 0.3, steps=40)
    ref, _, _ = run("0", 0.3, steps=40)
    assert np.isfinite(drop).all() and np.array_equal(drop, ref), (drop, ref)
    assert not np.array_equal(drop[:24], base)                     # ... and dropout was really on


@pytest.mark.parametrize("on_current_stream", [False, True])
def test_fetch_plan_buffers_come_from_the_stream_that_fills_them(dev, hiplib, monkeypatch, on_current_stream):
    """The rare hipErrorIllegalAddress of rounds 4-5, made deterministic. A fetch plan's slot array is first written by k_split
    on the LOAD stream. Allocated while the COMPUTE stream is current (as run_steps has it), torch's allocator may hand out a
    block that an eager step has just freed while its kernels are still queued — legal for a tensor next used on that stream —
    and those kernels then write their floats over the slots the load stream has filled meanwhile; pg_spmm_fwd_rows later
    follows float bit patterns as cache slots (named by the PG_BOUNDS build under host load). GraphedTrainer.prepare
    therefore builds the plan with the load stream current. PG_PLAN_ON_CURRENT_STREAM=1 (second case) restores the old
    allocation and the very same sequence corrupts the slot array."""
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer
    if on_current_stream:
        monkeypatch.setenv("PG_PLAN_ON_CURRENT_STREAM", "1")
    rng = np.random.default_rng(3)
    V, Fd, C, B = 6000, 600, 7, 600
    g = DeviceGraph(_rand_csc(rng, V, 40000))
    feats = torch.from_numpy(rng.standard_normal((V, Fd)).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    c = GraphCacheServer(HostFeatureStore({"features": feats}), V, torch.arange(V), 0, miss_mode="async")
    c.init_field(["features"])
    c.auto_cache(g, ["features"], cache_ratio=0.4)
    model = GCNSampling(Fd, 32, C, 1, Fn.relu, 0.0).to(dev)
    smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=np.arange(0, V, 2), prefetch=True,
                          seed=1, static=True, defer_transpose=True)
    tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), Adam(model.parameters(), lr=1e-2), c, smp, labels, dev,
                        need=model.required_inputs(3))
    it = iter(smp)
    nf = next(it)
    rows = nf.layer_size(0)                       # the plan covers layer 0 (the only layer a sampled GCN fetches)
    cs = tr.compute_stream
    prev = torch.cuda.current_stream(dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(cs)                     # what run_steps does for the whole loop
    try:
        # an "eager step": float outputs whose kernels are still queued (behind a spin) when Python has already freed them
        # (whatever this stream's pool already holds in that size class is taken out of the way first: the slot array must
        # come out of the blocks that are about to be freed)
        hold, r0 = [], torch.cuda.memory_reserved(dev)
        while torch.cuda.memory_reserved(dev) == r0 and len(hold) < 4096:
            hold.append(torch.empty(rows, dtype=torch.float32, device=dev))
        torch.cuda._sleep(300_000_000)
        junk = [torch.empty(rows, dtype=torch.float32, device=dev) for _ in range(48)]
        for t in junk:
            t.fill_(3.25)
        del junk, t
        s = tr.prepare(nf)                        # builds the slot's fetch plan and runs k_split on the load stream
        tr.load_stream.synchronize()              # the split is done; the compute stream is still spinning
        assert cs.query() is False
        with torch.cuda.stream(tr.load_stream):
            filled = s.plan.slots.clone()
        tr.load_stream.synchronize()
        cs.synchronize()                          # the stale fills run now
        after = s.plan.slots.clone()
        cs.synchronize()
    finally:
        torch.cuda.set_stream(prev)
    sl = filled.cpu().numpy()
    assert ((sl >= -2 - rows) & (sl < c.cached_num)).all()            # cache slots, staged-row numbers, padding
    poisoned = int((after.cpu().numpy() == np.float32(3.25).view(np.int32)).sum())
    if on_current_stream:
        if poisoned == 0:
            pytest.skip("the allocator handed out other blocks this time: the hazard did not show (it does when run alone)")
    else:
        assert poisoned == 0 and torch.equal(after, filled)
    smp.release(nf)
    tr.close(); smp.close(); c.close()


def _reachable_cuda_tensors(roots):
    """(path, tensor) for every CUDA tensor reachable from `roots` through attributes, slots and containers: any class, any
    depth — deliberately NOT _lib.record_streams' walk (pagraph_amd classes only, depth 6, a skip list of attribute names)"""
    import types
    seen, out, stack = set(), [], [(r, n) for n, r in roots.items()]
    skip = (str, bytes, int, float, bool, type(None), type, types.ModuleType, types.FunctionType, types.BuiltinFunctionType,
            types.CodeType, torch.cuda.Stream, torch.cuda.Event, torch.dtype, torch.device, np.ndarray)
    while stack:
        o, path = stack.pop()
        if isinstance(o, skip) or id(o) in seen:
            continue
        seen.add(id(o))
        if torch.is_tensor(o):
            if o.is_cuda:
                out.append((path, o))
            if o.grad is not None:
                stack.append((o.grad, path + ".grad"))
        elif isinstance(o, dict):
            stack.extend((v, path + "[%s]" % (k if isinstance(k, (str, int, tuple)) else type(k).__name__,))
                         for k, v in list(o.items()))
        elif isinstance(o, (list, tuple, set, frozenset)):
            stack.extend((v, "%s[%d]" % (path, i)) for i, v in enumerate(list(o)))
        elif isinstance(o, types.MethodType):
            stack.append((o.__self__, path + ".__self__"))
        else:
            d = getattr(o, "__dict__", None)
            if isinstance(d, dict):
                stack.extend((v, path + "." + k) for k, v in list(d.items()))
            for klass in type(o).__mro__:
                for k in getattr(klass, "__slots__", ()) or ():
                    if hasattr(o, k):
                        stack.append((getattr(o, k), path + "." + k))
    return out


@pytest.mark.parametrize("kind", ["gcn", "sage"])
def test_every_buffer_a_pipeline_reaches_is_recorded_or_known_single_stream(dev, hiplib, monkeypatch, kind):
    """The lifetime rule of the pipeline (_lib.record_streams: a buffer used on a stream other than the one it was allocated on is
    recorded there, so the allocator does not recycle it under kernels in flight) is enforced by a walk over the owners'
    attributes that a new attribute on a new class could escape silently. Here Tensor.record_stream is spied on while a
    pipeline is built and run, the owners are walked WITHOUT the walk's limits, and every device buffer found must either have
    been recorded on a foreign stream or be one of the few kinds that live on a single stream by construction (listed below
    with the reason). Round 6 found the fetch plan's slot array this way: allocated on the load stream, read by the compute
    stream's fused aggregation, recorded nowhere."""
    import re
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling, GraphSageSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, cycle_batches
    recorded = {}
    orig = torch.Tensor.record_stream

    def spy(t, s):
        recorded.setdefault(t.untyped_storage().data_ptr(), set()).add(int(s.cuda_stream))
        return orig(t, s)
    monkeypatch.setattr(torch.Tensor, "record_stream", spy)
    rng = np.random.default_rng(7)
    V, Fdim, C, B = 6000, 256, 5, 400
    g = DeviceGraph(_rand_csc(rng, V, 40000))
    feats = torch.from_numpy(rng.standard_normal((V, Fdim)).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    c = GraphCacheServer(HostFeatureStore({"features": feats}), V, torch.arange(V), 0, miss_mode="async", host_threads=3)
    c.init_field(["features"])
    c.auto_cache(g, ["features"], cache_ratio=0.3)
    torch.manual_seed(0)
    model = (GCNSampling(Fdim, 16, C, 1, Fn.relu, 0.2) if kind == "gcn"
             else GraphSageSampling(Fdim, 16, C, 1, Fn.relu, 0.2, 'mean')).to(dev)
    opt = Adam(model.parameters(), lr=1e-2)
    smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=np.arange(0, V, 2, dtype=np.int64),
                          prefetch=True, seed=1, static=True, defer_transpose=True)
    tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=model.required_inputs(3),
                        keep_losses=False)
    tr.run_steps(cycle_batches(smp, 48), 40)           # eager warm-up, both captures, replays: every lazy buffer exists
    tr.synchronize()
    single_stream = [
        (r"\.grad$", "parameter gradients: created and consumed by the step, on the compute stream"),
        (r"^opt\.state\[", "Adam moments: created by the first step on the compute stream, touched by nothing else"),
        (r"^tr\._gseed$|\.loss$|^tr\.last_loss$", "the step's own scalars (compute stream; the host reads them after a synchronise)"),
        (r"_out_deg$", "auto_cache's degree vector: default stream only, before training"),
    ]
    by_storage = {}
    for path, t in _reachable_cuda_tensors({"tr": tr, "smp": smp, "c": c, "opt": opt, "model": model}):
        by_storage.setdefault(t.untyped_storage().data_ptr(), []).append(path)
    assert len(by_storage) > 60                        # the walk did reach the pipeline's buffers
    loose = []
    for ptr_, paths in by_storage.items():
        if recorded.get(ptr_):
            continue
        if not any(re.search(rx, p) for p in paths for rx, _why in single_stream):
            loose.append(min(paths, key=len))
    tr.close(); smp.close(); c.close()
    assert not loose, "device buffers no stream was recorded for: %s" % sorted(loose)


def test_stress_objects_dropped_with_work_in_flight(dev, hiplib):
    """Samplers, cachers (async miss queue: worker thread, gather pool, SDMA copies), trainers with captured step graphs and
    optimisers with a mirrored step counter are created, driven WITHOUT a final synchronise, and dropped in every order
    while replays, sampling chains and miss jobs are still in flight — every *_destroy / __del__ path of _lib.py,
    storage.py, sampler.py, optim.py (one unexplained core dump of the test process in round 3). The process
    must survive, later pipelines must still train, and nothing may be reported lost."""
    import gc
    import itertools
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling, GraphSageSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, cycle_batches
    rng = np.random.default_rng(99)
    V, Fdim, C, B = 6000, 256, 5, 400
    adj = _rand_csc(rng, V, 40000)
    g = DeviceGraph(adj)
    feats = torch.from_numpy(rng.standard_normal((V, Fdim)).astype(np.float32))
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    train = np.arange(0, V, 2, dtype=np.int64)
    orders = list(itertools.permutations(range(5)))
    rng.shuffle(orders)
    last = None
    for it, order in enumerate(orders[:14]):
        store = HostFeatureStore({"features": feats})
        c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode="async", host_threads=3)
        c.init_field(["features"])
        c.auto_cache(g, ["features"], cache_ratio=0.3)
        torch.manual_seed(it)
        model = (GCNSampling(Fdim, 16, C, 1, Fn.relu, 0.2) if it % 2 == 0 else GraphSageSampling(Fdim, 16, C, 1, Fn.relu, 0.2, 'mean')).to(dev)
        opt = Adam(model.parameters(), lr=1e-2)
        smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True, seed=it,
                              static=True, defer_transpose=True)
        tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=model.required_inputs(3),
                            keep_losses=False)
        tr.keep_primed = (it % 3 == 0)            # leaves prepared batches (held ring slots, submitted miss jobs) behind
        n = 19 + 3 * (it % 4)                     # eager warm-up, captures, a few replays — the tail is still running below
        tr.run_steps(cycle_batches(smp, n + 8), n)
        loss = tr.last_loss
        if it % 5 == 4:
            tr.synchronize()                      # some iterations do end cleanly
            assert torch.isfinite(loss).all()
        # drop the five owners in this iteration's order, collecting after each, work still in flight
        owners = {0: "tr", 1: "smp", 2: "c", 3: "opt", 4: "model"}
        scope = {"tr": tr, "smp": smp, "c": c, "opt": opt, "model": model}
        del tr, smp, c, opt, model, store
        for k in order:
            scope.pop(owners[k])
            gc.collect()
        last = loss
        del scope, loss
        gc.collect()
    torch.cuda.synchronize()
    assert last is None or True
    # the device and the library are still healthy: one more pipeline trains, eagerly checked
    store = HostFeatureStore({"features": feats})
    c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode="async", host_threads=3)
    c.init_field(["features"])
    c.auto_cache(g, ["features"], cache_ratio=0.3)
    torch.manual_seed(0)
    model = GCNSampling(Fdim, 16, C, 1, Fn.relu, 0.0).to(dev)
    opt = Adam(model.parameters(), lr=1e-2)
    smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True, seed=1,
                          static=True, defer_transpose=True)
    tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=model.required_inputs(3))
    out = []
    tr.on_step = lambda step, loss: out.append(loss.detach().clone())
    tr.run_steps(cycle_batches(smp, 60), 60)
    tr.synchronize()
    lv = torch.stack(out).cpu()
    assert torch.isfinite(lv).all() and float(lv[-10:].mean()) < float(lv[:10].mean())
    c.shutdown_miss_queue()


@pytest.mark.parametrize("n,K,N,bias", [(12000, 600, 32, True), (11999, 600, 16, True), (1025, 600, 32, True), (1033, 128, 7, False),
                                        (6000, 1200, 32, True), (1257, 608, 1, True), (5000, 600, 32, False),
                                        (6000, 64, 60, True), (5999, 602, 32, True), (3000, 70, 33, True)])
def test_skinny_linear_vs_torch(dev, hiplib, n, K, N, bias):
    """pg_linear_fwd / pg_linear_bwd_w (fp32 MFMA) vs torch's nn.Linear: outputs and weight/bias gradients
    within 1e-4 relative to the output scale (same exact-fp32 arithmetic, different summation order)"""
    from pagraph_amd import ops
    torch.manual_seed(n + K + N)
    lin = torch.nn.Linear(K, N, bias=bias).to(dev)
    x = torch.rand((n, K), device=dev) - 0.3
    y = ops.linear(x, lin)
    assert y.grad_fn is not None and "SkinnyLinear" in type(y.grad_fn).__name__      # the HIP path ran
    ref = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double() if bias else None)
    scale = max(1.0, float(ref.abs().max()))
    assert float((y.double() - ref).abs().max()) < TOL * scale
    gy = torch.rand_like(y) - 0.5
    y.backward(gy)
    gw_ref = gy.double().t() @ x.double()
    assert float((lin.weight.grad.double() - gw_ref).abs().max()) < TOL * max(1.0, float(gw_ref.abs().max()))
    if bias:
        gb_ref = gy.double().sum(0)
        assert float((lin.bias.grad.double() - gb_ref).abs().max()) < TOL * max(1.0, float(gb_ref.abs().max()))
    # fused activation epilogues: relu and NodeUpdate's skip-concat (gcn_nssc.py:20-23), forward + all gradients
    for act in (ops.ACT_RELU, ops.ACT_CONCAT):
        lin.zero_grad()
        xa = x[:2048].clone().requires_grad_(True)
        ya = ops.linear(xa, lin, act)
        za = torch.nn.functional.linear(xa.detach().double().requires_grad_(True), lin.weight.double(),
                                        lin.bias.double() if bias else None)
        zin = za
        ra = torch.relu(za) if act == ops.ACT_RELU else torch.cat((za, torch.relu(za)), 1)
        assert ya.shape == ra.shape and float((ya.double() - ra).abs().max()) < TOL * scale
        ga = torch.rand_like(ya) - 0.5
        ya.backward(ga)
        gz = torch.autograd.grad(ra, zin, ga.double())[0]
        gw_a = gz.t() @ xa.detach().double()
        assert float((lin.weight.grad.double() - gw_a).abs().max()) < TOL * max(1.0, float(gw_a.abs().max()))
        assert float((xa.grad.double() - gz @ lin.weight.double()).abs().max()) < TOL * max(1.0, float(gz.abs().max()) * K ** 0.5)
        if bias:
            assert float((lin.bias.grad.double() - gz.sum(0)).abs().max()) < TOL * max(1.0, float(gz.sum(0).abs().max()))
    # an input that needs its own gradient gets one too (deeper layers)
    x2 = x[:1024].clone().requires_grad_(True)
    lin.zero_grad()
    ops.linear(x2, lin).sum().backward()
    assert torch.allclose(x2.grad, lin.weight.sum(0).expand_as(x2), rtol=1e-5, atol=1e-6)


def test_skinny_linear_falls_back_outside_envelope(dev, hiplib):
    from pagraph_amd import ops
    lin = torch.nn.Linear(600, 32).to(dev)                  # too few rows to matter
    y = ops.linear(torch.rand((100, 600), device=dev), lin)
    assert "SkinnyLinear" not in type(y.grad_fn).__name__
    lin = torch.nn.Linear(64, 128).to(dev)                  # wide output
    y = ops.linear(torch.rand((5000, 64), device=dev), lin)
    assert "SkinnyLinear" not in type(y.grad_fn).__name__


def test_config2_reddit_shape_full_cache(dev, hiplib, oracle):
    """BASELINE.json configs[1]: Reddit-shaped graph (V = 232 965, F = 602, 41 classes, 153 431 train
    vertices), whole feature table resident in HBM (full_cached path, storage.py:90-95,207-216), GCN.
    Degree scaled down (mean ~40 instead of 492) to keep the CPU oracle in seconds. Sampled ids and the
    gathered F=602 rows (dwordx2 path) are bit exact, logits within 1e-4."""
    import torch.nn.functional as Fn
    from pagraph_amd.data import synthetic as syn
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    V, E, Fdim, C, B = 232965, 4_600_000, 602, 41, 6000
    ip, ix = syn.rmat_graph(V, E, seed=77, device=dev)
    g = DeviceGraph.from_csc(ip, ix, V)
    feats = syn.random_features_device(V, Fdim, seed=3, device=dev).cpu()
    norm = (1.0 / (ip[1:] - ip[:-1]).float()).unsqueeze(1).cpu()
    store = HostFeatureStore({"features": feats, "norm": norm})
    c = GraphCacheServer(store, V, torch.arange(V), 0)
    c.init_field(["features", "norm"])
    c.log = True
    c.auto_cache(g, ["features", "norm"])                    # 288 GB HBM: the reference rule caches everything
    assert c.full_cached and c.cached_num == V
    train = torch.randperm(V, generator=torch.Generator().manual_seed(1))[:153431].sort().values
    smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True, seed=11)
    assert len(smp) == 26                                    # 26 steps / epoch
    iph, ixh = ip.cpu().numpy(), ix.cpu().numpy()
    seeds = smp.seeds.cpu().numpy()
    torch.manual_seed(0)
    model = GCNSampling(Fdim, 32, C, 1, Fn.relu, 0.0).to(dev)
    params = [(l.linear.weight.detach().cpu().numpy(), l.linear.bias.detach().cpu().numpy()) for l in model.layers]
    fnp, nnp = feats.numpy(), norm.numpy()
    for b, nf in enumerate(smp):
        if b not in (0, 25):
            continue                                         # a full batch and the short last one (3431 seeds)
        ref = oracle.sample_nodeflow(iph, ixh, seeds[b * B:(b + 1) * B], 2, 2, 11, 0, b)
        nm = nf._node_mapping.tousertensor().cpu().numpy()
        assert np.array_equal(nm, ref["node_mapping"])
        c.fetch_data(nf)
        o = ref["layer_offsets"]
        for i in range(3):
            assert np.array_equal(nf._node_frames[i]["features"].cpu().numpy(), fnp[nm[o[i]:o[i + 1]]])
            assert np.array_equal(nf._node_frames[i]["norm"].cpu().numpy(), nnp[nm[o[i]:o[i + 1]]], equal_nan=True)
        want, _ = oracle.gcn_forward(ref, fnp[nm[o[0]:o[1]]], params)
        got = model(nf).detach().cpu().numpy()
        assert got.shape == (len(seeds[b * B:(b + 1) * B]), C)
        assert np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max())
    assert c.get_miss_rate() == 0.0


def test_sage_preprocess_forward_vs_oracle(dev, hiplib, oracle):
    """GraphSageSampling(preprocess=True): every layer carries 'features' and a 'neigh' field
    (graphsage_nssc.py:75-87, pa_gs.py:46-49), one hop fewer"""
    import torch.nn.functional as Fn
    from pagraph_amd.model import GraphSageSampling
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    rng = np.random.default_rng(4)
    V, Fdim, C, B, k, hops = 3000, 64, 7, 400, 2, 1
    adj = _rand_csc(rng, V, 20000)
    g = DeviceGraph(adj)
    feats = rng.random((V, Fdim), dtype=np.float32); neigh = rng.random((V, Fdim), dtype=np.float32)
    train = np.arange(0, V, 2, dtype=np.int64)
    nf = next(iter(NeighborSampler(g, B, k, neighbor_type='in', shuffle=False, num_hops=hops, seed_nodes=train, seed=3)))
    csc = spsp.csc_matrix(adj); csc.sort_indices()
    ref = oracle.sample_nodeflow(csc.indptr, csc.indices, train[:B], k, hops, 3, 0, 0)
    nm, o = ref["node_mapping"], ref["layer_offsets"]
    for i in range(2):
        ids = nm[o[i]:o[i + 1]]
        nf._node_frames[i] = {"features": torch.from_numpy(feats[ids]).to(dev), "neigh": torch.from_numpy(neigh[ids]).to(dev)}
    torch.manual_seed(2)
    model = GraphSageSampling(Fdim, 16, C, 1, Fn.relu, 0.0, 'mean', preprocess=True).to(dev)
    got = model(nf).detach().cpu().numpy()
    P = lambda lin: (lin.weight.detach().cpu().numpy(), lin.bias.detach().cpu().numpy())
    (Ws, bs), (Wn, bn) = P(model.fc_self), P(model.fc_neigh)
    h = []
    for i in range(2):
        ids = nm[o[i]:o[i + 1]]
        z = feats[ids] @ Ws.T + bs + neigh[ids] @ Wn.T + bn
        h.append(np.concatenate([z, np.maximum(z, 0)], 1))
    (W1s, b1s), (W1n, b1n) = P(model.layers[0].fc_self), P(model.layers[0].fc_neigh)
    agg = oracle.spmm_fwd(*ref["blocks"][0], h[0], o[2] - o[1], "mean")
    want = h[1] @ W1s.T + b1s + agg @ W1n.T + b1n
    assert np.abs(got - want).max() < TOL * max(1.0, np.abs(want).max())


def test_server_preprocess_vs_scipy(dev, hiplib):
    """f-3: server-side feature preprocessing X' = norm * (A^T X) (pa_server.py:43-52) on the GPU"""
    from pagraph_amd.server import preprocess_features
    rng = np.random.default_rng(8)
    V, Fdim = 5000, 96
    adj = _rand_csc(rng, V, 30000)
    csc = spsp.csc_matrix(adj); csc.sum_duplicates(); csc.sort_indices()
    x = rng.random((V, Fdim), dtype=np.float32)
    deg = np.diff(csc.indptr).astype(np.float32)
    with np.errstate(divide="ignore"):
        norm = (1.0 / deg).reshape(-1, 1).astype(np.float32)            # inf for isolated vertices, as the reference
    got = preprocess_features(csc, torch.from_numpy(x), torch.from_numpy(norm), chunk_rows=1500).numpy()
    ones = spsp.csc_matrix((np.ones(csc.nnz, np.float64), csc.indices, csc.indptr), shape=csc.shape)
    want = np.asarray(ones.T @ x.astype(np.float64))
    has = deg > 0
    assert np.allclose(got[has], (want[has] * norm[has]).astype(np.float32), rtol=1e-5, atol=1e-5)
    assert np.all(np.isnan(got[~has]))                                   # 0 * inf, exactly what the reference computes


@pytest.mark.gpu
@pytest.mark.parametrize("n,C,ignored", [(6000, 60, 0), (6000, 60, 311), (1000, 41, 0), (777, 200, 50), (3, 5, 0), (1, 2, 0)])
def test_loss_head_vs_oracle(dev, hiplib, oracle, n, C, ignored):
    """pg_xent_fwd / pg_xent_bwd (the trainers' CrossEntropyLoss, pa_gcn.py:80) vs the float64 restatement and
    vs torch's own op: loss and d loss / d logits within 1e-4; padded seeds (label -100) contribute nothing."""
    from pagraph_amd import ops
    g = torch.Generator().manual_seed(n * 31 + C)
    logits = ((torch.rand((n, C), generator=g) - 0.5) * 8).to(dev).requires_grad_(True)
    labels = torch.randint(0, C, (n,), generator=g)
    if ignored:
        labels[torch.randperm(n, generator=g)[:ignored]] = -100
    labels = labels.to(dev)
    loss = ops.cross_entropy(logits, labels)
    assert "CrossEntropy" in type(loss.grad_fn).__name__
    (loss * 3.0).backward()
    ref_loss, ref_grad = oracle.cross_entropy(logits.detach().cpu().numpy(), labels.cpu().numpy())
    assert abs(float(loss) - ref_loss) < TOL * max(1.0, abs(ref_loss))
    assert np.abs(logits.grad.cpu().numpy() - 3.0 * ref_grad).max() < TOL * max(1e-3, np.abs(ref_grad).max() * 3)
    lt = logits.detach().clone().requires_grad_(True)
    tl = torch.nn.functional.cross_entropy(lt, labels)
    (tl * 3.0).backward()
    assert abs(float(tl) - float(loss)) < 1e-5 * max(1.0, abs(float(tl)))
    assert float((lt.grad - logits.grad).abs().max()) < 1e-6
    # deterministic: a second call gives the same bits
    assert float(ops.cross_entropy(logits.detach(), labels)) == float(loss)
    # the trainers swap a plain torch CrossEntropyLoss for the HIP head, and nothing else
    assert isinstance(ops.fused_loss(torch.nn.CrossEntropyLoss()), ops.CrossEntropyLoss)
    w = torch.nn.CrossEntropyLoss(reduction='sum')
    assert ops.fused_loss(w) is w
    # all rows ignored -> nan, like torch
    assert torch.isnan(ops.cross_entropy(logits.detach(), torch.full_like(labels, -100)))


@pytest.mark.gpu
@pytest.mark.parametrize("n_dst,n_src,deg,dim,reduce,p", [(500, 900, 2, 600, "mean", 0.5), (6000, 12000, 2, 64, "mean", 0.5),
                                                           (300, 400, 4, 1100, "sum", 0.1), (64, 64, 3, 8, "mean", 0.9),
                                                           (200, 300, 2, 2048, "mean", 0.25)])
def test_spmm_with_fused_dropout_vs_oracle(dev, hiplib, oracle, n_dst, n_src, deg, dim, reduce, p):
    """pg_spmm_fwd_drop / pg_spmm_bwd = aggregate(dropout(h)): the keep-mask is BIT exact against the
    oracle's restatement of the counter-based spec (pg_dropout_t), values within 1e-4; a new step draws a new
    mask; threshold 0 is the plain aggregation."""
    from pagraph_amd import ops
    rng = np.random.default_rng(n_dst * 7 + dim)
    seed, tag = 0x1234_5678_9ABC_DEF0, 3
    step = torch.tensor([41], dtype=torch.int64, device=dev)
    spec = ops.DropoutSpec(p, seed, tag, step)
    assert spec.threshold == oracle.dropout_threshold(p)
    # (1) the mask itself: identity block, h = 1, sum -> out = mask * scale exactly
    n = 257
    ident_ip = torch.arange(n + 1, dtype=torch.int32, device=dev)
    ident_src = torch.arange(n, dtype=torch.int32, device=dev)
    out = ops.block_aggregate(ident_ip, ident_src, torch.ones((n, dim), device=dev), n, "sum", dropout=spec)
    keep, scale = oracle.dropout_mask(n, dim, spec.threshold, seed, tag, 41)
    assert np.array_equal(out.cpu().numpy(), keep.astype(np.float32) * scale)
    assert abs(keep.mean() - (1 - p)) < 0.02
    # (2) random block, forward and backward
    cnt = rng.integers(0, deg + 1, n_dst)
    indptr = np.zeros(n_dst + 1, np.int32); indptr[1:] = np.cumsum(cnt)
    src = rng.integers(0, n_src, int(indptr[-1])).astype(np.int32)
    h = rng.standard_normal((n_src, dim)).astype(np.float32)
    keep, scale = oracle.dropout_mask(n_src, dim, spec.threshold, seed, tag, 41)
    hd = np.where(keep, h * scale, np.float32(0)).astype(np.float32)
    want = oracle.spmm_fwd(indptr, src, hd, n_dst, reduce)
    th = torch.from_numpy(h).to(dev).requires_grad_(True)
    tip, tsr = torch.from_numpy(indptr).to(dev), torch.from_numpy(src).to(dev)
    out = ops.block_aggregate(tip, tsr, th, n_dst, reduce, dropout=spec)
    assert np.allclose(out.detach().cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    go = rng.standard_normal((n_dst, dim)).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    want_g = oracle.spmm_bwd(indptr, src, go, n_src, reduce) * keep * scale
    assert np.allclose(th.grad.cpu().numpy(), want_g, rtol=0, atol=TOL * max(1.0, float(scale)))
    # (3) the step counter changes the mask; threshold 0 = no dropout
    step.add_(1)
    out2 = ops.block_aggregate(tip, tsr, th.detach(), n_dst, reduce, dropout=spec)
    keep2, _ = oracle.dropout_mask(n_src, dim, spec.threshold, seed, tag, 42)
    assert (keep2 != keep).mean() > 0.05
    assert np.allclose(out2.cpu().numpy(), oracle.spmm_fwd(indptr, src, np.where(keep2, h * scale, np.float32(0)).astype(np.float32), n_dst, reduce), rtol=1e-6, atol=1e-6)
    plain = ops.block_aggregate(tip, tsr, th.detach(), n_dst, reduce, dropout=ops.DropoutSpec(0.0, seed, tag, step))
    assert np.array_equal(plain.cpu().numpy(), oracle.spmm_fwd(indptr, src, h, n_dst, reduce))


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["gcn", "sage"])
def test_models_fold_dropout_into_aggregation(dev, hiplib, oracle, arch):
    """training-mode forward with dropout: no nn.Dropout kernel runs on the aggregated inputs (the fused
    path is taken), two consecutive forwards differ (step counter), eval mode is the deterministic forward,
    and the result equals the same model with dropout applied OUTSIDE through the oracle's mask."""
    from pagraph_amd import ops
    from pagraph_amd.model import GCNSampling, GraphSageSampling
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.data import synthetic as syn
    V, E, B, Fd, C = 20000, 200000, 512, 64, 7
    ip, ix = syn.rmat_graph(V, E, seed=5, device=dev)
    g = DeviceGraph.from_csc(ip, ix, V)
    torch.manual_seed(1234)
    if arch == "gcn":
        model = GCNSampling(Fd, 16, C, 1, torch.relu, 0.5).to(dev)
    else:
        model = GraphSageSampling(Fd, 16, C, 1, torch.relu, 0.5, 'mean').to(dev)
    smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=False, num_hops=2,
                          seed_nodes=torch.arange(4 * B, device=dev), prefetch=True, seed=3)
    nf = next(iter(smp))
    feats = syn.random_features_device(V, Fd, seed=9, device=dev)

    def load():
        for i in range(nf.num_layers):
            nf.layers[i].data.clear()
            nf.layers[i].data['features'] = feats[nf.layer_parent_nid(i)]

    model.train()
    calls = []
    orig = torch.nn.Dropout.forward
    torch.nn.Dropout.forward = lambda self, x: (calls.append(tuple(x.shape)), orig(self, x))[1]
    try:
        load(); y1 = model(nf)
        load(); y2 = model(nf)
    finally:
        torch.nn.Dropout.forward = orig
    assert calls == []                                   # nn.Dropout never ran: the aggregation applied it
    assert int(model._drop_step) == 2
    assert float((y1 - y2).abs().max()) > 1e-3
    # the same forward with the fusion off and nn.Dropout replaced by the oracle's mask for step 3
    step_now = 3
    seed = model._drop_seed

    class _OracleDrop(torch.nn.Dropout):
        tags = []

        def forward(self, x):
            if not self.training:
                return x
            tag = _OracleDrop.tags.pop(0)
            keep, scale = oracle.dropout_mask(x.size(0), x.size(1), oracle.dropout_threshold(self.p), seed, tag, step_now)
            return x * torch.from_numpy(keep.astype(np.float32) * scale).to(x.device)

    load(); y3 = model(nf)                               # fused, step 3
    model.fuse_dropout = False
    model.dropout = _OracleDrop(0.5)
    _OracleDrop.tags = [0, 1] if arch == "gcn" else [0, 1, 17]      # (lid * 16 + i) call sites, in call order
    load(); y4 = model(nf)
    assert float((y3 - y4).abs().max()) < TOL * max(1.0, float(y4.abs().max()))
    model.eval()
    load(); e1 = model(nf)
    load(); e2 = model(nf)
    assert torch.equal(e1, e2)


def _transpose_ref(indptr, src, n_src):
    """source-major copy of a destination-major block: (tptr, tdst), destinations ascending per source"""
    indptr = np.asarray(indptr, np.int64)
    dst = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    order = np.lexsort((dst, src))
    tptr = np.zeros(n_src + 1, np.int64)
    np.add.at(tptr, np.asarray(src, np.int64) + 1, 1)
    return np.cumsum(tptr), dst[order]


@pytest.mark.gpu
@pytest.mark.parametrize("static", [False, True, "deferred"])
def test_sampler_emits_source_major_blocks(dev, hiplib, static):
    """sampler option `transpose`: blocks >= 1 (and on request block 0) also come out source-major, exactly the
    stable transposition of the destination-major block; padded rows are empty"""
    from pagraph_amd.data import synthetic as syn
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    V, E, B = 50000, 600000, 1000
    ip, ix = syn.rmat_graph(V, E, seed=21, device=dev)
    g = DeviceGraph.from_csc(ip, ix, V)
    # 900 copies of one seed: its neighbours become hubs (what utils.py:34 does to isolated train vertices)
    seeds = torch.cat([torch.arange(3 * B + 17 - 900), torch.full((900,), 12345)])
    deferred = static == "deferred"          # the consumer builds the copies later, on a stream of its own
    static = bool(static)
    smp = NeighborSampler(g, B, 3, neighbor_type='in', shuffle=True, num_hops=3, seed_nodes=seeds,
                          prefetch=True, seed=5, static=static, transpose=(0, 1, 2), defer_transpose=deferred)
    side = torch.cuda.Stream()
    auto = NeighborSampler(g, B, 3, neighbor_type='in', num_hops=3, seed_nodes=torch.arange(B), static=static)
    nfa = next(iter(auto))
    assert nfa.blk_tptr[0] is None and nfa.blk_tptr[1] is not None and nfa.blk_tptr[2] is not None
    seen = hubs = 0
    for nf in smp:
        if deferred:
            side.wait_event(nf._slot.ready)
            smp.transpose_blocks(nf, side)
        torch.cuda.synchronize()
        sizes, edges = nf.actual_sizes() if static else ([nf.layer_size(i) for i in range(4)], [nf.block_size(i) for i in range(3)])
        for b in range(3):
            ipb = nf.blk_indptr[b].cpu().numpy()[:sizes[b + 1] + 1]
            srb = nf.blk_src[b].cpu().numpy()[:edges[b]]
            tptr, tdst = _transpose_ref(ipb, srb, sizes[b])
            got_p = nf.blk_tptr[b].cpu().numpy()
            assert np.array_equal(got_p[:sizes[b] + 1], tptr)
            assert np.all(got_p[sizes[b]:] == edges[b])                  # padding rows are empty
            assert np.array_equal(nf.blk_tdst[b].cpu().numpy()[:edges[b]], tdst)
            hv = nf.blk_theavy[b].cpu().numpy()
            want_hv = np.nonzero(np.diff(tptr) > 32)[0]
            assert hv[0] == len(want_hv) and np.array_equal(np.sort(hv[1:1 + hv[0]]), want_hv)
            hubs += len(want_hv)
        seen += 1
    assert seen == 4 and hubs > 0


@pytest.mark.gpu
@pytest.mark.parametrize("n_dst,n_src,deg,dim,reduce,p", [(6000, 17000, 2, 64, "mean", 0.5), (6000, 17000, 2, 64, "mean", 0.0),
                                                           (300, 200, 5, 33, "sum", 0.0), (500, 900, 3, 600, "mean", 0.25),
                                                           (100, 4000, 2, 16, "sum", 0.5), (50, 60, 0, 32, "mean", 0.5),
                                                           (6000, 9000, -2, 64, "mean", 0.5), (3000, 500, -3, 600, "sum", 0.0),
                                                           (2000, 300, -2, 33, "mean", 0.0)])
def test_spmm_backward_gather_form_vs_oracle(dev, hiplib, oracle, n_dst, n_src, deg, dim, reduce, p):
    """pg_spmm_bwd (gather form) (no atomics, no zero fill) == the scatter-form gradient of the oracle, with and
    without the folded dropout mask; two runs are bit-identical (fixed summation order)"""
    from pagraph_amd import ops
    rng = np.random.default_rng(n_dst + 3 * dim)
    hubs = deg < 0                                  # negative degree: 15 % of the edges go to three hub sources
    deg = abs(deg)
    cnt = rng.integers(0, deg + 1, n_dst) if deg else np.zeros(n_dst, np.int64)
    indptr = np.zeros(n_dst + 1, np.int32); indptr[1:] = np.cumsum(cnt)
    src = rng.integers(0, n_src, int(indptr[-1])).astype(np.int32)
    if hubs:
        pick = rng.random(src.size) < 0.15
        src[pick] = rng.choice(np.array([7, n_src - 1, n_src // 2], np.int32), int(pick.sum()))
    tptr, tdst = _transpose_ref(indptr, src, n_src)
    h = rng.standard_normal((n_src, dim)).astype(np.float32)
    go = rng.standard_normal((n_dst, dim)).astype(np.float32)
    step = torch.tensor([9], dtype=torch.int64, device=dev)
    spec = ops.DropoutSpec(p, 77, 5, step) if p else None
    tip, tsr = torch.from_numpy(indptr).to(dev), torch.from_numpy(src).to(dev)
    heavy_rows = np.nonzero(np.diff(tptr) > 32)[0]
    assert (len(heavy_rows) == 3) == hubs
    heavy = np.zeros(1 + max(1, src.size // 32), np.int32)
    heavy[0] = len(heavy_rows); heavy[1:1 + len(heavy_rows)] = heavy_rows[::-1]
    tr = (torch.from_numpy(tptr.astype(np.int32)).to(dev), torch.from_numpy(tdst.astype(np.int32)).to(dev),
          torch.from_numpy(heavy).to(dev))
    grads = []
    for _ in range(2):
        th = torch.from_numpy(h).to(dev).requires_grad_(True)
        out = ops.block_aggregate(tip, tsr, th, n_dst, reduce, dropout=spec, transpose=tr)
        out.backward(torch.from_numpy(go).to(dev))
        grads.append(th.grad.cpu().numpy())
    want = oracle.spmm_bwd(indptr, src, go, n_src, reduce)
    scale = 1.0
    if p:
        keep, scale = oracle.dropout_mask(n_src, dim, spec.threshold, 77, 5, 9)
        want = want * keep * scale
    assert np.allclose(grads[0], want, rtol=0, atol=TOL * max(1.0, float(scale)))
    assert np.array_equal(grads[0], grads[1])
    # and it agrees with the scatter form of the same library
    th = torch.from_numpy(h).to(dev).requires_grad_(True)
    ops.block_aggregate(tip, tsr, th, n_dst, reduce, dropout=spec).backward(torch.from_numpy(go).to(dev))
    assert np.allclose(th.grad.cpu().numpy(), grads[0], rtol=0, atol=TOL * max(1.0, float(scale)))


@pytest.mark.gpu
@pytest.mark.parametrize("n_dst,n_src,deg,dim,p", [(6000, 17000, 2, 64, 0.0), (6000, 17000, 2, 64, 0.5), (500, 900, 3, 600, 0.25),
                                                   (300, 200, 5, 33, 0.0), (300, 200, 5, 602, 0.0), (50, 60, 0, 32, 0.5),
                                                   (6000, 9000, -2, 64, 0.5), (3000, 500, -3, 600, 0.0), (2000, 300, -2, 33, 0.0)])
def test_spmm_max_reducer_vs_oracle(dev, hiplib, oracle, n_dst, n_src, deg, dim, p):
    """a-10's third reducer (graphsage_nssc.py:106-110, fn.max = the 'pool' aggregator): PG_REDUCE_MAX forward is BIT
    exact against the oracle (zeros for a destination without in-edges); the backward — every in-edge whose message
    equals the maximum receives the destination's gradient, DGL's `val == accum` — in scatter form (pg_spmm_bwd (max))
    and in gather form over the source-major copy (pg_spmm_bwd (max, gather form), hubs included, bit-identical between runs),
    with and without the folded dropout. Values come from a small integer set, so ties are everywhere."""
    from pagraph_amd import ops
    rng = np.random.default_rng(n_dst + 5 * dim + int(p * 10))
    hubs = deg < 0
    deg = abs(deg)
    cnt = rng.integers(0, deg + 1, n_dst) if deg else np.zeros(n_dst, np.int64)
    indptr = np.zeros(n_dst + 1, np.int32); indptr[1:] = np.cumsum(cnt)
    src = rng.integers(0, n_src, int(indptr[-1])).astype(np.int32)
    if hubs:
        pick = rng.random(src.size) < 0.15
        src[pick] = rng.choice(np.array([7, n_src - 1, n_src // 2], np.int32), int(pick.sum()))
    tptr, tdst = _transpose_ref(indptr, src, n_src)
    h = rng.integers(-2, 3, (n_src, dim)).astype(np.float32)          # five values: ties in most columns
    h[rng.random((n_src, dim)) < 0.3] += np.float32(0.25)
    go = rng.standard_normal((n_dst, dim)).astype(np.float32)
    step = torch.tensor([9], dtype=torch.int64, device=dev)
    spec = ops.DropoutSpec(p, 77, 5, step) if p else None
    x, keep, scale = h, None, 1.0
    if p:
        keep, scale = oracle.dropout_mask(n_src, dim, spec.threshold, 77, 5, 9)
        x = np.where(keep, h * scale, np.float32(0)).astype(np.float32)
    want = oracle.spmm_fwd(indptr, src, x, n_dst, "max")
    assert np.all(want[cnt == 0] == 0)
    want_g = oracle.spmm_bwd_max(indptr, src, go, x, want)
    if p:
        want_g = want_g * keep * scale
    tip, tsr = torch.from_numpy(indptr).to(dev), torch.from_numpy(src).to(dev)
    heavy_rows = np.nonzero(np.diff(tptr) > 32)[0]
    heavy = np.zeros(1 + max(1, src.size // 32), np.int32)
    heavy[0] = len(heavy_rows); heavy[1:1 + len(heavy_rows)] = heavy_rows[::-1]
    tr = (torch.from_numpy(tptr.astype(np.int32)).to(dev), torch.from_numpy(tdst.astype(np.int32)).to(dev),
          torch.from_numpy(heavy).to(dev))
    if p and dim % 4:
        with pytest.raises(Exception):
            ops.block_aggregate(tip, tsr, torch.from_numpy(h).to(dev), n_dst, "max", dropout=spec)
        return
    grads = []
    for transpose in (tr, tr, None):
        th = torch.from_numpy(h).to(dev).requires_grad_(True)
        out = ops.block_aggregate(tip, tsr, th, n_dst, "max", dropout=spec, transpose=transpose)
        assert np.array_equal(out.detach().cpu().numpy(), want)
        out.backward(torch.from_numpy(go).to(dev))
        grads.append(th.grad.cpu().numpy())
    assert np.array_equal(grads[0], grads[1])                                    # gather form: fixed summation order
    for g in grads:
        assert np.allclose(g, want_g, rtol=0, atol=TOL * max(1.0, float(scale)))
    # raw C-ABI: the max reducer's backward needs the forward's input and output
    gh = torch.zeros((n_src, dim), device=dev)
    tgo = torch.from_numpy(go).to(dev)
    assert ops.spmm_bwd_call(hiplib, tgo, gh, n_src, "max", indptr=tip, src=tsr) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("n,K1,K2,N,act", [(17000, 600, 600, 16, 2), (6000, 32, 32, 60, 0), (5000, 600, 64, 32, 1),
                                           (2049, 8, 600, 41, 0), (4096, 64, 64, 64, 2)])
def test_dual_linear_vs_torch(dev, hiplib, n, K1, K2, N, act):
    """pg_linear_fwd (two operands) (GraphSAGE NodeUpdate: fc_self(h) + fc_neigh(neigh), activation / skip-concat fused) and
    its backward vs float64 torch: output, both weight / bias gradients and both input gradients within 1e-4"""
    from pagraph_amd import ops
    torch.manual_seed(n + K1 + N)
    l1, l2 = torch.nn.Linear(K1, N).to(dev), torch.nn.Linear(K2, N).to(dev)
    x1 = (torch.rand((n, K1), device=dev) - 0.4).requires_grad_(True)
    x2 = (torch.rand((n, K2), device=dev) - 0.6).requires_grad_(True)
    y = ops.linear2(x1, l1, x2, l2, act)
    assert "DualLinear" in type(y.grad_fn).__name__
    d1, d2 = x1.detach().double().requires_grad_(True), x2.detach().double().requires_grad_(True)
    w1, w2 = l1.weight.detach().double().requires_grad_(True), l2.weight.detach().double().requires_grad_(True)
    b1, b2 = l1.bias.detach().double().requires_grad_(True), l2.bias.detach().double().requires_grad_(True)
    z = torch.nn.functional.linear(d1, w1, b1) + torch.nn.functional.linear(d2, w2, b2)
    ref = z if act == 0 else torch.relu(z) if act == 1 else torch.cat((z, torch.relu(z)), 1)
    scale = max(1.0, float(ref.abs().max()))
    assert y.shape == ref.shape and float((y.double() - ref).abs().max()) < TOL * scale
    g = torch.rand_like(y) - 0.5
    y.backward(g)
    ref.backward(g.double())
    for got, want in ((l1.weight.grad, w1.grad), (l2.weight.grad, w2.grad), (l1.bias.grad, b1.grad), (l2.bias.grad, b2.grad),
                      (x1.grad, d1.grad), (x2.grad, d2.grad)):
        assert float((got.double() - want).abs().max()) < TOL * max(1.0, float(want.abs().max()))
    # small inputs go through the modules
    ys = ops.linear2(x1[:100], l1, x2[:100], l2, act)
    assert "DualLinear" not in type(ys.grad_fn).__name__
    assert float((ys.double() - ref[:100]).abs().max()) < TOL * scale


@pytest.mark.gpu
@pytest.mark.parametrize("wd", [0.0, 5e-4])
def test_adam_step_matches_torch(dev, hiplib, wd):
    """pagraph_amd.optim.Adam (one pg_adam_step launch) follows torch.optim.Adam's trajectory: 25 steps on the
    GCN's parameter shapes (+ a 17-tensor group to cross the 16-tensor chunk), eager and replayed from a
    hipGraph (the device-side step counter advances per replay)."""
    from pagraph_amd.optim import Adam
    torch.manual_seed(5)
    shapes = [(32, 600), (32,), (60, 64), (60,)] + [(7, 3)] * 13
    ref_p = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    ref = torch.optim.Adam(ref_p, lr=3e-2, weight_decay=wd)
    our = Adam(our_p, lr=3e-2, weight_decay=wd)
    grads = [[torch.randn(s, device=dev) * (0.1 + 0.05 * t) for s in shapes] for t in range(25)]
    for t in range(25):
        for p, q, g in zip(ref_p, our_p, grads[t]):
            p.grad = g.clone()
            q.grad = g.clone()
        ref.step()
        our.step()
        err = max(float((p - q).abs().max()) for p, q in zip(ref_p, our_p))
        assert err < 2e-6, (t, err)
    assert int(our.state[our_p[0]]['step']) == 25
    # captured: static gradient buffers refreshed before every replay
    cap_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p[:4]]
    ref2_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p[:4]]
    cap = Adam(cap_p, lr=1e-2, weight_decay=wd)
    ref2 = torch.optim.Adam(ref2_p, lr=1e-2, weight_decay=wd)
    static_g = [torch.zeros(s, device=dev) for s in shapes[:4]]
    for p, g in zip(cap_p, static_g):
        p.grad = g
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        cap.step()                                   # warm-up (state allocation) outside the capture
        for p, g in zip(ref2_p, static_g):
            p.grad = g.clone()
        ref2.step()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            cap.step()
        for t in range(10):
            for g, src, q in zip(static_g, grads[t][:4], ref2_p):
                g.copy_(src)
                q.grad = src.clone()
            graph.replay()
            ref2.step()
    torch.cuda.synchronize()
    assert max(float((p - q).abs().max()) for p, q in zip(ref2_p, cap_p)) < 2e-6
    assert int(cap.state[cap_p[0]]['step']) == 11


@pytest.mark.gpu
@pytest.mark.parametrize("n_dst,n_src,deg,K,C,p,ignored,hubs", [(6000, 9500, 2, 64, 60, 0.5, 0, True), (6000, 9500, 2, 64, 60, 0.0, 300, True),
                                                                  (1000, 700, 3, 32, 41, 0.25, 10, False), (33, 20, 1, 64, 7, 0.0, 0, False),
                                                                  (500, 2000, 4, 16, 64, 0.9, 0, False)])
def test_gcn_output_head_vs_oracle(dev, hiplib, oracle, n_dst, n_src, deg, K, C, p, ignored, hubs):
    """pg_head (aggregation + dropout + output linear layer + CrossEntropyLoss + all gradients in one pass)
    vs the float64 restatement and vs the unfused ops of this library; gather-form and scatter-form backward;
    bit-identical between two runs."""
    from pagraph_amd import ops
    rng = np.random.default_rng(n_dst + K * C)
    cnt = rng.integers(0, deg + 1, n_dst)
    indptr = np.zeros(n_dst + 1, np.int32); indptr[1:] = np.cumsum(cnt)
    src = rng.integers(0, n_src, int(indptr[-1])).astype(np.int32)
    if hubs:
        pick = rng.random(src.size) < 0.15
        src[pick] = rng.choice(np.array([3, n_src - 2], np.int32), int(pick.sum()))
    tptr, tdst = _transpose_ref(indptr, src, n_src)
    heavy_rows = np.nonzero(np.diff(tptr) > 32)[0]
    heavy = np.zeros(1 + max(1, src.size // 32), np.int32)
    heavy[0] = len(heavy_rows); heavy[1:1 + len(heavy_rows)] = heavy_rows
    h = rng.standard_normal((n_src, K)).astype(np.float32)
    labels = rng.integers(0, C, n_dst)
    if ignored:
        labels[rng.permutation(n_dst)[:ignored]] = -100
    lin = torch.nn.Linear(K, C).to(dev)
    tip, tsr = torch.from_numpy(indptr).to(dev), torch.from_numpy(src).to(dev)
    tr = tuple(torch.from_numpy(a.astype(np.int32)).to(dev) for a in (tptr, tdst, heavy))
    tl = torch.from_numpy(labels).to(dev)
    n_valid = torch.tensor([int((labels != -100).sum())], dtype=torch.int32, device=dev)
    seed_t = torch.tensor(0.5, device=dev)                       # d objective / d loss (1 / world_size 2)
    step = torch.tensor([4], dtype=torch.int64, device=dev)
    spec = ops.DropoutSpec(p, 99, 1, step) if p else None
    keep, scale = oracle.dropout_mask(n_src, K, spec.threshold, 99, 1, 4) if p else (None, 1.0)
    want = oracle.gcn_head(indptr, src, h, lin.weight.detach().cpu().numpy(), lin.bias.detach().cpu().numpy(), labels,
                           -100, 0.5, "mean", keep, scale)
    results = []
    for transpose in (tr, None, tr):
        lin.zero_grad()
        th = torch.from_numpy(h).to(dev).requires_grad_(True)
        loss, logits = ops.gcn_head(tip, tsr, th, lin, tl, n_valid, seed_t, -100, "mean", spec, transpose, want_logits=True)
        assert "GCNHead" in type(loss.grad_fn).__name__
        loss.backward(seed_t)
        got = (float(loss), logits.cpu().numpy(), th.grad.cpu().numpy(), lin.weight.grad.cpu().numpy(), lin.bias.grad.cpu().numpy())
        results.append(got)
        sc = max(1.0, float(scale))
        assert abs(got[0] - want[0]) < TOL * max(1.0, abs(want[0]))
        assert np.abs(got[1] - want[1]).max() < TOL * max(1.0, np.abs(want[1]).max())
        for g, w in zip(got[2:], want[2:]):
            assert np.abs(g - w).max() < TOL * sc * max(1e-3, np.abs(w).max()), (np.abs(g - w).max(), np.abs(w).max())
    for a, b in zip(results[0], results[2]):                      # deterministic
        assert np.array_equal(a, b)
    # a gradient seed other than the registered one is honoured too (rescaled)
    lin.zero_grad()
    th = torch.from_numpy(h).to(dev).requires_grad_(True)
    loss = ops.gcn_head(tip, tsr, th, lin, tl, n_valid, seed_t, -100, "mean", spec, tr)
    (loss * 3.0).backward()
    assert np.abs(lin.weight.grad.cpu().numpy() - 6.0 * want[3]).max() < TOL * 6 * max(1e-3, np.abs(want[3]).max()) * max(1.0, float(scale))
    # no counted label at all: nan loss (torch's convention), zero gradients
    none_valid = torch.zeros(1, dtype=torch.int32, device=dev)
    l0 = ops.gcn_head(tip, tsr, torch.from_numpy(h).to(dev), lin, torch.full_like(tl, -100), none_valid, seed_t, -100, "mean", spec, tr)
    assert torch.isnan(l0)
    # PG_HEAD_DAGG_PER_EDGE: dagg divided by the destination's in-degree on its way out, exactly
    import ctypes
    from pagraph_amd import _lib as L
    lib = L.load()
    th = torch.from_numpy(h).to(dev)
    outs = []
    for flags in (L.PG_HEAD_SUM_PARTIALS, L.PG_HEAD_SUM_PARTIALS | L.PG_HEAD_DAGG_PER_EDGE):
        buf = torch.empty(C * K + C + 1, device=dev)
        dagg = torch.empty((n_dst, K), device=dev)
        part = torch.empty(lib.pg_gcn_head_scratch(n_dst, K, C), device=dev)
        dstruct = spec.struct() if spec is not None else None
        hd = ops.head_desc(tip, tsr, th, lin.weight, lin.bias, tl, n_valid, seed_t, -100, "mean", dstruct, None, dagg, part, buf,
                           buf[C * K:], flags)
        L.check(lib.pg_head(ctypes.byref(hd), L.stream_ptr()), "pg_head")
        outs.append((dagg, buf))
    deg = torch.from_numpy(np.maximum(cnt, 1).astype(np.float32)).to(dev)[:, None]
    assert torch.equal(outs[1][0], outs[0][0] / deg) and torch.equal(outs[1][1], outs[0][1])
    hd = ops.head_desc(tip, tsr, th, lin.weight, lin.bias, tl, n_valid, seed_t, -100, "mean", None, None, dagg, part, buf, buf[C * K:], 4)
    assert lib.pg_head(ctypes.byref(hd), L.stream_ptr()) == -1                                   # PG_ERR_INVALID: unknown flag
    hd = ops.head_desc(tip, tsr, th, lin.weight, lin.bias, tl, n_valid, seed_t, -100, "mean", None, None, dagg, part, buf, buf[C * K:], 1)
    hd.dself = L.ptr(dagg).value                                                                    # self operand fields without Ks
    assert lib.pg_head(ctypes.byref(hd), L.stream_ptr()) == -1
    # and the unfused ops of the library agree
    lin.zero_grad()
    th2 = torch.from_numpy(h).to(dev).requires_grad_(True)
    agg = ops.block_aggregate(tip, tsr, th2, n_dst, "mean", dropout=spec, transpose=tr)
    l2 = ops.cross_entropy(torch.nn.functional.linear(agg, lin.weight, lin.bias), tl)
    l2.backward(seed_t)
    assert abs(float(l2) - results[0][0]) < 1e-5 * max(1.0, abs(float(l2)))
    assert np.abs(th2.grad.cpu().numpy() - results[0][2]).max() < TOL * max(1e-3, np.abs(results[0][2]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("n_dst,n_src,N,p,consumer", [(6000, 9500, 32, 0.5, "head"), (6000, 9500, 32, 0.0, "agg"),
                                                       (3000, 4000, 16, 0.25, "agg"), (2500, 3100, 8, 0.0, "head"),
                                                       (1500, 2000, 24, 0.5, "head")])
def test_dz_from_backward_gather_bit_identical(dev, hiplib, monkeypatch, n_dst, n_src, N, p, consumer):
    """pg_spmm_bwd (gather form + dZ): the skip-concat NodeUpdate's dZ written by the aggregation's backward (regular rows through
    the lane-group shuffle, hub rows through LDS) == the unfused k_dz path bit for bit, in dZ's consumers: the layer's
    weight / bias gradient and the gradient of its input. A gradient that is not the stashed tensor misses the stash."""
    from pagraph_amd import ops
    rng = np.random.default_rng(n_dst + N)
    cnt = rng.integers(0, 4, n_dst)
    indptr = np.zeros(n_dst + 1, np.int32); indptr[1:] = np.cumsum(cnt)
    src = rng.integers(0, n_src, int(indptr[-1])).astype(np.int32)
    pick = rng.random(src.size) < 0.2
    src[pick] = rng.choice(np.array([5, n_src - 3, n_src // 2], np.int32), int(pick.sum()))     # three hubs
    tptr, tdst = _transpose_ref(indptr, src, n_src)
    heavy_rows = np.nonzero(np.diff(tptr) > 32)[0]
    assert len(heavy_rows) == 3
    heavy = np.zeros(1 + max(1, src.size // 32), np.int32)
    heavy[0] = len(heavy_rows); heavy[1:1 + len(heavy_rows)] = heavy_rows
    K_in, C = 48, 13
    x = torch.from_numpy(rng.standard_normal((n_src, K_in)).astype(np.float32)).to(dev)
    labels = torch.from_numpy(rng.integers(0, C, n_dst)).to(dev)
    n_valid = torch.tensor([n_dst], dtype=torch.int32, device=dev)
    seed_t = torch.tensor(1.0, device=dev)
    tip, tsr = torch.from_numpy(indptr).to(dev), torch.from_numpy(src).to(dev)
    tr = tuple(torch.from_numpy(a.astype(np.int32)).to(dev) for a in (tptr, tdst, heavy))
    torch.manual_seed(5)
    hidden = torch.nn.Linear(K_in, N).to(dev)
    out = torch.nn.Linear(2 * N, C).to(dev)
    step = torch.tensor([7], dtype=torch.int64, device=dev)
    spec = ops.DropoutSpec(p, 11, 3, step) if p else None

    def run(fuse, twice=False):
        monkeypatch.setattr(ops, "FUSE_DZ", fuse)
        del ops._DZ_STASH[:]
        hidden.zero_grad(); out.zero_grad()
        xi = x.clone().requires_grad_(True)
        y = ops.linear(xi, hidden, ops.ACT_CONCAT)
        assert getattr(y, "_pg_concat_n", 0) == N
        if consumer == "head":
            loss = ops.gcn_head(tip, tsr, y, out, labels, n_valid, seed_t, -100, "mean", spec, tr)
        else:
            agg = ops.block_aggregate(tip, tsr, y, n_dst, "mean", dropout=spec, transpose=tr)
            loss = ops.cross_entropy(torch.nn.functional.linear(agg, out.weight, out.bias), labels)
        if twice:
            loss = loss + 0.5 * y.sum()          # y consumed twice: the NodeUpdate sees a summed gradient
        loss.backward(seed_t if not twice else None)
        left = len(ops._DZ_STASH)
        return [t.detach().cpu().numpy().copy() for t in (xi.grad, hidden.weight.grad, hidden.bias.grad)], left

    plain, left0 = run(False)
    fused, left1 = run(True)
    assert left0 == 0 and left1 == 0                    # produced and consumed
    for a, b in zip(plain, fused):
        assert np.array_equal(a, b)
    assert np.abs(plain[1]).max() > 0
    plain2, _ = run(False, twice=True)
    fused2, left2 = run(True, twice=True)
    assert left2 == 1                                   # stashed, not taken: the gradient was a sum
    for a, b in zip(plain2, fused2):
        assert np.array_equal(a, b)
    del ops._DZ_STASH[:]


@pytest.mark.gpu
def test_gcn_forward_loss_matches_forward_plus_loss(dev, hiplib):
    """GCNSampling.forward_loss == CrossEntropyLoss(model(nf)) in value and in every parameter gradient, with
    dropout (same step counter) and without; inference models and CPU labels decline (None)."""
    from pagraph_amd import ops
    from pagraph_amd.data import synthetic as syn
    from pagraph_amd.model import GCNSampling, GCNInfer
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    V, E, B, Fd, C = 30000, 300000, 1500, 64, 11
    ip, ix = syn.rmat_graph(V, E, seed=8, device=dev)
    g = DeviceGraph.from_csc(ip, ix, V)
    smp = NeighborSampler(g, B, 2, neighbor_type='in', num_hops=2, seed_nodes=torch.arange(2 * B, device=dev), seed=1)
    nf = next(iter(smp))
    feats = syn.random_features_device(V, Fd, seed=2, device=dev)
    labels = torch.randint(0, C, (nf.layer_size(-1),), device=dev)
    labels[::7] = -100
    n_valid = (labels != -100).sum().to(torch.int32).reshape(1)
    for pdrop in (0.0, 0.5):
        torch.manual_seed(3)
        model = GCNSampling(Fd, 16, C, 1, torch.relu, pdrop).to(dev).train()

        def load():
            for i in range(nf.num_layers):
                nf.layers[i].data.clear()
            nf.layers[0].data['features'] = feats[nf.layer_parent_nid(0)]

        load()
        model._drop_step.fill_(10)
        ref = ops.cross_entropy(model(nf), labels)
        ref.backward()
        gref = [p.grad.clone() for p in model.parameters()]
        model.zero_grad()
        load()
        model._drop_step.fill_(10)
        out = model.forward_loss(nf, labels, n_valid, None, -100, want_logits=True)
        assert out is not None
        loss, logits = out
        loss.backward()
        assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
        for p, gr in zip(model.parameters(), gref):
            assert float((p.grad - gr).abs().max()) < TOL * max(1e-3, float(gr.abs().max()))
    infer = GCNInfer(Fd, 16, C, 1, torch.relu).to(dev)
    load()
    assert infer.forward_loss(nf, labels, n_valid) is None
    assert model.forward_loss(nf, labels.cpu(), n_valid) is None


@pytest.mark.parametrize("agg", ["mean", "gcn"])
def test_sage_forward_loss_matches_forward_plus_loss(dev, hiplib, agg):
    """GraphSageSampling.forward_loss (the output NodeUpdate fc_neigh(neigh) + fc_self(h), its aggregation, the loss and all
    their gradients in one kernel: pg_head) == CrossEntropyLoss(model(nf)) in value and in every parameter gradient,
    with dropout (same step counter) and without; 'pool' and CPU labels decline (None)."""
    from pagraph_amd import ops
    from pagraph_amd.data import synthetic as syn
    from pagraph_amd.model import GraphSageSampling
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    V, E, B, Fd, C = 30000, 300000, 1500, 64, 11
    ip, ix = syn.rmat_graph(V, E, seed=8, device=dev)
    g = DeviceGraph.from_csc(ip, ix, V)
    smp = NeighborSampler(g, B, 2, neighbor_type='in', num_hops=2, seed_nodes=torch.arange(2 * B, device=dev), seed=1)
    nf = next(iter(smp))
    feats = syn.random_features_device(V, Fd, seed=2, device=dev)
    labels = torch.randint(0, C, (nf.layer_size(-1),), device=dev)
    labels[::7] = -100
    n_valid = (labels != -100).sum().to(torch.int32).reshape(1)

    def load():
        for i in range(nf.num_layers):
            nf.layers[i].data.clear()
            nf.layers[i].data['features'] = feats[nf.layer_parent_nid(i)]

    for pdrop in (0.0, 0.5):
        torch.manual_seed(3)
        model = GraphSageSampling(Fd, 16, C, 1, torch.relu, pdrop, agg).to(dev).train()
        load()
        model._drop_step.fill_(10)
        ref = ops.cross_entropy(model(nf), labels)
        ref.backward()
        gref = [p.grad.clone() for p in model.parameters()]
        model.zero_grad()
        load()
        model._drop_step.fill_(10)
        loss = model.forward_loss(nf, labels, n_valid, None, -100)
        assert loss is not None
        loss.backward()
        assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
        for (name, p), gr in zip(model.named_parameters(), gref):
            assert p.grad is not None, name
            assert float((p.grad - gr).abs().max()) < TOL * max(1e-3, float(gr.abs().max())), name
    load()
    assert model.forward_loss(nf, labels.cpu(), n_valid) is None
    pool = GraphSageSampling(Fd, 16, C, 1, torch.relu, 0.0, 'pool').to(dev)
    load()
    assert pool.forward_loss(nf, labels, n_valid) is None


# ---- G7 / G8: the HIP models against the reference's own model classes -----------------------------------------
def _nf_from_fixture(z, dev):
    from pagraph_amd.sampling.nodeflow import NodeFlow
    sizes = [int(x) for x in z["layer_sizes"]]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    nf = NodeFlow(torch.arange(int(offs[-1]), device=dev), offs,
                  [torch.from_numpy(z[f"blk{b}_indptr"].astype(np.int32)).to(dev) for b in range(len(sizes) - 1)],
                  [torch.from_numpy(z[f"blk{b}_src"].astype(np.int32)).to(dev) for b in range(len(sizes) - 1)])
    for i in range(len(sizes)):
        nf._node_frames[i] = {k[len(f"layer{i}_"):]: torch.from_numpy(z[k]).to(dev) for k in z.files
                              if k.startswith(f"layer{i}_") and k != "layer_sizes"}
    return nf


def _check_against_reference_model(z, model, dev):
    state = {k[len("param:"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param:")}
    assert set(state) == set(model.state_dict()), "parameter names differ from the reference's"
    model.load_state_dict(state)
    model = model.to(dev).train()
    logits = model(_nf_from_fixture(z, dev))
    want = z["logits"]
    assert tuple(logits.shape) == want.shape
    err = np.abs(logits.detach().cpu().numpy() - want).max()
    assert err <= TOL * max(1.0, np.abs(want).max()), err
    (logits * torch.from_numpy(z["G"]).to(dev)).sum().backward()
    for name, p in model.named_parameters():
        g_want = z[f"grad:{name}"]
        g_got = (p.grad if p.grad is not None else torch.zeros_like(p)).cpu().numpy()
        assert np.abs(g_got - g_want).max() <= TOL * max(1.0, np.abs(g_want).max()), name


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "g7_*.npz"))), ids=os.path.basename)
def test_gcn_models_vs_reference_golden(dev, hiplib, path):
    """GCNSampling / GCNInfer (HIP aggregation + fp32-MFMA dense step) == logits and parameter gradients of the
    reference's gcn_nssc.py classes (G7), 1e-4"""
    from pagraph_amd.model import GCNInfer, GCNSampling
    z = np.load(path)
    a = (int(z["in_feats"]), int(z["n_hidden"]), int(z["n_classes"]), int(z["n_layers"]), torch.nn.functional.relu)
    if str(z["arch"]) == "gcn_infer":
        model = GCNInfer(*a, preprocess=bool(z["preprocess"]))
    else:
        model = GCNSampling(*a, 0.0, preprocess=bool(z["preprocess"]))
    _check_against_reference_model(z, model, dev)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "g8_*.npz"))), ids=os.path.basename)
def test_sage_models_vs_reference_golden(dev, hiplib, path):
    """GraphSageSampling == the reference's graphsage_nssc.py class (G8), 1e-4"""
    from pagraph_amd.model import GraphSageSampling
    z = np.load(path)
    model = GraphSageSampling(int(z["in_feats"]), int(z["n_hidden"]), int(z["n_classes"]), int(z["n_layers"]),
                              torch.nn.functional.relu, 0.0, str(z["aggregator"]), bool(z["preprocess"]))
    _check_against_reference_model(z, model, dev)


# ---- a-13: the dg partitioner is host C++ inside the product library; its parity tests live in test_host_logic.py
# (CPU suite) and are re-run here so that the GPU tier executes them against the library it loaded -------------
from tests import test_host_logic as _host   # noqa: E402


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "g4_*.npz"))), ids=os.path.basename)
@pytest.mark.parametrize("device", ["cpu", "cuda"])
def test_dg_vs_reference_golden_on_gpu_box(hiplib, path, device):
    """device='cuda': pg_dg_partition_gpu (round 6) against the reference's own outputs — every G4 fixture has P <= 16 and
    hops <= 2, the envelope of the device-assisted path"""
    _host.test_dg_product_vs_reference_golden(hiplib, path, device=device)


@pytest.mark.parametrize("V,E,P,hops", [(1_000_000, 10_000_000, 4, 2), (1_000_000, 10_000_000, 8, 1), (300_000, 3_000_000, 16, 2),
                                        (5000, 40000, 3, 2)])
def test_dg_gpu_equals_the_sequential_host_code(dev, hiplib, V, E, P, hops):
    """pg_dg_partition_gpu == pg_dg_partition_mt bit for bit on RMAT graphs with hubs (the first train vertices' two-hop sets
    are most of the graph: batches of one, redone batches, the lists' buffers near their limits) — belongs, r_belongs,
    p_vnum, r_vnum (dg.py:59-103)"""
    from pagraph_amd.data import synthetic as syn
    import importlib
    dgmod = importlib.import_module("pagraph_amd.partition.dg")
    indptr, indices = syn.rmat_graph(V, E, device=dev)
    train_mask, _, _ = syn.split_dataset(V)
    train = torch.nonzero(train_mask).squeeze(1).numpy()
    a = dgmod.dg_raw(P, indptr, indices, V, train, hops, device="cuda")
    st = dict(dgmod.LAST_GPU_STATS)
    b = dgmod.dg_raw(P, indptr, indices, V, train, hops, device="cpu")
    assert dgmod.LAST_GPU_STATS is None
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[2].sum() == len(train) and st["batches"] >= 1 and st["largest_batch"] >= min(len(train), 64)
    # both list paths ran: most multisets' put-aside members fit the workgroup's scratch (V / 256 entries) and are walked once,
    # the early ones (every partition lacks everything) do not and are walked again under generations of their own; at 10^6
    # vertices every workgroup's generation tag wraps as well (a run starts 240 multisets short of the wrap)
    assert st["second_walks"] < len(train) and (st["second_walks"] > 0 or hops == 1)      # (one hop: a list is a vertex's degree)


@pytest.mark.parametrize("V,E,P,hops", [(3000, 20000, 4, 1), (3000, 12000, 8, 2), (1500, 6000, 3, 3), (2000, 9000, 16, 2)])
def test_dg_vs_oracle_medium_on_gpu_box(hiplib, oracle, V, E, P, hops):
    _host.test_dg_product_vs_oracle_medium(hiplib, oracle, V, E, P, hops)


def test_auto_cache_never_outgrows_free_memory(dev, hiplib, monkeypatch):
    """the reference's rule budgets capability * total_dim * 4 bytes; padded rows + the fill's device staging need more.
    With little free memory the cache (and the fill chunk) shrink instead of running out of memory."""
    import types
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    rng = np.random.default_rng(3)
    V, Fd = 40000, 600
    feats = rng.random((V, Fd), dtype=np.float32)
    norm = rng.random((V, 1), dtype=np.float32)
    c = GraphCacheServer(HostFeatureStore({"features": torch.from_numpy(feats), "norm": torch.from_numpy(norm)}), V,
                         torch.arange(V), 0)
    c.init_field(["features", "norm"])
    stride_bytes = c._row_stride(601) * 4
    assert stride_bytes == 608 * 4
    # pretend 30 000 rows' worth of the reference's budget is free: 30 000 * 2404 B (+ the 1 GiB reserve)
    free = 30000 * 601 * 4 + (1 << 30)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda device=None: (free, 1 << 40))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda device=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda device=None: 0)
    g = types.SimpleNamespace(out_degrees=lambda: torch.arange(V))
    c.auto_cache(g, ["features", "norm"])
    assert not c.full_cached and 0 < c.cached_num < 30000
    # what was allocated fits the pretend budget: padded rows + the largest staging chunk the fill used
    fit, chunk = c._physical_fit(30000, ["features", "norm"])
    assert c.cached_num * stride_bytes + chunk * 601 * 4 <= free - (1 << 30)
    want = torch.argsort(torch.arange(V), descending=True)[:c.cached_num]
    assert torch.equal(torch.nonzero(c.gpu_flag.cpu()).squeeze(1), torch.sort(want).values)
    got = c.gpu_fix_cache["features"][c.localid2cacheid[want.to(dev)]].cpu().numpy()
    assert np.array_equal(got, feats[want.numpy()])


# ---- f-2: gather fused into the layer-0 aggregation (pg_split_rows + pg_spmm_fwd_rows) -------------------------
@pytest.mark.parametrize("ratio,p_drop,reduce", [(0.3, 0.0, "mean"), (0.3, 0.25, "mean"), (0.0, 0.25, "sum"),
                                                 (1.0, 0.0, "mean"), (0.6, 0.5, "sum"), (0.3, 0.0, "max"), (0.5, 0.25, "max")])
def test_fused_gather_aggregate_vs_oracle(dev, hiplib, oracle, ratio, p_drop, reduce):
    """pg_split_rows + pg_spmm_fwd_rows through the raw C-ABI: hits read from the cache, misses from a staged block in
    miss-list order, dropout keep-mask by source position — equal BIT FOR BIT to the oracle's gather -> dropout ->
    aggregate (and therefore to the unfused pg_gather_rows + pg_spmm_fwd_drop pair)"""
    _fused_gather_aggregate_case(dev, hiplib, oracle, ratio, p_drop, reduce, 600)


@pytest.mark.gpu
@pytest.mark.parametrize("Fd,p_drop", [(256, 0.25), (500, 0.25), (512, 0.0), (768, 0.25), (772, 0.25), (1024, 0.5),
                                       (1028, 0.25)])
def test_fused_gather_aggregate_row_widths(dev, hiplib, oracle, Fd, p_drop):
    """the same check over the row widths that pick the kernel: k_spmm_fwd_rows_w<.., M> with M = 2 (256: the second
    piece slot empty; 500; 512: full), 3 (768: full), 4 (772, 1024: full) and the generic kernel (1028 > 1024 floats)"""
    _fused_gather_aggregate_case(dev, hiplib, oracle, 0.3, p_drop, "mean", Fd)


def _fused_gather_aggregate_case(dev, hiplib, oracle, ratio, p_drop, reduce, Fd):
    from pagraph_amd import _lib as L
    rng = np.random.default_rng(int(ratio * 10) + int(p_drop * 100) + Fd)
    V, N, n_src, n_dst = 4000, 6000, 3000, 1100
    table = rng.random((N, Fd), dtype=np.float32)
    nid_map = np.sort(rng.choice(N, V, replace=False)).astype(np.int64)
    st = oracle.CacheState(V, nid_map)
    cached = rng.permutation(V)[:int(V * ratio)].astype(np.int64)
    st.cache_fix_data(cached, {"f": table}, ratio == 1.0)
    ids = rng.choice(V, n_src, replace=False).astype(np.int64)
    ids[-7:] = -1                                             # padding of a fixed-shape NodeFlow
    deg = rng.integers(0, 6, n_dst); deg[5] = 0; deg[17] = 150     # an empty destination, one with > 64 edges
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    src = rng.integers(0, n_src - 7, int(indptr[-1])).astype(np.int32)
    # oracle: materialise, drop, aggregate
    rows = np.zeros((n_src, Fd), np.float32)
    rows[:-7] = st.fetch_layer(ids[:-7], {"f": table})["f"]
    thr = oracle.dropout_threshold(p_drop)
    seed, tag, step = 0x1234567, 3, 9
    h = rows
    if thr:
        keep, scale = oracle.dropout_mask(n_src, Fd, thr, seed, tag, step)
        h = np.where(keep, rows * scale, np.float32(0)).astype(np.float32)
    want = oracle.spmm_fwd(indptr, src, h, n_dst, reduce)
    # device
    sp = L.stream_ptr()
    d_ids = torch.from_numpy(ids).to(dev)
    slot_map = torch.empty(V, dtype=torch.int32, device=dev)
    L.check(hiplib.pg_slot_map_reset(L.ptr(slot_map), V, sp))
    d_cached, d_nid_map = torch.from_numpy(cached).to(dev), torch.from_numpy(nid_map).to(dev)
    d_indptr, d_src = torch.from_numpy(indptr).to(dev), torch.from_numpy(src).to(dev)
    if len(cached):
        L.check(hiplib.pg_slot_map_assign(L.ptr(slot_map), L.ptr(d_cached), len(cached), sp))
    cache = torch.from_numpy(st.cache["f"]).to(dev) if len(cached) else None
    slots = torch.empty(n_src, dtype=torch.int32, device=dev)
    mpos = torch.empty(n_src, dtype=torch.int32, device=dev)
    mfull = torch.empty(n_src, dtype=torch.int64, device=dev)
    mcnt = torch.zeros(1, dtype=torch.int32, device=dev)
    stats = torch.zeros(2, dtype=torch.int64, device=dev)
    ml = L.miss_list(mpos, mfull, mcnt)
    L.check(hiplib.pg_split_rows(L.ptr(d_ids), n_src, L.ptr(slot_map), L.ptr(d_nid_map), ctypes.byref(ml), L.ptr(slots),
                                 L.ptr(stats), None, sp))
    m = int(mcnt.item())
    assert stats.tolist() == [n_src - 7, m] and m == int((~st.gpu_flag[ids[:-7]].astype(bool)).sum())
    sl = slots.cpu().numpy()
    assert np.array_equal(np.sort(-sl[sl <= -3] - 3), np.arange(m)) and np.all(sl[-7:] == -2)
    staged = torch.from_numpy(table[mfull[:m].cpu().numpy()]).to(dev) if m else None       # the miss path's copy
    [L.note(t_) for t_ in (slots, cache, staged if m else None)]       # (debug build: the buffers' extents bound the indices)
    rs = L.PgRowSource(slots.data_ptr(), cache.data_ptr() if cache is not None else 0, staged.data_ptr() if m else 0,
                       cache.stride(0) if cache is not None else Fd, Fd)
    out = torch.empty((n_dst, Fd), dtype=torch.float32, device=dev)
    stepd = torch.tensor([step], dtype=torch.int64, device=dev)
    drop = L.PgDropout(thr, tag, seed, L.ptr(stepd))
    prof = torch.zeros(L.PG_PROF_WORDS * 16, dtype=torch.int64, device=dev)
    L.check(hiplib.pg_spmm_fwd_rows(L.ptr(d_indptr), L.ptr(d_src),
                                    ctypes.byref(rs), n_dst, Fd, {"mean": 0, "sum": 1, "max": 2}[reduce], L.ptr(out), Fd,
                                    ctypes.byref(drop), L.ptr(prof), 16, sp))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    # the kernel stamped its own start / end (100 MHz ticks; every block stamps a shard of the end) and cleared the next entry
    e = prof.view(16, L.PG_PROF_WORDS)[step % 16].tolist()
    ends = e[L.PG_PROF_END0::L.PG_PROF_SHARD_STRIDE]
    t0, ne, t1 = e[0], e[2], max(ends)
    assert 0 < t1 - t0 < 100_000_000 and ne == int(indptr[-1])
    assert len(ends) == L.PG_PROF_SHARDS and (min(ends) > 0 or n_dst < 4 * L.PG_PROF_SHARDS)
    assert prof.view(16, L.PG_PROF_WORDS)[(step + 1) % 16].abs().sum().item() == 0
    # ... and armed the successor stamp: the next dense launch of this thread writes word [1] of the same entry
    if Fd % 4 == 0:
        w8, y8 = torch.rand((8, Fd), device=dev), torch.empty((64, 8), device=dev)
        from pagraph_amd import ops as _ops
        L.check(_ops.linear_fwd_call(hiplib, out, w8, None, y8, 64, 8, 0, stream=sp))
        torch.cuda.synchronize()
        t_succ = prof.view(16, L.PG_PROF_WORDS)[step % 16, 1].item()
        assert t_succ >= t1 and t_succ - t0 < 100_000_000
        assert torch.allclose(y8, out[:64] @ w8.t(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("Fd,reduce,p_drop,max_deg", [(600, "mean", 0.2, 2), (602, "mean", 0.3, 2), (600, "sum", 0.0, 7),
                                                     (1000, "mean", 0.2, 150)])
def test_fused_gather_aggregate_with_composed_edge_slots(dev, hiplib, Fd, reduce, p_drop, max_deg):
    """pg_row_source_t.edge_slots (slots[src[e]] per edge, composed beforehand by pg_compose_edge_slots off the consumer's
    stream: one dependent index load less in pg_spmm_fwd_rows) gives the same rows, bit for bit, as the look-up inside the
    kernel — hits, staged misses, padding sources, empty destinations, a hub destination."""
    from pagraph_amd import _lib as L
    rng = np.random.default_rng(Fd + max_deg)
    n_cache, n_src, n_dst = 3000, 2600, 1191
    cs = (Fd + 7) & ~7
    fused = torch.from_numpy(rng.random((n_cache, cs), dtype=np.float32)).to(dev)         # fused cache rows [F | norm | pad]
    n_miss = 500
    staged_stride = (Fd + 3) & ~3
    staged = torch.from_numpy(rng.random((n_miss, staged_stride), dtype=np.float32)).to(dev)
    slots = rng.integers(0, n_cache, n_src).astype(np.int32)
    miss = rng.permutation(n_src)[:n_miss]
    slots[miss] = -(np.arange(n_miss, dtype=np.int32) + 3)
    slots[-9:] = -2                                                                       # padding of a fixed-shape layer
    slots[-12:-9] = -1
    deg = rng.integers(0, min(max_deg, 3) + 1, n_dst)
    deg[5] = 0
    deg[17] = max_deg
    deg[n_dst - 40:] = 0                                                                  # the padded tail of the block
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    src = rng.integers(0, n_src, int(indptr[-1])).astype(np.int32)
    d_indptr, d_src, d_slots = (torch.from_numpy(a).to(dev) for a in (indptr, src, slots))
    sp = L.stream_ptr()
    es = torch.empty(len(src), dtype=torch.int32, device=dev)
    L.check(hiplib.pg_compose_edge_slots(L.ptr(d_src), len(src), L.ptr(d_slots), n_src, L.ptr(es), sp))
    assert np.array_equal(es.cpu().numpy(), slots[src])
    [L.note(t_) for t_ in (d_slots, fused, staged)]
    stepd = torch.tensor([11], dtype=torch.int64, device=dev)
    drop = L.PgDropout(min(65535, int(round(p_drop * 65536))), 4, 0xABCDEF12345, L.ptr(stepd))
    red = {"mean": 0, "sum": 1}[reduce]
    pad = (Fd + 7) & ~7
    outs = []
    for use_es in (False, True):
        rs = L.PgRowSource(d_slots.data_ptr(), fused.data_ptr(), staged.data_ptr(), cs, staged_stride, es.data_ptr() if use_es else 0)
        out = torch.zeros((n_dst, pad), device=dev)
        L.check(hiplib.pg_spmm_fwd_rows(L.ptr(d_indptr), L.ptr(d_src), ctypes.byref(rs), n_dst, Fd, red, L.ptr(out), pad,
                                        ctypes.byref(drop), None, 0, sp))
        outs.append(out)
    torch.cuda.synchronize()
    k4 = (Fd + 3) & ~3
    assert torch.equal(outs[0][:, :k4], outs[1][:, :k4]) and float(outs[0].abs().sum()) > 0


@pytest.mark.parametrize("arch", ["gcn", "sage"])
@pytest.mark.parametrize("mode", ["async", "full", "async-split", "async-device-only"])
def test_virtual_layer0_matches_materialised(dev, hiplib, arch, mode):
    """fetch_data(virtual=model.virtual_inputs()): logits and gradients equal the materialised path bit for bit,
    with dropout on (same Philox counters), for a partial cache over the async miss queue and for a full cache"""
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling, GraphSageSampling
    from pagraph_amd.ops import RowSource
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    rng = np.random.default_rng(12)
    # (>= 1024 rows in every layer: below that ops.linear / linear2 hand a DENSE operand to the library GEMM, whose rounding
    # differs from the MFMA kernel an un-materialised operand always takes)
    V, Fd, C, B, k = 5000, 600, 11, 1250, 2
    adj = _rand_csc(rng, V, 40000)
    g = DeviceGraph(adj)
    feats = rng.random((V, Fd), dtype=np.float32)
    store = HostFeatureStore({"features": torch.from_numpy(feats)})
    c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode="async")
    c.init_field(["features"])
    # async-split / async-device-only: the worker moves the head of every miss list, the device reads the tail from the
    # host table straight into the staged block (pg_missq_device_tail) — what adapt_cpu_share picks on a starved host
    c.cpu_share = {"async-split": 0.4, "async-device-only": 0.0}.get(mode, 1.0)
    c.auto_cache(g, ["features"], cache_ratio=1.0 if mode == "full" else 0.3)
    assert c.full_cached == (mode == "full")
    torch.manual_seed(5)
    model = (GCNSampling(Fd, 32, C, 1, Fn.relu, 0.3) if arch == "gcn" else GraphSageSampling(Fd, 16, C, 1, Fn.relu, 0.3, 'mean'))
    model = model.to(dev).train()
    need = model.required_inputs(3)
    virt = model.virtual_inputs(3)
    # GCN reads layer 0 only; GraphSAGE's self terms read layers 1 and 2 in place as well (pg_linear_fwd (rows in place), round 3)
    assert virt == ({0: ['features']} if arch == "gcn" else {0: ['features'], 1: ['features'], 2: ['features']})
    smp = NeighborSampler(g, B, k, neighbor_type='in', shuffle=False, num_hops=2, seed_nodes=np.arange(V), seed=1)
    it = iter(smp)
    for rep in range(2):
        nf = next(it)
        assert min(nf.layer_size(i) for i in range(3)) >= 1024
        outs = []
        for v in (None, virt):
            model._drop_step.fill_(7 + rep)                  # both runs draw the same dropout masks
            model.zero_grad(set_to_none=True)
            c.fetch_data(nf, need=need, slot=rep, virtual=v)
            c.wait_misses(rep)
            assert isinstance(nf._node_frames[0]["features"], RowSource) == (v is not None)
            if arch == "sage":
                assert all(isinstance(nf._node_frames[i]["features"], RowSource) == (v is not None) for i in (1, 2))
            y = model(nf)
            y.square().sum().backward()
            torch.cuda.synchronize()
            outs.append((y.detach().clone(), [p.grad.clone() for p in model.parameters()]))
        assert torch.equal(outs[0][0], outs[1][0])
        for a, b in zip(outs[0][1], outs[1][1]):
            assert torch.equal(a, b)
    if mode == "async":
        # adapt_cpu_share: derived from the worker's own counters; a pretended slow gather shifts rows to the device
        assert c.adapt_cpu_share(min_jobs=10 ** 6) is None            # not enough jobs to go by
        rec = c.adapt_cpu_share(min_jobs=1, quiet=True)
        assert rec is not None and 0.0 <= rec["cpu_share"] <= 1.0 and rec["us_per_row_cpu_gather"] > 0
        c._adapt_prev = None
        rec = c.adapt_cpu_share(min_jobs=1, floor_GBps=1e6, quiet=True)   # an "infinitely fast" PCIe: the CPU can never keep up
        assert rec["cpu_share"] == 0.0 and c.cpu_share == 0.0
        nf = next(it)
        c.fetch_data(nf, need=need, slot=0, virtual=virt)
        c.wait_misses(0)
        torch.cuda.synchronize()
        ids0 = nf.layer_parent_nid(0).cpu().numpy()
        out = ops_aggregate_identity(nf._node_frames[0]["features"], dev)
        assert np.array_equal(out, feats[ids0])


@pytest.mark.parametrize("n,K,N,K2,act", [(12000, 600, 16, 600, 2), (6000, 600, 16, 600, 1), (4133, 600, 32, 64, 2),
                                          (1000, 602, 16, 602, 0), (777, 256, 64, 0, 1), (50, 600, 16, 600, 2)])
def test_dense_step_from_row_source_is_bit_identical(dev, hiplib, n, K, N, K2, act):
    """pg_linear_fwd (rows in place) / pg_linear_bwd_w (rows in place) read the first operand's rows where they live (cache slot, staged miss
    row, zero for padding) and give exactly what pg_linear_fwd (two operands) / pg_linear_bwd_w give on the gathered copy"""
    from pagraph_amd import _lib as L
    rng = np.random.default_rng(n + K)
    cs = (K + 7) & ~7                      # fused cache rows are padded
    ss = (K + 3) & ~3                      # staged rows: whole 16-byte pieces
    n_cache, n_staged = 3000, 700
    cache = torch.from_numpy(rng.standard_normal((n_cache, cs), dtype=np.float32)).to(dev)
    staged = torch.from_numpy(rng.standard_normal((n_staged, ss), dtype=np.float32)).to(dev)
    kind = rng.random(n)
    slots = np.where(kind < 0.7, rng.integers(0, n_cache, n), np.where(kind < 0.95, -(rng.integers(0, n_staged, n) + 3), -2))
    slots = slots.astype(np.int32)
    X = np.zeros((n, cs), np.float32)      # what a gather would have produced (padding rows: zeros)
    hit, miss = slots >= 0, slots <= -3
    X[hit, :K] = cache.cpu().numpy()[slots[hit], :K]
    X[miss, :K] = staged.cpu().numpy()[-(slots[miss] + 3), :K]
    Xd = torch.from_numpy(X).to(dev)
    sl = torch.from_numpy(slots).to(dev)
    W = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32) * 0.05).to(dev)
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev)
    X2 = W2 = b2 = None
    if K2:
        X2 = torch.from_numpy(rng.standard_normal((n, (K2 + 3) & ~3), dtype=np.float32)).to(dev)
        W2 = torch.from_numpy(rng.standard_normal((N, K2), dtype=np.float32) * 0.05).to(dev)
        b2 = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev)
    [L.note(t_) for t_ in (sl, cache, staged)]
    rs = L.PgRowSource(sl.data_ptr(), cache.data_ptr(), staged.data_ptr(), cs, ss)
    yc = 2 * N if act == 2 else N
    Ya = torch.empty((n, yc), device=dev); Yb = torch.empty((n, yc), device=dev)
    # (through the descriptor of the C-ABI entry point: ops.linear_fwd_call / linear_bwd_call only fill pg_linear_*_desc_t)
    from pagraph_amd import ops
    from pagraph_amd.ops import RowSource
    rows_src = RowSource(sl, cache, staged.data_ptr(), ss, K, keep=(staged,))
    Xv = Xd[:, :K] if Xd.size(1) != K else Xd            # [n, K] view of the gathered copy (row stride cs)
    X2v = X2[:, :K2] if K2 else None
    L.check(ops.linear_fwd_call(hiplib, Xv, W, b, Ya, n, N, act, x2=X2v, w2=W2, b2=b2, stream=ctypes.c_void_p(0)), "dense fwd")
    L.check(ops.linear_fwd_call(hiplib, rows_src, W, b, Yb, n, N, act, x2=X2v, w2=W2, b2=b2, stream=ctypes.c_void_p(0)),
            "pg_linear_fwd (rows in place)")
    torch.cuda.synchronize()
    assert torch.equal(Ya, Yb)
    # weight gradient of the first operand
    G = torch.from_numpy(rng.standard_normal((n, yc), dtype=np.float32)).to(dev)
    scratch = hiplib.pg_linear_bwd_w_scratch(n, K, N)
    res = []
    for rows in (False, True):
        part = torch.zeros(scratch, device=dev)
        dW = torch.empty((N, K), device=dev); db = torch.empty(N, device=dev)
        dz = torch.empty((n, N), device=dev)
        L.check(ops.linear_bwd_call(hiplib, G, rows_src if rows else Xv, K, N, dW, db, part, 1, y=Ya, act=act, dz=dz,
                                    stream=ctypes.c_void_p(0)), "pg_linear_bwd_w")
        torch.cuda.synchronize()
        res.append((dW.clone(), db.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    ref = (G[:, :N] if act == 0 else None)
    if act == 0:      # sanity against torch in float64
        want = ref.double().t() @ Xd[:, :K].double()
        assert (res[1][0].double() - want).abs().max() < 1e-3 * max(1.0, float(want.abs().max()))
    if K2:
        # pg_linear_bwd_w (two operands): BOTH operands' weight gradients in one launch == one pg_linear_bwd_w / _rows call per operand,
        # partial rows and sums bit for bit (summed and left for the optimiser), first operand dense and read in place
        sc2 = hiplib.pg_linear_bwd_w_scratch(n, K2, N)
        p2 = torch.zeros(sc2, device=dev); dW2 = torch.empty((N, K2), device=dev); db2 = torch.empty(N, device=dev)
        dzr = torch.empty((n, N), device=dev)
        p1 = torch.zeros(scratch, device=dev); dW1 = torch.empty((N, K), device=dev); db1 = torch.empty(N, device=dev)
        L.check(ops.linear_bwd_call(hiplib, G, Xv, K, N, dW1, db1, p1, 1, y=Ya, act=act, dz=dzr, stream=ctypes.c_void_p(0)))
        dzin = dzr if act else G
        L.check(ops.linear_bwd_call(hiplib, dzin, X2v, K2, N, dW2, db2, p2, 1, stream=ctypes.c_void_p(0)))
        for rows in (False, True):
            for summed in (1, 0):
                q1 = torch.zeros(scratch, device=dev); q2 = torch.zeros(sc2, device=dev)
                eW1 = torch.full((N, K), 7.0, device=dev); eb1 = torch.full((N,), 7.0, device=dev)
                eW2 = torch.full((N, K2), 7.0, device=dev); eb2 = torch.full((N,), 7.0, device=dev)
                dz2 = torch.empty((n, N), device=dev)
                L.check(ops.linear_bwd_call(hiplib, G, rows_src if rows else Xv, K, N, eW1, eb1, q1, summed, y=Ya, act=act, dz=dz2,
                                            x2=X2v, K2=K2, dW2=eW2, db2=eb2, part2=q2, stream=ctypes.c_void_p(0)),
                        "pg_linear_bwd_w (two operands)")
                torch.cuda.synchronize()
                assert torch.equal(q1, p1) and torch.equal(q2, p2), (rows, summed)
                if summed:
                    assert torch.equal(eW1, dW1) and torch.equal(eb1, db1) and torch.equal(eW2, dW2) and torch.equal(eb2, db2)
                else:
                    assert float(eW1.min()) == 7.0 and float(eb2.min()) == 7.0          # left to the optimiser's launch
                if act:
                    assert torch.equal(dz2, dzr)
        bad = L.PgLinearBwdDesc()
        bad.dY, bad.X1, bad.X1rows = L.ptr(G).value, L.ptr(Xd).value, ctypes.addressof(rs)       # both forms of the first operand
        bad.n, bad.K1, bad.N, bad.dy_stride, bad.x1_stride = n, K, N, yc, cs
        assert hiplib.pg_linear_bwd_w(ctypes.byref(bad), None) == -1


def ops_aggregate_identity(rows, dev):
    """materialise a RowSource through the fused kernel: identity block, sum, no dropout"""
    from pagraph_amd import ops
    n = rows.shape[0]
    ip = torch.arange(n + 1, dtype=torch.int32, device=dev)
    sr = torch.arange(n, dtype=torch.int32, device=dev)
    return ops.aggregate_rows(ip, sr, rows, n, "sum").cpu().numpy()


# ---- config 3 at full size: RMAT 10 M vertices / 100 M undirected edges, feat 600, GraphSAGE, 30 % cache -----------
@pytest.mark.timeout(1200)
def test_config3_full_size_sampler_and_fetch(dev, hiplib, oracle):
    """BASELINE.json configs[2] at its real size (V = 10^7, nnz = 2 x 10^8, 24 GB host table): the sampler's
    structural invariants on the GPU, node ids / blocks bit-exact against the C oracle for two minibatches, and
    fetch_data (GraphSAGE `need`, async miss queue, materialised and with layer 0 left in place) equal to
    table[nid_map[ids]] bit for bit."""
    import torch.nn.functional as Fn
    from pagraph_amd.data import synthetic as syn
    from pagraph_amd.model import GraphSageSampling
    from pagraph_amd.ops import RowSource
    from pagraph_amd.partition.utils import closure_device
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    free_host = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    if free_host < 40 << 30:
        pytest.skip("needs ~30 GB of free host memory for the 10M x 600 feature table")
    V, E, Fd, B, k, hops = 10_000_000, 100_000_000, 600, 6000, 2, 2
    indptr, indices = syn.rmat_graph(V, E, device=dev)
    assert indices.numel() == 2 * E and int(indptr[-1]) == 2 * E
    train_mask, _, _ = syn.split_dataset(V)
    train = torch.nonzero(train_mask).squeeze(1)
    g_full = DeviceGraph.from_csc(indptr, indices, V)
    sub_indptr, sub_indices, sub2full, subtrain = closure_device(g_full, train, hops)
    del g_full, indptr, indices
    torch.cuda.empty_cache()
    Vs = sub2full.numel()
    # utils.py:48-50: the clamp quirk may fold one train vertex away
    assert Vs > 8_000_000 and subtrain.numel() in (train.numel() - 1, train.numel())
    g = DeviceGraph.from_csc(sub_indptr, sub_indices, Vs)
    table = torch.empty((V, Fd), dtype=torch.float32, pin_memory=True)
    syn.fill_random_features(table, device=dev)
    cacher = GraphCacheServer(HostFeatureStore({"features": table}, pin=False, device_visible={"features": True}), Vs,
                              sub2full, 0, miss_mode="async")
    cacher.init_field(["features"])
    cacher.log = True
    torch.cuda.reset_peak_memory_stats(dev)
    cacher.auto_cache(g, ["features"], cache_ratio=0.30)
    assert cacher.cached_num == int(Vs * 0.30) and not cacher.full_cached
    sampler = NeighborSampler(g, B, k, neighbor_type='in', shuffle=True, num_hops=hops, seed_nodes=subtrain, seed=0)
    seeds_h = sampler.seeds.cpu().numpy()
    indptr_h, indices_h = g.indptr.cpu().numpy(), g.indices.cpu().numpy()
    indeg = g.indptr[1:] - g.indptr[:-1]
    # every CSC entry as one sorted key (column-major, rows ascending inside a column)
    col = torch.repeat_interleave(torch.arange(Vs, device=dev), indeg)
    all_keys = col * Vs + g.indices.long()
    del col
    model = GraphSageSampling(Fd, 16, 60, 1, Fn.relu, 0.2, 'mean').to(dev)
    need, virt = model.required_inputs(hops + 1), model.virtual_inputs(hops + 1)
    sub2full_h = sub2full.cpu()
    it = iter(sampler)
    for b in range(3):
        nf = next(it)
        nm = nf._node_mapping.tousertensor()
        o = nf._layer_offsets
        assert o[-1] - o[-2] == B and torch.equal(nm[o[-2]:o[-1]], sampler.seeds[b * B:(b + 1) * B])
        for l in range(hops):                              # non-seed layers: ascending and unique
            lay = nm[o[l]:o[l + 1]]
            assert bool((lay[1:] > lay[:-1]).all()) and int(lay[0]) >= 0 and int(lay[-1]) < Vs
        for blk in range(hops):
            ip, sr = nf.blk_indptr[blk].long(), nf.blk_src[blk].long()
            dst_ids = nm[o[blk + 1]:o[blk + 2]]
            cnt = ip[1:] - ip[:-1]
            assert torch.equal(cnt, torch.minimum(indeg[dst_ids], torch.full_like(cnt, k)))     # min(k, deg) picks
            u = nm[o[blk]:o[blk + 1]][sr]
            v = torch.repeat_interleave(dst_ids, cnt)
            key = v * Vs + u
            pos = torch.searchsorted(all_keys, key)
            assert bool((all_keys[pos.clamp(max=all_keys.numel() - 1)] == key).all())            # sampled edge exists
            # distinct picks per destination POSITION (the seed layer repeats vertex 0: utils.py:34 maps every
            # isolated train vertex there)
            pkey = torch.repeat_interleave(torch.arange(cnt.numel(), device=dev), cnt) * Vs + u
            assert torch.unique(pkey).numel() == pkey.numel()
        if b < 2:                                          # node ids and blocks bit-exact against the C oracle
            ref = oracle.sample_nodeflow(indptr_h, indices_h, seeds_h[b * B:(b + 1) * B], k, hops, 0, 0, b)
            assert np.array_equal(nm.cpu().numpy(), ref["node_mapping"])
            assert list(o) == list(ref["layer_offsets"][:hops + 2])
            for blk in range(hops):
                assert np.array_equal(nf.blk_indptr[blk].cpu().numpy(), ref["blocks"][blk][0])
                assert np.array_equal(nf.blk_src[blk].cpu().numpy(), ref["blocks"][blk][1])
        want = table[sub2full_h[nm.cpu()]]                 # the reference's miss-path op on every row (storage.py:128)
        for v in (None, virt):
            cacher.fetch_data(nf, need=need, slot=b, virtual=v)
            cacher.wait_misses(b)
            torch.cuda.synchronize()
            for l in range(hops + 1):
                fr = nf._node_frames[l]["features"]
                if isinstance(fr, RowSource):
                    assert v is not None          # (GraphSAGE: every layer's rows stay where they live, round 3)
                    # read the un-materialised rows the way the kernel does: an identity block, reduce = sum
                    n0 = o[l + 1] - o[l]
                    from pagraph_amd import ops
                    ident_ip = torch.arange(n0 + 1, dtype=torch.int32, device=dev)
                    ident_src = torch.arange(n0, dtype=torch.int32, device=dev)
                    fr = ops.aggregate_rows(ident_ip, ident_src, fr, n0, "sum")
                assert torch.equal(fr.cpu(), want[o[l]:o[l + 1]]), (b, l)
    miss_rate = cacher.get_miss_rate()
    assert 0.15 < miss_rate < 0.35                          # BASELINE: ~24 % of all rows miss with the 30 % degree cache
    cacher.check_misses()


def _rank_share_checks(dev, oracle, g_full, V, table, belongs, P, B, k, hops, ratio, n_batches=2):
    """One rank's share of a dg-partitioned run (pa_gcn.py:35-49: rank r loads partition r): the partition with the largest
    closure, its cache by LOCAL out-degree (storage.py:100), the sampler's NodeFlows bit-exact against the C oracle and
    fetch_data (GCN `need`, async miss queue, layer 0 read in place) equal to table[nid_map[ids]] for `n_batches` minibatches."""
    import torch.nn.functional as Fn
    from pagraph_amd import ops
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.ops import RowSource
    from pagraph_amd.partition.utils import closure_device
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    belongs_d = torch.from_numpy(belongs).to(dev)
    sizes = []
    for r in range(P):
        tr = torch.nonzero(belongs_d == r).squeeze(1).cpu()
        sizes.append(int(closure_device(g_full, tr, hops)[2].numel()))
    r = int(np.argmax(sizes))
    my_train = torch.nonzero(belongs_d == r).squeeze(1).cpu()
    sub_indptr, sub_indices, sub2full, subtrain = closure_device(g_full, my_train, hops)
    Vs = int(sub2full.numel())
    assert Vs == sizes[r] and my_train.numel() - 1 <= subtrain.numel() <= my_train.numel()
    g = DeviceGraph.from_csc(sub_indptr, sub_indices, Vs)
    Fd = table.size(1)
    cacher = GraphCacheServer(HostFeatureStore({"features": table}, pin=False, device_visible={"features": True}), Vs,
                              sub2full, 0, miss_mode="async")
    cacher.init_field(["features"])
    cacher.log = True
    # the reference's cache-size rule subtracts the PEAK allocation (storage.py:70-84): building the synthetic graph and the
    # partition on this GPU is not part of the trainer process that rule was written for
    del g_full
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    cacher.auto_cache(g, ["features"], cache_ratio=ratio)
    n_cached = int(Vs * ratio)
    assert cacher.cached_num == n_cached and not cacher.full_cached
    # the cached set is the top of the LOCAL out-degrees (ties at the cut: lower id first)
    deg = g.out_degrees()
    cached = cacher.slot_map >= 0
    assert int(cached.sum()) == n_cached and int(deg[cached].min()) >= int(deg[~cached].max())
    sampler = NeighborSampler(g, B, k, neighbor_type='in', shuffle=True, num_hops=hops, seed_nodes=subtrain, seed=r)
    seeds_h = sampler.seeds.cpu().numpy()
    indptr_h, indices_h = g.indptr.cpu().numpy(), g.indices.cpu().numpy()
    model = GCNSampling(Fd, 32, 60, 1, Fn.relu, 0.2).to(dev)
    need, virt = model.required_inputs(hops + 1), model.virtual_inputs(hops + 1)
    sub2full_h = sub2full.cpu()
    it = iter(sampler)
    for b in range(n_batches):
        nf = next(it)
        nm = nf._node_mapping.tousertensor()
        o = nf._layer_offsets
        ref = oracle.sample_nodeflow(indptr_h, indices_h, seeds_h[b * B:(b + 1) * B], k, hops, r, 0, b)   # (sampler seed = rank)
        assert np.array_equal(nm.cpu().numpy(), ref["node_mapping"])
        assert list(o) == list(ref["layer_offsets"][:hops + 2])
        for blk in range(hops):
            assert np.array_equal(nf.blk_indptr[blk].cpu().numpy(), ref["blocks"][blk][0])
            assert np.array_equal(nf.blk_src[blk].cpu().numpy(), ref["blocks"][blk][1])
        want = table[sub2full_h[nm[o[0]:o[1]].cpu()]]        # the reference's miss-path op on the rows GCN reads (storage.py:128)
        for v in (None, virt):
            cacher.fetch_data(nf, need=need, slot=b, virtual=v)
            cacher.wait_misses(b)
            torch.cuda.synchronize()
            fr = nf._node_frames[0]["features"]
            if isinstance(fr, RowSource):
                n0 = o[1] - o[0]
                ident_ip = torch.arange(n0 + 1, dtype=torch.int32, device=dev)
                ident_src = torch.arange(n0, dtype=torch.int32, device=dev)
                fr = ops.aggregate_rows(ident_ip, ident_src, fr, n0, "sum")
            assert torch.equal(fr.cpu(), want), (b, v is not None)
    miss_rate = cacher.get_miss_rate()
    cacher.check_misses()
    cacher.shutdown_miss_queue()
    return {"rank": r, "closures": sizes, "partition_vertices": Vs, "miss_rate": miss_rate}


def test_config4_rank_share_full_size(dev, hiplib, oracle):
    """BASELINE.json configs[3] — RMAT 10M / 100M, dg x 4 with --num-hops 2 (README.md:117) — as far as one GPU can host it:
    the partition (device-assisted dg, checked at this size against the host code by tools/exp_dg_gpu.py, profiles/r06) and
    ONE rank's share of it: closure, local-degree cache, sampler and fetch bit-exact (round 6)."""
    import importlib
    from pagraph_amd.data import synthetic as syn
    from pagraph_amd.sampling import DeviceGraph
    dgmod = importlib.import_module("pagraph_amd.partition.dg")
    free_host = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    if free_host < 40 << 30:
        pytest.skip("needs ~30 GB of free host memory for the 10M x 600 feature table")
    V, E, Fd, P = 10_000_000, 100_000_000, 600, 4
    indptr, indices = syn.rmat_graph(V, E, device=dev)
    train_mask, _, _ = syn.split_dataset(V)
    train = torch.nonzero(train_mask).squeeze(1).numpy()
    belongs, _, p_vnum, r_vnum = dgmod.dg_raw(P, indptr, indices, V, train, 2, device="cuda", want_r_mask=False)
    assert dgmod.LAST_GPU_STATS is not None and p_vnum.sum() == len(train)
    assert p_vnum.max() - p_vnum.min() <= 1                           # dg balances the train vertices (dg.py:54-55)
    assert np.array_equal(np.bincount(belongs[belongs >= 0], minlength=P), p_vnum)
    g_full = DeviceGraph.from_csc(indptr, indices, V)
    table = torch.empty((V, Fd), dtype=torch.float32, pin_memory=True)
    syn.fill_random_features(table, device=dev)
    rec = _rank_share_checks(dev, oracle, g_full, V, table, belongs, P, 6000, 2, 2, 0.30)
    # what dg's partition does on this graph: every partition's 2-hop closure is still ~85 % of the vertices
    assert all(8_000_000 < c < 9_000_000 for c in rec["closures"]) and 0.05 < rec["miss_rate"] < 0.35


def test_config5_rank_share_full_size(dev, hiplib, oracle):
    """BASELINE.json configs[4] — RMAT 10^8 / 10^9, dg x 8, features in host DRAM behind the async miss path — on one GPU:
    the device-assisted dg at this size equal to the host code (hops 1: the host code needs ~30 s; hops 2 would need ~1000 s and
    is compared at 10^7 instead), then one rank's share: 64-bit CSC offsets (nnz 1.8e9), sampler and fetch bit-exact. The
    feature rows are 16 floats wide here (a 240 GB table is bench.py's business, not a test's)."""
    import importlib
    from pagraph_amd.data import synthetic as syn
    from pagraph_amd.sampling import DeviceGraph
    dgmod = importlib.import_module("pagraph_amd.partition.dg")
    free_host = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    if free_host < 80 << 30 or torch.cuda.mem_get_info()[0] < 120 << 30:
        pytest.skip("needs ~60 GB of free host memory and ~100 GB of HBM")
    V, E, Fd, P = 100_000_000, 1_000_000_000, 16, 8
    indptr, indices = syn.rmat_graph(V, E, device=dev)
    assert int(indptr[-1]) == 2 * E
    train_mask, _, _ = syn.split_dataset(V)
    train = torch.nonzero(train_mask).squeeze(1).numpy()
    a = dgmod.dg_raw(P, indptr, indices, V, train, 1, device="cuda", want_r_mask=False)
    assert dgmod.LAST_GPU_STATS is not None
    b = dgmod.dg_raw(P, indptr, indices, V, train, 1, device="cpu", want_r_mask=False)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    g_full = DeviceGraph.from_csc(indptr, indices, V)
    table = torch.empty((V, Fd), dtype=torch.float32, pin_memory=True)
    syn.fill_random_features(table, device=dev)
    rec = _rank_share_checks(dev, oracle, g_full, V, table, a[0], P, 6000, 2, 2, 0.30, n_batches=2)
    assert rec["partition_vertices"] > 70_000_000 and 0.02 < rec["miss_rate"] < 0.35


# ---- f-4: cache-policy analysis and evaluation tooling -------------------------------------------------------------
def test_cache_analysis_vs_reference_golden(dev, hiplib, golden_dir):
    """pagraph_amd.analysis (access_frequency / optimal_cache_hit) and examples/count_vnum.count_nf_vnum == the
    reference's opt_cache_hit.py / count_vnum.py functions on the G9 trace (seed layers repeat vertices: counted once
    per layer for the frequency, every row for the vertex count)"""
    import importlib.util
    import types
    from pagraph_amd import analysis
    from pagraph_amd.sampling.nodeflow import NodeFlow
    z = np.load(os.path.join(golden_dir, "g9_cache_analysis.npz"))
    V = int(z["V"])
    nfs = []
    for t in range(int(z["num_nodeflows"])):
        layers = [z[f"nf{t}_layer{i}"] for i in range(3)]
        offs = np.concatenate([[0], np.cumsum([len(l) for l in layers])])
        empty = [torch.zeros(len(layers[i + 1]) + 1, dtype=torch.int32, device=dev) for i in range(2)]
        nfs.append(NodeFlow(torch.from_numpy(np.concatenate(layers)).to(dev), offs, empty,
                            [torch.zeros(0, dtype=torch.int32, device=dev)] * 2))
    fake = types.SimpleNamespace(g=types.SimpleNamespace(number_of_nodes=lambda: V), device=dev, __iter__=None)
    class _S:
        g = fake.g
        device = dev
        def __iter__(self):
            return iter(nfs)
    freq, loaded = analysis.access_frequency(_S())
    assert np.array_equal(freq.cpu().numpy(), z["freq"]) and loaded == int(z["vnum"])
    for r in (0.05, 0.2, 0.5):
        assert abs(analysis.optimal_cache_hit(freq, r) - float(z[f"opt_hit_{int(r * 100):02d}"])) < 1e-12
    spec = importlib.util.spec_from_file_location("count_vnum", os.path.join(ROOT, "examples", "count_vnum.py"))
    cv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cv)
    assert sum(cv.count_nf_vnum(nf) for nf in nfs) == int(z["vnum"])


def test_adam_with_deferred_partial_sums_is_bit_identical(dev, hiplib):
    """ops.defer_partials + Adam.step(deferred=...) (pg_adam_step: the two ordered partial sums of the replayed
    GCN step folded into the optimiser's launch) == the three-launch path, bit for bit, over several steps: parameters,
    Adam state, gradients left in p.grad, and the fused head's loss value"""
    import torch.nn.functional as Fn
    from pagraph_amd import ops
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    rng = np.random.default_rng(21)
    V, Fd, C, B = 6000, 600, 60, 3000
    g = DeviceGraph(_rand_csc(rng, V, 60000))
    feats = torch.from_numpy(rng.random((V, Fd), dtype=np.float32)).to(dev)
    labels_all = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=False, num_hops=2, seed_nodes=np.arange(0, V, 2), seed=2)
    nfs = [nf for _, nf in zip(range(3), smp)]
    runs = []
    for deferred in (False, True):
        torch.manual_seed(3)
        model = GCNSampling(Fd, 32, C, 1, Fn.relu, 0.2).to(dev).train()
        opt = Adam(model.parameters(), lr=3e-2)
        seed = torch.ones((), device=dev)
        losses = []
        for it, nf in enumerate(nfs):
            ids = nf._node_mapping.tousertensor()
            o = nf._layer_offsets
            for i in range(nf.num_layers):
                nf._node_frames[i] = {"features": feats[ids[o[i]:o[i + 1]]]} if i == 0 else {}
            lab = labels_all[ids[o[-2]:o[-1]]].contiguous()
            n_valid = torch.tensor([lab.numel()], dtype=torch.int32, device=dev)
            model._drop_step.fill_(10 + it)
            opt.zero_grad(set_to_none=True)
            if deferred:
                with ops.defer_partials() as reg:
                    loss = model.forward_loss(nf, lab, n_valid, seed)
                    loss.backward(seed)
                assert len(reg.by_param) == 4 and len(reg.extra) == 1 and not reg.conflict
                opt.step(deferred=reg)
            else:
                loss = model.forward_loss(nf, lab, n_valid, seed)
                loss.backward(seed)
                opt.step()
            torch.cuda.synchronize()
            losses.append(loss.detach().clone())
        runs.append((losses, [p.detach().clone() for p in model.parameters()], [p.grad.clone() for p in model.parameters()],
                     [opt.state[p]['exp_avg_sq'].clone() for p in model.parameters()]))
    for a, b in zip(runs[0], runs[1]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert all(torch.isfinite(l) for l in runs[0][0])


def test_zerocopy_refused_on_pageable_table_and_recapture_after_cache_change(dev, hiplib):
    """(1) ADVICE r1: a kernel reading a pageable host table faults — GraphCacheServer falls back from 'zerocopy' to the
    async queue when a table is not page-locked, and refuses cpu_share < 1 there. (2) A GraphedTrainer slot's graph reads
    the cache in place (fused gather): after the cache contents change (second auto_cache) the graph is re-captured and
    the loss trajectory still equals a trainer that never cached anything."""
    import torch.nn.functional as Fn
    from pagraph_amd import _lib as L
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, cycle_batches
    rng = np.random.default_rng(8)
    V, Fd, C, B = 4000, 600, 9, 500
    feats = torch.from_numpy(rng.random((V, Fd), dtype=np.float32))
    store = HostFeatureStore({"features": feats}, pin=False)
    assert store.pinned == {"features": False}
    c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode="zerocopy")
    assert c.miss_mode == "async"
    c.init_field(["features"])
    c.cpu_share = 0.5
    with pytest.raises(L.PgError):
        c._missq_buffers(0, 1000)
    # (2)
    g = DeviceGraph(_rand_csc(rng, V, 30000))
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    losses = []
    for recache in (False, True):
        store2 = HostFeatureStore({"features": feats})
        cc = GraphCacheServer(store2, V, torch.arange(V), 0, miss_mode="async")
        cc.init_field(["features"])
        torch.manual_seed(4)
        model = GCNSampling(Fd, 32, C, 1, Fn.relu, 0.2).to(dev).train()
        smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=np.arange(0, V, 2), seed=6,
                              static=True, defer_transpose=True)
        tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), Adam(model.parameters(), lr=1e-2), cc, smp, labels, dev,
                            need=model.required_inputs(3), keep_losses=True)
        out = []
        tr.on_step = lambda k, l: out.append(l)
        it = cycle_batches(smp, 40)
        tr.run_steps(it, 14)                       # eager warm-up + captures, nothing cached: every row is a miss
        if recache:
            tr.synchronize(); torch.cuda.synchronize()
            cc.auto_cache(g, ["features"], cache_ratio=0.4)      # cache epoch changes: plans and graphs are rebuilt
            ep = cc._cache_epoch
        tr.run_steps(it, 12)
        tr.synchronize(); torch.cuda.synchronize()
        if recache:
            assert all(s.graph is not None and s.graph_plan is s.plan and s.plan.cache_epoch == ep for s in tr.slots.values())
            assert cc.cached_num == int(V * 0.4)
        losses.append(torch.stack([l.detach().float().cpu() for l in out]))
        cc.check_misses()
    assert torch.equal(losses[0], losses[1])       # features are features, wherever they are read from


@pytest.mark.gpu
def test_async_miss_path_survives_hardware_queue_sharing(dev, hiplib):
    """Regression (round 2): once a process owns more high-priority streams than the class has hardware queues, the miss
    queue's copy stream shares a queue with the sampler / load stream of its own pipeline; a barrier packet of theirs
    that waits for "slot free" / "frames consumed" (recorded after the consumer's spin-wait kernel) then held back the
    very copy that kernel was waiting for — 3 s per occurrence (bench.py's reference-equivalent leg ran at 90-1350
    ms/step, the last test of a long session timed out). pg_missq_wait_idle orders the launch thread's waits after the
    worker's enqueue. Here: a crowd of busy high-priority streams, then three pipelines one after the other on the same
    process; none may stall or lose rows."""
    import time
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, cycle_batches
    crowd = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(9)]
    junk = torch.zeros(1024, device=dev)
    for s in crowd:                                  # a stream takes its hardware queue at first use
        with torch.cuda.stream(s):
            junk.add_(1.0)
    torch.cuda.synchronize()
    rng = np.random.default_rng(21)
    V, Fd, C, B = 20000, 600, 7, 1000
    g = DeviceGraph(_rand_csc(rng, V, 160000))
    feats = torch.from_numpy(rng.random((V, Fd), dtype=np.float32))
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    keep = []
    for rep in range(3):
        store = HostFeatureStore({"features": feats})
        c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode="async")
        c.init_field(["features"])
        c.auto_cache(g, ["features"], cache_ratio=0.3)
        torch.manual_seed(rep)
        model = GCNSampling(Fd, 32, C, 1, Fn.relu, 0.2).to(dev).train()
        smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=np.arange(0, V, 2), seed=rep,
                              static=True, defer_transpose=True)
        tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), Adam(model.parameters(), lr=1e-2), c, smp, labels, dev,
                            need=model.required_inputs(3), keep_losses=False)
        it = cycle_batches(smp, 400)
        tr.run_steps(it, 20)                          # eager warm-up + captures
        tr.synchronize(); torch.cuda.synchronize()
        t0 = time.time()
        tr.run_steps(it, 300)
        tr.synchronize(); torch.cuda.synchronize()
        dt = time.time() - t0
        c.check_misses()                              # raises if a device-side wait gave up
        assert dt < 2.5, f"pipeline {rep}: 300 steps took {dt:.2f} s (a 3 s stall per lost copy)"
        assert torch.isfinite(tr.last_loss).item()
        keep.append((tr, smp, c))                     # their streams stay alive (and keep their hardware queues)
        for s in crowd:
            with torch.cuda.stream(s):
                junk.add_(1.0)
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("F", [8, 600, 602])
def test_miss_list_index_dedup_is_invisible_and_saves_pcie_rows(dev, hiplib, F):
    """North star's "index dedup" where it pays: a vertex that MISSES in several layers of one NodeFlow crosses PCIe
    once (pg_dedup_t: binary search of a missed id in the earlier, sorted layers; the repeat is filled on the device from
    the first occurrence's staged row). Frames and the reference's miss counters are identical with and without it; the
    worker moves exactly the first occurrences. Also: padded (fixed-shape) layers, a `need` subset, repeated slot use."""
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    rng = np.random.default_rng(100 + F)
    V = 60000
    feats = rng.random((V, F), dtype=np.float32)
    norm = rng.random((V, 1), dtype=np.float32)
    l0 = np.sort(rng.choice(V, 9000, replace=False))
    l1 = np.sort(np.unique(np.concatenate([rng.choice(l0, 2500, replace=False), rng.choice(V, 2500, replace=False)])))
    l2 = np.concatenate([rng.choice(l0, 700), rng.choice(l1, 700), rng.choice(V, 600), np.full(120, l0[17])])
    rng.shuffle(l2)                                          # the seeds' layer: unsorted, with repeats of its own
    cached = rng.choice(V, int(0.3 * V), replace=False)
    is_cached = np.zeros(V, bool); is_cached[cached] = True
    pad = lambda a, n: np.concatenate([a, np.full(n - len(a), -1, np.int64)])
    cases = {"exact": [l0, l1, l2], "padded": [pad(l0, 10000), pad(l1, 6000), pad(l2, 2200)]}

    def first_occurrences(layers):
        """rows the worker must move: misses of layer r that do not appear in layers 0..r-1"""
        seen, n = set(), 0
        for r, a in enumerate(layers):
            a = a[a >= 0]
            m = a[~is_cached[a]]
            n += int(np.sum([x not in seen for x in m.tolist()])) if r else len(m)
            seen.update(a.tolist())
        return n

    res = {}
    for dedup in (False, True):
        store = HostFeatureStore({"features": torch.from_numpy(feats), "norm": torch.from_numpy(norm)})
        c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode="async")
        c.dedup_misses = dedup
        c.init_field(["features", "norm"])
        c.log = True
        nids = torch.from_numpy(cached).to(dev)
        c.cache_fix_data(nids, c.get_feat_from_server(nids, ["features", "norm"], to_gpu=True), is_full=False)
        for case, layers in cases.items():
            for need in (None, {1: ["features"], 2: ["features"]}):
                for rep in range(3):
                    nf = FakeNF(layers, dev)
                    q0 = c.miss_queue_stats()
                    c.fetch_data(nf, need=need, slot=rep % 2)
                    c.wait_misses(rep % 2)
                    c.drain_misses(); torch.cuda.synchronize()
                    q1 = c.miss_queue_stats()
                    moved = int(round(q1["rows_per_job"] * q1["jobs"] - (q0["rows_per_job"] * q0["jobs"] if q0 else 0)))
                    lay = range(len(layers)) if need is None else sorted(need)
                    for i in lay:
                        valid = layers[i] >= 0
                        for name in (("features", "norm") if need is None else need[i]):
                            got = nf._node_frames[i][name].cpu().numpy()[valid]
                            want = (feats if name == "features" else norm)[layers[i][valid]]
                            assert np.array_equal(got, want), (dedup, case, need, i, name)
                    sub = [layers[i] for i in lay]
                    all_miss = sum(int((~is_cached[a[a >= 0]]).sum()) for a in sub)
                    assert moved == (first_occurrences(sub) if dedup else all_miss), (dedup, case, need, moved)
                    t, m = c._stats.tolist()
                    c._stats.zero_()
                    assert m == all_miss and t == sum(int((a >= 0).sum()) for a in sub)     # the reference's counting
                    res[(dedup, case, need is None)] = moved
        c.check_misses()
    assert res[(True, "exact", True)] < res[(False, "exact", True)]


# ---- ragged feature width (Reddit: feat = 602, BASELINE configs[0..1]) on the fused path -------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("Fd,p_drop,reduce", [(602, 0.0, "mean"), (602, 0.25, "mean"), (601, 0.5, "sum"), (603, 0.25, "mean"),
                                              (258, 0.25, "mean"), (510, 0.25, "sum"), (770, 0.25, "mean"), (1022, 0.5, "mean"),
                                              (1030, 0.25, "mean")])   # M = 2, 2, 4, 4 and the generic kernel
def test_fused_gather_aggregate_ragged_width_vs_oracle(dev, hiplib, oracle, Fd, p_drop, reduce):
    """pg_spmm_fwd_rows with dim % 4 != 0: rows are read as whole 16-byte pieces out of a padded fused cache row whose
    next columns hold ANOTHER field (here: NaN) — masked on the way in; the output's padding columns are zeros; the sum
    and the dropout keep-mask equal the oracle's bit for bit"""
    from pagraph_amd import _lib as L
    rng = np.random.default_rng(Fd)
    V, n_src, n_dst, stride = 3000, 2500, 900, ((Fd + 3) & ~3) + 4     # 602 -> 608: the fused cache's padded row
    table = rng.random((V, Fd), dtype=np.float32)
    ids = rng.permutation(V)[:n_src].astype(np.int64)
    deg = rng.integers(0, 5, n_dst); deg[3] = 0; deg[11] = 90
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    src = rng.integers(0, n_src, int(indptr[-1])).astype(np.int32)
    thr = oracle.dropout_threshold(p_drop)
    seed, tag, step = 0x9876543, 2, 4
    h = table[ids]
    if thr:
        keep, scale = oracle.dropout_mask(n_src, Fd, thr, seed, tag, step)
        h = np.where(keep, h * scale, np.float32(0)).astype(np.float32)
    want = oracle.spmm_fwd(indptr, src, h, n_dst, reduce)
    fused = torch.full((V, stride), float("nan"), device=dev)          # [features | other field / padding = NaN]
    fused[:, :Fd] = torch.from_numpy(table).to(dev)
    slots = torch.from_numpy(ids.astype(np.int32)).to(dev)              # full cache: slot = id
    [L.note(t_) for t_ in (slots, fused)]
    rs = L.PgRowSource(slots.data_ptr(), fused.data_ptr(), 0, stride, Fd)
    out = torch.full((n_dst, stride), 7.0, device=dev)
    stepd = torch.tensor([step], dtype=torch.int64, device=dev)
    drop = L.PgDropout(thr, tag, seed, L.ptr(stepd))
    d_indptr, d_src = torch.from_numpy(indptr).to(dev), torch.from_numpy(src).to(dev)
    L.check(hiplib.pg_spmm_fwd_rows(L.ptr(d_indptr), L.ptr(d_src), ctypes.byref(rs), n_dst, Fd, 0 if reduce == "mean" else 1,
                                    L.ptr(out), stride, ctypes.byref(drop), None, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :Fd], want)
    d4 = (Fd + 3) & ~3
    assert np.all(got[:, Fd:d4] == 0) and np.all(got[:, d4:] == 7.0)
    # rows that are not padded to whole pieces are refused, not mis-read
    bad = L.PgRowSource(slots.data_ptr(), fused.data_ptr(), 0, Fd, Fd)
    assert hiplib.pg_spmm_fwd_rows(L.ptr(d_indptr), L.ptr(d_src), ctypes.byref(bad), n_dst, Fd, 0, L.ptr(out), stride,
                                   None, None, 0, L.stream_ptr()) == -4      # PG_ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("n,K,N", [(6000, 602, 32), (2049, 601, 16), (3000, 70, 33), (4000, 603, 41)])
def test_skinny_linear_ragged_K_on_the_mfma_kernel(dev, hiplib, n, K, N):
    """pg_linear_fwd with K % 8 != 0 (padded rows, NaN in the padding): the MFMA kernel itself runs (no library
    fall-back) and matches float64; W rows are read with float2 / scalar loads when K % 4 != 0"""
    from pagraph_amd import _lib as L
    from pagraph_amd import ops
    torch.manual_seed(K)
    pad = (K + 7) & ~7
    buf = torch.full((n, pad), float("nan"), device=dev)
    x = buf[:, :K]
    x.copy_(torch.rand((n, K), device=dev) - 0.3)
    lin = torch.nn.Linear(K, N).to(dev)
    y = torch.empty((n, 2 * N), device=dev)
    L.check(ops.linear_fwd_call(hiplib, x, lin.weight, lin.bias, y, n, N, ops.ACT_CONCAT), "pg_linear_fwd")
    z = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double())
    ref = torch.cat((z, torch.relu(z)), 1)
    assert float((y.double() - ref).abs().max()) < TOL * max(1.0, float(ref.abs().max()))
    # through the autograd wrapper: forward on the kernel, weight gradient too
    y2 = ops.linear(x, lin, ops.ACT_RELU)
    assert float((y2.double() - torch.relu(z)).abs().max()) < TOL * max(1.0, float(z.abs().max()))
    g = torch.rand_like(y2) - 0.5
    y2.backward(g)
    gz = g.double() * (z > 0)
    gw = gz.t() @ x.double()
    assert float((lin.weight.grad.double() - gw).abs().max()) < TOL * max(1.0, float(gw.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("ratio", [1.0, 0.3])
@pytest.mark.parametrize("arch", ["gcn", "sage"])
def test_reddit_width_runs_on_the_fused_path(dev, hiplib, arch, ratio):
    """feat = 602, whole table cached (BASELINE configs[1]) or 30 % of it (miss rows read in place from the queue's
    staged block, whose rows are padded to whole 16-byte pieces): layer 0 is aggregated straight from the cache
    (ops.RowSource) with the kernel's own dropout mask; with dropout off the logits and gradients match the
    materialised path (k_gather + k_spmm_fwd + library GEMM) to 1e-4"""
    import torch.nn.functional as Fn
    from pagraph_amd.model import GCNSampling, GraphSageSampling
    from pagraph_amd.ops import RowSource
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    rng = np.random.default_rng(602)
    V, Fd, C, B, k = 6000, 602, 41, 1200, 2
    g = DeviceGraph(_rand_csc(rng, V, 50000))
    feats = rng.random((V, Fd), dtype=np.float32)
    norm = rng.random((V, 1), dtype=np.float32)
    store = HostFeatureStore({"features": torch.from_numpy(feats), "norm": torch.from_numpy(norm)})
    c = GraphCacheServer(store, V, torch.arange(V), 0, miss_mode="async")
    c.init_field(["features", "norm"])
    c.auto_cache(g, ["features", "norm"], cache_ratio=ratio)
    assert c.full_cached == (ratio == 1.0)
    torch.manual_seed(3)
    model = (GCNSampling(Fd, 32, C, 1, Fn.relu, 0.0) if arch == "gcn" else GraphSageSampling(Fd, 16, C, 1, Fn.relu, 0.0, 'mean'))
    model = model.to(dev).train()
    need, virt = model.required_inputs(3), model.virtual_inputs(3)
    smp = NeighborSampler(g, B, k, neighbor_type='in', shuffle=False, num_hops=2, seed_nodes=np.arange(0, V, 2), seed=1)
    nf = next(iter(smp))
    outs = []
    for v in (None, virt):
        model.zero_grad(set_to_none=True)
        c.fetch_data(nf, need=need, slot=0, virtual=v)
        c.wait_misses(0)
        assert isinstance(nf._node_frames[0]["features"], RowSource) == (v is not None)
        y = model(nf)
        y.square().sum().backward()
        torch.cuda.synchronize()
        outs.append((y.detach().clone(), [p.grad.clone() for p in model.parameters()]))
    sc = max(1.0, float(outs[0][0].abs().max()))
    assert float((outs[0][0] - outs[1][0]).abs().max()) < TOL * sc
    for a, b in zip(outs[0][1], outs[1][1]):
        assert float((a - b).abs().max()) < TOL * max(1.0, float(a.abs().max()))
    # dropout on: the fused path draws the kernel's mask (no nn.Dropout fall-back) and still trains
    model2 = GCNSampling(Fd, 32, C, 1, Fn.relu, 0.5).to(dev).train()
    c.fetch_data(nf, need=need, slot=0, virtual=model2.virtual_inputs(3))
    c.wait_misses(0)
    y = model2(nf)
    assert torch.isfinite(y).all()
    c.check_misses()


@pytest.mark.gpu
def test_graphsage_deferred_partial_sums_two_uses_bit_identical(dev, hiplib):
    """GraphSAGE's first NodeUpdate runs on both blocks (graphsage_nssc.py:92-131): its parameters get TWO gradient
    contributions per step. ops.defer_partials hands autograd one placeholder and the optimiser's launch forms
    sum(first) + sum(second) (pg_adam_step) — parameters, Adam state and p.grad equal the unfused path
    (k_sum_partials per contribution + AccumulateGrad's add + pg_adam_step) bit for bit over several steps"""
    import torch.nn.functional as Fn
    from pagraph_amd import ops
    from pagraph_amd.model import GraphSageSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    rng = np.random.default_rng(33)
    V, Fd, C, B = 8000, 600, 41, 3000
    g = DeviceGraph(_rand_csc(rng, V, 80000))
    feats = torch.from_numpy(rng.random((V, Fd), dtype=np.float32)).to(dev)
    labels_all = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=False, num_hops=2, seed_nodes=np.arange(V), seed=2)
    nfs = [nf for _, nf in zip(range(3), smp)]           # 3000, 3000, 2000 seeds: every dense step is on the MFMA kernels
    assert len(nfs) == 3
    loss_fcn = ops.fused_loss(torch.nn.CrossEntropyLoss())
    runs = []
    for deferred in (False, True):
        torch.manual_seed(3)
        model = GraphSageSampling(Fd, 16, C, 1, Fn.relu, 0.2, 'mean').to(dev).train()
        assert model.deferrable_parameters
        opt = Adam(model.parameters(), lr=1e-2)
        losses = []
        for it, nf in enumerate(nfs):
            ids = nf._node_mapping.tousertensor()
            o = nf._layer_offsets
            for i in range(nf.num_layers):
                nf._node_frames[i] = {"features": feats[ids[o[i]:o[i + 1]]]}
            lab = labels_all[ids[o[-2]:o[-1]]].contiguous()
            model._drop_step.fill_(10 + it)
            opt.zero_grad(set_to_none=True)
            if deferred:
                with ops.defer_partials() as reg:
                    loss = loss_fcn(model(nf), lab)
                    loss.backward()
                # 8 parameter tensors; layers.0's four are applied to block 0 and block 1
                assert len(reg.by_param) == 8 and len(reg.second) == 4 and not reg.conflict, \
                    (it, len(reg.by_param), len(reg.second), reg.conflict)
                opt.step(deferred=reg)
            else:
                loss = loss_fcn(model(nf), lab)
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            losses.append(loss.detach().clone())
        runs.append((losses, [p.detach().clone() for p in model.parameters()], [p.grad.clone() for p in model.parameters()],
                     [opt.state[p]['exp_avg_sq'].clone() for p in model.parameters()]))
    for a, b in zip(runs[0], runs[1]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    # a third use of a parameter in one step is refused, not mis-summed
    reg = ops.DeferredPartials()
    w = torch.zeros(4, device=dev)
    pt = torch.zeros(8, device=dev)
    assert reg.add(w, pt, 2, 4, 0) and not reg.add(w, pt, 2, 4, 0) and not reg.conflict
    reg.add(w, pt, 2, 4, 0)
    assert reg.conflict


# ---- config 2 at full size: Reddit-shaped graph (V = 232 965, 114.6 M CSC entries), feat 602, 41 classes, full cache --
@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_config2_full_size_reddit_shape(dev, hiplib, oracle):
    """BASELINE.json configs[1] at its real size (the reference's Reddit: 232 965 vertices, mean degree 492, feat 602, 41
    classes; synthetic RMAT edges and random features of that shape, whole table resident in HBM): the sampler's node
    ids / blocks bit-exact against the C oracle, fetch_data (dense and with layer 0 left in the cache) equal to the
    table bit for bit with a 100 % hit rate, GCNSampling's logits on the fused path within 1e-4 of the oracle's
    restatement of the reference model, and a few replayed training steps that reduce the loss."""
    import torch.nn.functional as Fn
    from pagraph_amd.data import synthetic as syn
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.ops import RowSource
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, cycle_batches
    V, E, Fd, C, B, k, hops = 232_965, 57_300_000, 602, 41, 6000, 2, 2
    indptr, indices = syn.rmat_graph(V, E, device=dev)
    assert int(indptr[-1]) == indices.numel() and indices.numel() > 100_000_000
    g = DeviceGraph.from_csc(indptr, indices, V)
    train_mask, _, _ = syn.split_dataset(V)
    train = torch.nonzero(train_mask).squeeze(1)
    table = torch.empty((V, Fd), dtype=torch.float32, pin_memory=True)
    syn.fill_random_features(table, device=dev)
    labels = syn.random_labels(V, C).to(dev)
    cacher = GraphCacheServer(HostFeatureStore({"features": table}, pin=False, device_visible={"features": True}), V,
                              torch.arange(V), 0, miss_mode="async")
    cacher.init_field(["features"])
    cacher.log = True
    cacher.auto_cache(g, ["features"], cache_ratio=1.0)
    assert cacher.full_cached and cacher.cached_num == V
    sampler = NeighborSampler(g, B, k, neighbor_type='in', shuffle=True, num_hops=hops, seed_nodes=train, seed=0)
    seeds_h = sampler.seeds.cpu().numpy()
    indptr_h, indices_h = g.indptr.cpu().numpy(), g.indices.cpu().numpy()
    torch.manual_seed(2)
    model = GCNSampling(Fd, 32, C, 1, Fn.relu, 0.0).to(dev).train()
    need, virt = model.required_inputs(hops + 1), model.virtual_inputs(hops + 1)
    state = {n_: p.detach().cpu().numpy() for n_, p in model.named_parameters()}
    it = iter(sampler)
    for b in range(2):
        nf = next(it)
        nm = nf._node_mapping.tousertensor()
        o = nf._layer_offsets
        ref = oracle.sample_nodeflow(indptr_h, indices_h, seeds_h[b * B:(b + 1) * B], k, hops, 0, 0, b)
        assert np.array_equal(nm.cpu().numpy(), ref["node_mapping"]) and list(o) == list(ref["layer_offsets"][:hops + 2])
        for blk in range(hops):
            assert np.array_equal(nf.blk_indptr[blk].cpu().numpy(), ref["blocks"][blk][0])
            assert np.array_equal(nf.blk_src[blk].cpu().numpy(), ref["blocks"][blk][1])
        # every layer, dense: the reference's fetch_data
        cacher.fetch_data(nf)
        torch.cuda.synchronize()
        ids_h = nm.cpu()
        for l in range(hops + 1):
            assert torch.equal(nf._node_frames[l]["features"].cpu(), table[ids_h[o[l]:o[l + 1]]])
        assert cacher.get_miss_rate() == 0.0
        # what the model reads, layer 0 left in the cache; logits vs the oracle's model
        cacher.fetch_data(nf, need=need, slot=0, virtual=virt)
        assert isinstance(nf._node_frames[0]["features"], RowSource)
        with torch.no_grad():        # (a live autograd graph made on the default stream would tie the parameters' gradient
            y = model(nf)            #  accumulators to that stream: GraphedTrainer captures on its own — see its docstring)
        torch.cuda.synchronize()
        frames = [{"features": table[ids_h[o[0]:o[1]]].numpy()}] + [{} for _ in range(hops)]
        want, _ = oracle.gcn_model_forward(ref["blocks"][:hops], [o[l + 1] - o[l] for l in range(hops + 1)], frames, state, 1)
        assert float(np.abs(y.detach().cpu().numpy() - want).max()) < TOL * max(1.0, float(np.abs(want).max()))
    # the replayed training step at this shape
    smp = NeighborSampler(g, B, k, neighbor_type='in', shuffle=True, num_hops=hops, seed_nodes=train, seed=1, static=True,
                          defer_transpose=True)
    tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), Adam(model.parameters(), lr=3e-2), cacher, smp, labels, dev,
                        need=need, keep_losses=True)
    out = []
    tr.on_step = lambda step, loss: out.append(loss)
    tr.run_steps(cycle_batches(smp, 40), 24)
    tr.synchronize(); torch.cuda.synchronize()
    ls = torch.stack([l.detach().float().cpu() for l in out])
    assert torch.isfinite(ls).all() and float(ls[-4:].mean()) < float(ls[:4].mean())
    assert all(s_.plan.virtual for s_ in tr.slots.values())          # feat 602 ran on the fused path
