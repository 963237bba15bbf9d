#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE's own Python on CPU and dumps
inputs + outputs as small .npz fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY.  This script needs /root/reference, which exists only
in the build container; it never runs on the GPU box and nothing in the product
imports it.  It copies no reference source: it imports the reference modules in
place behind stubs for the two dependencies that are not installable here
(`dgl`, `numba`) and records what the reference computes.

What is pinned by the reference itself:
  G1  GraphCacheServer.fetch_data        (PaGraph/storage/storage.py:157-204)
  G2  GraphCacheServer.cache_fix_data    (storage.py:135-154)  -> state arrays
  G3  GraphCacheServer.fetch_from_cache  (storage.py:207-216)
  G4  dg()                               (PaGraph/partition/dg.py:59-103)
  G5  GraphCacheServer.auto_cache        (storage.py:70-104)   -> cached id set
  G6  get_sub_graph() numpy tail         (PaGraph/partition/utils.py:25-52)
      fed by a stand-in full-neighbour sampler (DGL's own C++ sampler is NOT
      available: that part of G6 is this build's reading of DGL 0.4.1 and is
      "parity unpinned").
  G7  GCNSampling.forward / preprocess_forward, GCNInfer.forward / preprocess_forward
      with NodeUpdate                    (PaGraph/model/gcn_nssc.py:6-164)
  G8  GraphSageSampling.forward ('mean', 'gcn', 'pool'; with and without preprocess)
      with NodeUpdate                    (PaGraph/model/graphsage_nssc.py:6-134)
      G7/G8 run the reference's model classes (layer stack, skip-concat, norm
      scaling, preprocess branches, parameter names, autograd) on a stand-in
      NodeFlow whose block_compute applies a STAND-IN copy_src + mean|sum|max reducer
      (torch index_add / index_reduce over the block's edges, zero rows for
      destinations without in-edges; max differentiates as DGL's `val == accum`
      [recollection]) in place of DGL's fused message-passing kernel, which is not
      available. Outputs: logits and every parameter gradient of
      sum(logits * G) for a stored G.
  G9  count_vertex_freq / optimal_cache_hit (examples/opt_cache_hit.py:22-31) and
      count_nf_vnum (examples/count_vnum.py:16-20) on a fixed trace of NodeFlows

numpy note (G4): dg.py:31 calls np.argsort with the default, UNSTABLE kind. On
CPUs with AVX2/AVX-512 numpy >= 2.0 dispatches it to x86-simd-sort, whose tie
order differs from numpy's scalar path (insertion sort for n <= 16, i.e. stable)
— so the reference's partition on score ties (every first vertex ties) depends
on the machine it runs on.  The fixtures pin the portable scalar behaviour: this
script disables numpy's SIMD dispatch before importing it.

Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import os
os.environ.setdefault("NPY_DISABLE_CPU_FEATURES",
                      "AVX2 AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL")
import contextlib
import importlib
import sys
import types

sys.dont_write_bytecode = True  # never write __pycache__ into /root/reference

import numpy as np
import scipy.sparse as spsp
import torch

REF = os.environ.get("PAGRAPH_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# --------------------------------------------------------------------------
# stubs for the un-installable third-party modules
# --------------------------------------------------------------------------
class _Frame:
    def __init__(self, d=None):
        self.d = dict(d or {})


class _FrameRef:
    def __init__(self, frame):
        self.frame = frame

    def __getitem__(self, k):
        return self.frame.d[k]


def install_stubs():
    if not hasattr(np, "int"):
        np.int = int  # reference uses the alias removed in numpy>=1.24
    dgl = types.ModuleType("dgl")
    dgl.DGLGraph = object
    frame = types.ModuleType("dgl.frame")
    frame.Frame, frame.FrameRef = _Frame, _FrameRef
    utils = types.ModuleType("dgl.utils")
    fn = types.ModuleType("dgl.function")
    # the builtin message / reduce descriptors the models pass to block_compute (gcn_nssc.py:72-73,
    # graphsage_nssc.py:99-110): plain records, interpreted by StandInBlockNodeFlow.block_compute
    fn.copy_src = lambda src, out: types.SimpleNamespace(kind="copy_src", src=src, out=out)
    fn.mean = lambda msg, out: types.SimpleNamespace(kind="mean", msg=msg, out=out)
    fn.sum = lambda msg, out: types.SimpleNamespace(kind="sum", msg=msg, out=out)
    fn.max = lambda msg, out: types.SimpleNamespace(kind="max", msg=msg, out=out)
    contrib = types.ModuleType("dgl.contrib")
    sampling = types.ModuleType("dgl.contrib.sampling")
    contrib.sampling = sampling
    dgl.frame, dgl.utils, dgl.function, dgl.contrib = frame, utils, fn, contrib
    for name, mod in [("dgl", dgl), ("dgl.frame", frame), ("dgl.utils", utils),
                      ("dgl.function", fn), ("dgl.contrib", contrib),
                      ("dgl.contrib.sampling", sampling),
                      ("numba", types.ModuleType("numba"))]:
        sys.modules[name] = mod
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "PaGraph", "partition"))
    return dgl


def cpu_shims():
    """make the reference's .cuda()/torch.cuda.* calls run on CPU"""
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.device = lambda *_a, **_k: contextlib.nullcontext()
    torch.cuda.LongTensor = lambda *a: (torch.LongTensor(*a))
    torch.cuda.FloatTensor = lambda *a: (torch.FloatTensor(*a))


class FakeStore:
    """stands in for dgl.contrib.graph_store's client: _node_frame._frame[name].data"""

    def __init__(self, fields):
        cols = {k: types.SimpleNamespace(data=v) for k, v in fields.items()}
        self._node_frame = types.SimpleNamespace(_frame=cols)


class FakeNodeFlow:
    def __init__(self, layers):
        self.layers_ = [torch.as_tensor(l, dtype=torch.int64) for l in layers]
        cat = torch.cat(self.layers_) if layers else torch.zeros(0, dtype=torch.int64)
        self._node_mapping = types.SimpleNamespace(tousertensor=lambda: cat)
        offs = [0]
        for l in self.layers_:
            offs.append(offs[-1] + int(l.numel()))
        self._layer_offsets = offs
        self.num_layers = len(layers)
        self._node_frames = [None] * len(layers)

    def layer_parent_nid(self, i):
        return self.layers_[i]


# --------------------------------------------------------------------------
# G1/G2/G3/G5: storage
# --------------------------------------------------------------------------
def gen_storage(storage):
    rng = np.random.default_rng(20260929)
    for F in (8, 600, 602):
        N, V_sub = 96, 64                      # full-graph rows / partition rows
        feats = rng.random((N, F), dtype=np.float32)
        norm = (1.0 / rng.integers(1, 50, size=(N, 1))).astype(np.float32)
        nid_map = np.sort(rng.choice(N, V_sub, replace=False)).astype(np.int64)
        store = FakeStore({"features": torch.from_numpy(feats), "norm": torch.from_numpy(norm)})
        cacher = storage.GraphCacheServer(store, V_sub, torch.from_numpy(nid_map), 0)
        cacher.init_field(["features", "norm"])
        cacher.log = True
        # cache 24 of 64 local ids, in a scrambled order (slot != id)
        cached = rng.permutation(V_sub)[:24].astype(np.int64)
        frame = cacher.get_feat_from_server(torch.from_numpy(cached), ["features", "norm"])
        cacher.cache_fix_data(torch.from_numpy(cached), frame, is_full=False)
        uncached = np.setdiff1d(np.arange(V_sub), cached)
        layers = [
            rng.integers(0, V_sub, size=37),                 # mixed, with duplicates
            rng.choice(cached, 11),                          # all hits (empty-miss)
            np.zeros(0, dtype=np.int64),                     # empty layer
            rng.choice(uncached, 9),                         # all misses (empty-hit)
            np.concatenate([cached[:5], uncached[:5], cached[:5]]),  # repeated rows
        ]
        nf = FakeNodeFlow(layers)
        cacher.fetch_data(nf)
        out = {
            "features_table": feats, "norm_table": norm, "nid_map": nid_map,
            "cached_nids": cached, "num_layers": np.int64(len(layers)),
            "try_num": np.int64(cacher.try_num), "miss_num": np.int64(cacher.miss_num),
            "state_localid2cacheid": cacher.localid2cacheid.numpy().copy(),
            "state_gpu_flag": cacher.gpu_flag.numpy().copy(),
            "state_cached_num": np.int64(cacher.cached_num),
        }
        for i, l in enumerate(layers):
            out[f"layer{i}_nids"] = np.asarray(l, dtype=np.int64)
            out[f"layer{i}_features"] = nf._node_frames[i]["features"].numpy().copy()
            out[f"layer{i}_norm"] = nf._node_frames[i]["norm"].numpy().copy()
        out["miss_rate"] = np.float64(cacher.get_miss_rate())
        np.savez(os.path.join(OUT, f"g1_fetch_data_F{F}.npz"), **out)

        # G3: the full-cache path (cacheid == local id)
        cacher2 = storage.GraphCacheServer(store, V_sub, torch.from_numpy(nid_map), 0)
        cacher2.init_field(["features", "norm"])
        full = torch.arange(V_sub)
        cacher2.cache_fix_data(full, cacher2.get_feat_from_server(full, ["features", "norm"]), is_full=True)
        nf2 = FakeNodeFlow([l for l in layers])
        cacher2.fetch_data(nf2)
        out3 = {"features_table": feats, "norm_table": norm, "nid_map": nid_map,
                "num_layers": np.int64(len(layers))}
        for i, l in enumerate(layers):
            out3[f"layer{i}_nids"] = np.asarray(l, dtype=np.int64)
            out3[f"layer{i}_features"] = nf2._node_frames[i]["features"].numpy().copy()
            out3[f"layer{i}_norm"] = nf2._node_frames[i]["norm"].numpy().copy()
        np.savez(os.path.join(OUT, f"g3_fetch_from_cache_F{F}.npz"), **out3)

    # G5: auto_cache selection (top-capability by out-degree, no ties at the cut)
    N, V_sub, F = 80, 50, 16
    feats = rng.random((N, F), dtype=np.float32)
    norm = rng.random((N, 1), dtype=np.float32)
    nid_map = np.sort(rng.choice(N, V_sub, replace=False)).astype(np.int64)
    store = FakeStore({"features": torch.from_numpy(feats), "norm": torch.from_numpy(norm)})
    out_deg = rng.permutation(V_sub * 3)[:V_sub].astype(np.int64)   # all distinct
    fake_g = types.SimpleNamespace(out_degrees=lambda: torch.from_numpy(out_deg))
    for tag, cap in (("partial", 17), ("full", V_sub + 5)):
        cacher = storage.GraphCacheServer(store, V_sub, torch.from_numpy(nid_map), 0)
        cacher.init_field(["features", "norm"])
        total_dim = F + 1
        want_avail = cap * total_dim * 4 + 3            # int(available/(total_dim*4)) == cap
        torch.cuda.max_memory_allocated = lambda device=None: 1000
        torch.cuda.max_memory_cached = lambda device=None: 2000
        total = want_avail + 1000 + 2000 + 1024 ** 3
        torch.cuda.get_device_properties = lambda dev: types.SimpleNamespace(total_memory=total)
        cacher.auto_cache(fake_g, ["features", "norm"])
        np.savez(os.path.join(OUT, f"g5_auto_cache_{tag}.npz"),
                 features_table=feats, norm_table=norm, nid_map=nid_map, out_degrees=out_deg,
                 total_memory=np.int64(total), peak_allocated=np.int64(1000), peak_cached=np.int64(2000),
                 capability=np.int64(cacher.capability), cached_num=np.int64(cacher.cached_num),
                 full_cached=np.bool_(cacher.full_cached),
                 gpu_flag=cacher.gpu_flag.numpy().copy(),
                 localid2cacheid=cacher.localid2cacheid.numpy().copy(),
                 cache_features=cacher.gpu_fix_cache["features"].numpy().copy(),
                 cache_norm=cacher.gpu_fix_cache["norm"].numpy().copy())


# --------------------------------------------------------------------------
# G4: dg()
# --------------------------------------------------------------------------
def small_graph(rng, V, E, powerlaw):
    if powerlaw:
        w = 1.0 / np.arange(1, V + 1) ** 0.9
        w /= w.sum()
        src = rng.choice(V, E, p=w)
        dst = rng.choice(V, E, p=w)
    else:
        src = rng.integers(0, V, E)
        dst = rng.integers(0, V, E)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    # symmetric like preprocess.py:36-38
    s = np.concatenate([src, dst]); d = np.concatenate([dst, src])
    adj = spsp.coo_matrix((np.ones(len(s), dtype=np.int64), (s, d)), shape=(V, V))
    return adj


def gen_dg(dgmod):
    rng = np.random.default_rng(777)
    cases = []
    for (V, E, pl) in ((60, 150, False), (200, 900, True), (400, 1500, True)):
        for P in (2, 4, 8):
            for hops in (1, 2):
                cases.append((V, E, pl, P, hops))
    # a tie-heavy case: ring graph, every score ties at the start
    cases.append(("ring", 48, None, 4, 1))
    cases.append(("ring", 48, None, 3, 2))
    # round 4 (appended: the files above keep their numbers): hops >= 3 — the `neighs = nids[-1]` quirk of dg.py:22-27, only
    # the LAST appended adjacency list is expanded from depth 2 on — and partition counts at and beyond numpy's stable
    # range: P = 16 (insertion sort), 17 (the first size its introsort partitions), 24, 40, and a ring with P = 20 where
    # every score ties
    cases.append((200, 900, True, 4, 3))
    cases.append((400, 1500, True, 8, 3))
    cases.append((150, 500, False, 3, 4))
    cases.append((400, 1500, True, 16, 1))
    cases.append((400, 1500, True, 16, 2))
    cases.append((400, 1500, True, 17, 1))
    cases.append((500, 2500, True, 24, 2))
    cases.append((600, 2400, False, 40, 1))
    cases.append(("ring", 120, None, 20, 1))
    cases.append((300, 1200, True, 17, 3))
    for idx, (V, E, pl, P, hops) in enumerate(cases):
        if V == "ring":
            V = E
            s = np.arange(V); d = (s + 1) % V
            adj = spsp.coo_matrix((np.ones(2 * V, dtype=np.int64),
                                   (np.concatenate([s, d]), np.concatenate([d, s]))), shape=(V, V))
        else:
            adj = small_graph(rng, V, E, pl)
        train = np.sort(rng.choice(V, int(V * 0.65), replace=False)).astype(np.int64)
        with contextlib.redirect_stdout(open(os.devnull, "w")):
            sub_v, sub_trainv = dgmod.dg(P, adj, train, hops)
        csc = adj.tocsc()
        csc.sum_duplicates(); csc.sort_indices()
        out = {"V": np.int64(V), "P": np.int64(P), "hops": np.int64(hops), "train_nids": train,
               "csc_indptr": csc.indptr.astype(np.int64), "csc_indices": csc.indices.astype(np.int64)}
        for p in range(P):
            out[f"sub_v_{p}"] = sub_v[p].astype(np.int64)
            out[f"sub_trainv_{p}"] = sub_trainv[p].astype(np.int64)
        np.savez(os.path.join(OUT, f"g4_dg_case{idx:02d}.npz"), **out)


# --------------------------------------------------------------------------
# G6: get_sub_graph() (numpy tail pinned; sampler part is a stand-in)
# --------------------------------------------------------------------------
class StandInFullNeighborSampler:
    """Full-neighbour `num_hops` in-edge expansion with per-layer dedup, the
    way this build reads DGL 0.4.1's NeighborSampler(expand_factor=V,
    neighbor_type='in', add_self_loop=False).  NOT reference code."""

    csc = None

    def __init__(self, g, batch_size, expand_factor, neighbor_type="in", shuffle=False,
                 num_workers=1, num_hops=1, seed_nodes=None, prefetch=False):
        assert neighbor_type == "in" and not shuffle
        self.seeds = np.asarray(seed_nodes, dtype=np.int64)
        self.hops = num_hops

    def __iter__(self):
        csc = StandInFullNeighborSampler.csc
        layers = [self.seeds]
        blocks = []
        for _ in range(self.hops):
            dst = layers[0]
            s_all, d_all = [], []
            for v in dst:
                nb = csc.indices[csc.indptr[v]:csc.indptr[v + 1]]
                s_all.append(nb); d_all.append(np.full(len(nb), v, dtype=np.int64))
            s_all = np.concatenate(s_all) if s_all else np.zeros(0, np.int64)
            d_all = np.concatenate(d_all) if d_all else np.zeros(0, np.int64)
            blocks.insert(0, (s_all.astype(np.int64), d_all))
            layers.insert(0, np.unique(s_all).astype(np.int64))
        nf = types.SimpleNamespace()
        nf.num_blocks = self.hops
        # remap_local=False -> ids are NodeFlow-global positions; we hand back
        # parent ids directly and make map_to_parent_nid the identity.
        nf.block_edges = lambda i, remap_local=False: (torch.from_numpy(blocks[i][0]),
                                                       torch.from_numpy(blocks[i][1]), None)
        nf.map_to_parent_nid = lambda t: t
        nf.layer_parent_nid = lambda i: torch.from_numpy(layers[i])
        yield nf


def gen_closure(dgl, utils):
    rng = np.random.default_rng(4242)
    dgl.contrib.sampling.NeighborSampler = StandInFullNeighborSampler
    idx = 0
    for (V, E, pl) in ((80, 160, False), (300, 700, True)):
        adj = small_graph(rng, V, E, pl)
        csc = adj.tocsc(); csc.sum_duplicates(); csc.sort_indices()
        StandInFullNeighborSampler.csc = csc
        for hops in (1, 2):
            for frac in (0.1, 0.4):
                train = np.sort(rng.choice(V, max(2, int(V * frac)), replace=False)).astype(np.int64)
                g = types.SimpleNamespace(number_of_nodes=lambda: V)
                with contextlib.redirect_stdout(open(os.devnull, "w")):
                    csr, sub2full, subtrain = utils.get_sub_graph(g, train, hops)
                np.savez(os.path.join(OUT, f"g6_closure_case{idx:02d}.npz"),
                         V=np.int64(V), hops=np.int64(hops), train_nids=train,
                         csc_indptr=csc.indptr.astype(np.int64), csc_indices=csc.indices.astype(np.int64),
                         sub_indptr=csr.indptr.astype(np.int64), sub_indices=csr.indices.astype(np.int64),
                         sub2full=sub2full.astype(np.int64), subtrainid=subtrain.astype(np.int64))
                idx += 1


# --------------------------------------------------------------------------
# G7/G8: the reference's model classes on a stand-in NodeFlow
# --------------------------------------------------------------------------
class StandInBlockNodeFlow:
    """The NodeFlow surface the reference models touch (gcn_nssc.py:62-76,
    graphsage_nssc.py:75-133): `layers[i].data` dicts, `num_layers`, `block_compute`.
    block_compute is a STAND-IN for DGL 0.4.1's fused copy_src + mean|sum kernel (NOT reference
    code): out[v] = reduce over block i's in-edges of v of src_field[u], zeros when v has none;
    then the reference's own node UDF runs on layer i+1, as DGL's apply_node_func does."""

    def __init__(self, layer_sizes, blocks):
        self.num_layers = len(layer_sizes)
        self.layers = [types.SimpleNamespace(data={}) for _ in layer_sizes]
        self.layer_sizes = list(layer_sizes)
        self.blocks = blocks            # per block: (indptr int64 [n_dst+1], src positions int64 [edges])

    def block_compute(self, i, message_func, reduce_func, apply_node_func=None):
        assert message_func.kind == "copy_src" and reduce_func.msg == message_func.out
        assert reduce_func.kind in ("mean", "sum", "max")
        indptr, src = self.blocks[i]
        h = self.layers[i].data[message_func.src]
        n_dst = self.layer_sizes[i + 1]
        deg = indptr[1:] - indptr[:-1]
        dst = torch.repeat_interleave(torch.arange(n_dst), deg)
        if reduce_func.kind == "max":
            out = _StandInMaxReduce.apply(h, src, dst, n_dst)
        else:
            out = torch.zeros((n_dst, h.shape[1]), dtype=h.dtype).index_add(0, dst, h[src])
        if reduce_func.kind == "mean":
            out = out / deg.clamp(min=1).to(h.dtype).unsqueeze(1)
        d = self.layers[i + 1].data
        d[reduce_func.out] = out
        if apply_node_func is not None:
            d.update(apply_node_func(types.SimpleNamespace(data=d)))


class _StandInMaxReduce(torch.autograd.Function):
    """STAND-IN for DGL 0.4.1's copy_src + max kernel (NOT reference code; DGL is absent): out[v] = element-wise maximum
    of the in-edge messages, zeros without in-edges; backward as DGL's ReduceMax functor [recollection]: (val == accum),
    i.e. every edge that attains the maximum receives the whole gradient."""

    @staticmethod
    def forward(ctx, h, src, dst, n_dst):
        out = torch.full((n_dst, h.shape[1]), float("-inf"), dtype=h.dtype)
        out = out.index_reduce(0, dst, h[src], "amax", include_self=True)
        out = torch.where(torch.isinf(out) & (out < 0), torch.zeros_like(out), out)
        ctx.save_for_backward(h, src, dst, out)
        return out

    @staticmethod
    def backward(ctx, g):
        h, src, dst, out = ctx.saved_tensors
        hit = (h[src] == out[dst]).to(g.dtype)
        return torch.zeros_like(h).index_add(0, src, g[dst] * hit), None, None, None


def _rand_nodeflow(rng, sizes, max_deg):
    """random block structure: destinations get 0..max_deg in-edges (some none, some duplicates)"""
    blocks = []
    for b in range(len(sizes) - 1):
        deg = rng.integers(0, max_deg + 1, size=sizes[b + 1])
        deg[rng.integers(0, sizes[b + 1])] = 0                       # at least one destination without in-edges
        indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int64)
        src = rng.integers(0, sizes[b], size=int(indptr[-1])).astype(np.int64)
        blocks.append((indptr, src))
    return blocks


def _run_model_case(name, model, sizes, blocks, frames, rng, extra):
    nf = StandInBlockNodeFlow(sizes, [(torch.from_numpy(ip), torch.from_numpy(sr)) for ip, sr in blocks])
    for i, fr in enumerate(frames):
        for k, v in fr.items():
            nf.layers[i].data[k] = torch.from_numpy(v)
    model.train()        # dropout p = 0: train mode is the path pa_gcn.py runs
    logits = model(nf)
    G = rng.standard_normal(tuple(logits.shape)).astype(np.float32)
    (logits * torch.from_numpy(G)).sum().backward()
    out = dict(extra)
    out["num_layers"] = np.int64(len(sizes))
    out["layer_sizes"] = np.asarray(sizes, np.int64)
    for b, (ip, sr) in enumerate(blocks):
        out[f"blk{b}_indptr"] = ip
        out[f"blk{b}_src"] = sr
    for i, fr in enumerate(frames):
        for k, v in fr.items():
            out[f"layer{i}_{k}"] = v
    for k, v in model.state_dict().items():
        out[f"param:{k}"] = v.detach().numpy().copy()
    for k, v in model.named_parameters():
        out[f"grad:{k}"] = (v.grad if v.grad is not None else torch.zeros_like(v)).numpy().copy()
    out["logits"] = logits.detach().numpy().copy()
    out["G"] = G
    np.savez(os.path.join(OUT, name), **out)


def gen_models(gcn, sage):
    import torch.nn.functional as Fn
    rng = np.random.default_rng(70807)
    Fdim, H, C = 40, 32, 7
    # ---- G7: gcn_nssc.py ---------------------------------------------------------------
    cases = [("gcn_L1", dict(n_layers=1, preprocess=False, infer=False), [90, 40, 16]),
             ("gcn_L2", dict(n_layers=2, preprocess=False, infer=False), [120, 70, 30, 12]),
             ("gcn_pre_L1", dict(n_layers=1, preprocess=True, infer=False), [60, 20]),
             ("gcn_pre_L2", dict(n_layers=2, preprocess=True, infer=False), [90, 40, 16]),
             ("infer_L1", dict(n_layers=1, preprocess=False, infer=True), [90, 40, 16]),
             ("infer_pre_L1", dict(n_layers=1, preprocess=True, infer=True), [60, 20])]
    for idx, (tag, cfg, sizes) in enumerate(cases):
        torch.manual_seed(100 + idx)
        if cfg["infer"]:
            model = gcn.GCNInfer(Fdim, H, C, cfg["n_layers"], Fn.relu, preprocess=cfg["preprocess"])
        else:
            model = gcn.GCNSampling(Fdim, H, C, cfg["n_layers"], Fn.relu, 0.0, preprocess=cfg["preprocess"])
        blocks = _rand_nodeflow(rng, sizes, 5)
        frames = [{"features": rng.random((sizes[0], Fdim), dtype=np.float32)}] + [{} for _ in sizes[1:]]
        if cfg["infer"]:
            for i in range(len(sizes)):
                frames[i]["norm"] = (1.0 / rng.integers(1, 30, size=(sizes[i], 1))).astype(np.float32)
        _run_model_case(f"g7_{tag}.npz", model, sizes, blocks, frames, rng,
                        dict(arch=np.str_("gcn_infer" if cfg["infer"] else "gcn"), n_layers=np.int64(cfg["n_layers"]),
                             preprocess=np.bool_(cfg["preprocess"]), in_feats=np.int64(Fdim), n_hidden=np.int64(H),
                             n_classes=np.int64(C)))
    # ---- G8: graphsage_nssc.py -----------------------------------------------------------
    H = 16
    cases = [("sage_mean_L1", dict(n_layers=1, preprocess=False, agg="mean"), [90, 40, 16]),
             ("sage_mean_L2", dict(n_layers=2, preprocess=False, agg="mean"), [120, 70, 30, 12]),
             ("sage_gcn_L1", dict(n_layers=1, preprocess=False, agg="gcn"), [90, 40, 16]),
             ("sage_mean_pre_L1", dict(n_layers=1, preprocess=True, agg="mean"), [60, 20]),
             ("sage_mean_pre_L2", dict(n_layers=2, preprocess=True, agg="mean"), [90, 40, 16]),
             # appended in round 3 (the generator's draws for the cases above are unchanged)
             ("sage_pool_L1", dict(n_layers=1, preprocess=False, agg="pool"), [90, 40, 16]),
             ("sage_pool_L2", dict(n_layers=2, preprocess=False, agg="pool"), [120, 70, 30, 12]),
             ("sage_pool_pre_L1", dict(n_layers=1, preprocess=True, agg="pool"), [60, 20])]
    for idx, (tag, cfg, sizes) in enumerate(cases):
        torch.manual_seed(200 + idx)
        model = sage.GraphSageSampling(Fdim, H, C, cfg["n_layers"], Fn.relu, 0.0, cfg["agg"], cfg["preprocess"])
        blocks = _rand_nodeflow(rng, sizes, 5)
        frames = [{"features": rng.random((n, Fdim), dtype=np.float32)} for n in sizes]
        if cfg["preprocess"]:
            for i, n in enumerate(sizes):
                frames[i]["neigh"] = rng.random((n, Fdim), dtype=np.float32)
        _run_model_case(f"g8_{tag}.npz", model, sizes, blocks, frames, rng,
                        dict(arch=np.str_("sage"), n_layers=np.int64(cfg["n_layers"]), preprocess=np.bool_(cfg["preprocess"]),
                             aggregator=np.str_(cfg["agg"]), in_feats=np.int64(Fdim), n_hidden=np.int64(H),
                             n_classes=np.int64(C)))


# --------------------------------------------------------------------------
# G9: cache-policy analysis helpers of examples/opt_cache_hit.py and examples/count_vnum.py
# --------------------------------------------------------------------------
def gen_analysis():
    sys.path.insert(0, os.path.join(REF, "examples"))
    och = importlib.import_module("opt_cache_hit")      # count_vertex_freq, optimal_cache_hit, count_nf_vnum
    cvn = importlib.import_module("count_vnum")
    rng = np.random.default_rng(909)
    V = 500
    w = 1.0 / np.arange(1, V + 1) ** 1.1
    w /= w.sum()
    freq = np.zeros(V, dtype=np.int64)
    out = {"V": np.int64(V)}
    vnum = 0
    n_nf = 12
    for t in range(n_nf):
        layers = [np.unique(rng.choice(V, 160, p=w)), np.unique(rng.choice(V, 60, p=w)),
                  rng.choice(V, 25, p=w)]              # seed layer: in seed order, WITH repeats
        nf = types.SimpleNamespace(num_layers=3,
                                   layer_parent_nid=lambda i, L=layers: torch.from_numpy(L[i]),
                                   layer_nid=lambda i, L=layers: torch.from_numpy(L[i]))
        och.count_vertex_freq(nf, freq)
        vnum += cvn.count_nf_vnum(nf)
        for i, l in enumerate(layers):
            out[f"nf{t}_layer{i}"] = l.astype(np.int64)
    out["num_nodeflows"] = np.int64(n_nf)
    out["freq"] = freq.copy()
    out["vnum"] = np.int64(vnum)
    for r in (0.05, 0.2, 0.5):
        out[f"opt_hit_{int(r * 100):02d}"] = np.float64(och.optimal_cache_hit(freq, r))
    np.savez(os.path.join(OUT, "g9_cache_analysis.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    dgl = install_stubs()
    cpu_shims()
    storage = importlib.import_module("PaGraph.storage.storage")
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        gen_storage(storage)
    utils = importlib.import_module("utils")        # PaGraph/partition/utils.py
    dgmod = importlib.import_module("dg")           # PaGraph/partition/dg.py
    gen_dg(dgmod)
    gen_closure(dgl, utils)
    gen_models(importlib.import_module("PaGraph.model.gcn_nssc"), importlib.import_module("PaGraph.model.graphsage_nssc"))
    gen_analysis()
    n = len([f for f in os.listdir(OUT) if f.endswith(".npz")])
    sz = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {n} fixtures, {sz/1e6:.2f} MB -> {os.path.normpath(OUT)}")


if __name__ == "__main__":
    main()
