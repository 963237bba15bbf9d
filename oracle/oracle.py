"""Python face of the CPU oracle (TEST INFRASTRUCTURE — see pgc_oracle.c header).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  numpy restatements follow the reference's numpy/torch code line by line
(citations inline); loops that would be slow in Python live in pgc_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i64p = ctypes.POINTER(ctypes.c_int64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_f32p = ctypes.POINTER(ctypes.c_float)


def build():
    """compile pgc_oracle.c -> libpgc_oracle.so (gcc; seconds)"""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libpgc_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.pgc_fetch_rows.restype = ctypes.c_int64
        L.pgc_fetch_rows.argtypes = [_i64p, ctypes.c_int64, _u8p, _i64p, _i64p, _f32p, _f32p, ctypes.c_int32, _f32p]
        L.pgc_sample_nodeflow.restype = ctypes.c_int
        L.pgc_sample_nodeflow.argtypes = [_i64p, _i32p, _i64p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                          ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _i64p, _i32p, _i32p,
                                          _i64p, _i32p, _i64p, _i32p]
        L.pgc_sample_epoch.restype = ctypes.c_int64
        L.pgc_sample_epoch.argtypes = [_i64p, _i32p, _i64p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int64,
                                       ctypes.c_int64, _i64p, ctypes.c_int64]
        L.pgc_spmm_fwd.restype = None
        L.pgc_spmm_fwd.argtypes = [_i32p, _i32p, _f32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int, _f32p]
        L.pgc_spmm_bwd.restype = None
        L.pgc_spmm_bwd.argtypes = [_i32p, _i32p, _f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                   ctypes.c_int, _f32p]
        L.pgc_spmm_fwd_max.restype = None
        L.pgc_spmm_fwd_max.argtypes = [_i32p, _i32p, _f32p, ctypes.c_int64, ctypes.c_int32, _f32p]
        L.pgc_spmm_bwd_max.restype = None
        L.pgc_spmm_bwd_max.argtypes = [_i32p, _i32p, _f32p, _f32p, _f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                       _f32p]
        L.pgc_rmat_edges.restype = None
        L.pgc_rmat_edges.argtypes = [ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_uint32,
                                     ctypes.c_uint32, ctypes.c_int64, ctypes.c_int64, _i64p, _i64p]
        L.pgc_random_features.restype = None
        L.pgc_random_features.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, _f32p,
                                          ctypes.c_int64]
        L.pgc_philox4x32_10.restype = None
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


# --------------------------------------------------------------------------
# RNG
# --------------------------------------------------------------------------
def philox4x32_10(ctr, key):
    c = (ctypes.c_uint32 * 4)(*[int(x) & 0xFFFFFFFF for x in ctr])
    k = (ctypes.c_uint32 * 2)(*[int(x) & 0xFFFFFFFF for x in key])
    o = (ctypes.c_uint32 * 4)()
    lib().pgc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def philox4x32_10_np(c0, c1, c2, c3, k0, k1):
    """vectorised Philox4x32-10 over numpy arrays (uint64 arithmetic); checked against the C one in tests"""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    mask = np.uint64(0xFFFFFFFF)
    c = [np.asarray(x, dtype=np.uint64) & mask for x in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & mask,
             (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & mask]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c


def dropout_threshold(p):
    return int(round(float(p) * 65536.0))


def dropout_mask(rows, dim, threshold, seed, tag, step):
    """keep-mask [rows, dim] of the dropout folded into the aggregation (the spec in include/pagraph_hip.h,
    pg_dropout_t; it stands for nn.Dropout at gcn_nssc.py:66-69 / graphsage_nssc.py:86-89) and the
    float32 scale of the kept values."""
    col = np.arange(dim, dtype=np.int64)
    piece = col >> 2
    q = ((piece >> 7) << 6) | (piece & 63)
    half = (piece >> 6) & 1
    j = col & 3
    uq, inv = np.unique(q, return_inverse=True)
    r = np.arange(rows, dtype=np.uint64)[:, None]
    w = philox4x32_10_np(r, uq[None, :].astype(np.uint64), np.uint64(tag), np.uint64(int(step) & 0xFFFFFFFF),
                         int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    w = np.stack(w, axis=-1)                                      # [rows, |uq|, 4]
    word = w[:, inv, 2 * half + (j >> 1)]                         # [rows, dim]
    u16 = np.where(j & 1, word >> np.uint64(16), word & np.uint64(0xFFFF))
    scale = np.float32(65536.0) / np.float32(65536 - threshold)
    return u16 >= np.uint64(threshold), scale


# --------------------------------------------------------------------------
# feature cache (PaGraph/storage/storage.py)
# --------------------------------------------------------------------------
class CacheState:
    """The reference's per-partition state (storage.py:31-56) as numpy arrays."""

    def __init__(self, node_num, nid_map):
        self.node_num = int(node_num)
        self.nid_map = _c(nid_map, np.int64)                      # :34
        self.gpu_flag = np.zeros(node_num, dtype=np.uint8)        # :38
        self.localid2cacheid = np.zeros(node_num, dtype=np.int64)  # :50
        self.cached_num = 0
        self.full_cached = False
        self.cache = {}                                           # gpu_fix_cache, :48
        self.try_num = 0
        self.miss_num = 0

    def cache_fix_data(self, nids, tables, is_full=False):
        """storage.py:135-154; `tables` = full host tables, rows fetched as :117-131"""
        nids = _c(nids, np.int64)
        rows = len(nids)
        self.localid2cacheid[nids] = np.arange(rows)              # :145
        self.cached_num = rows                                    # :146
        for name, tab in tables.items():
            self.cache[name] = np.ascontiguousarray(tab[self.nid_map[nids]], dtype=np.float32)  # :117,131,151
        self.gpu_flag[nids] = 1                                   # :153
        self.full_cached = is_full                                # :154

    def auto_cache_select(self, out_degrees, capability):
        """storage.py:90-104: full when capability >= node_num, else top-`capability`
        by out-degree, descending.  torch.argsort there is unstable; the build
        defines ties as 'lower id first' (stable sort on -degree)."""
        if capability >= self.node_num:
            return np.arange(self.node_num, dtype=np.int64), True
        order = np.argsort(-np.asarray(out_degrees, dtype=np.int64), kind="stable")
        return order[:capability].astype(np.int64), False

    def fetch_layer(self, tnid, tables):
        """storage.py:176-204 for one layer; returns {name: rows}"""
        tnid = _c(tnid, np.int64)
        out = {}
        miss = 0
        for name, tab in tables.items():
            tab = _c(tab, np.float32)
            dim = tab.shape[1]
            o = np.empty((len(tnid), dim), dtype=np.float32)
            cache = self.cache.get(name)
            if cache is None or cache.size == 0:
                cache = np.zeros((1, dim), dtype=np.float32)
            miss = lib().pgc_fetch_rows(_p(tnid, _i64p), len(tnid), _p(self.gpu_flag, _u8p),
                                        _p(self.localid2cacheid, _i64p), _p(self.nid_map, _i64p),
                                        _p(cache, _f32p), _p(tab, _f32p), dim, _p(o, _f32p))
            out[name] = o
        self.try_num += len(tnid)                                 # :203-204,219-221
        self.miss_num += int(miss)
        return out

    def get_miss_rate(self):
        """storage.py:223-227"""
        r = float(self.miss_num) / self.try_num
        self.miss_num = 0
        self.try_num = 0
        return r


def fetch_layer_numpy(state, tnid, tables):
    """Pure-numpy twin of CacheState.fetch_layer (same lines), used to cross-check the C loop."""
    tnid = np.asarray(tnid, dtype=np.int64)
    mask = state.gpu_flag[tnid].astype(bool)                      # :179
    out = {}
    for name, tab in tables.items():
        o = np.empty((len(tnid), tab.shape[1]), dtype=np.float32)
        if mask.any():
            o[mask] = state.cache[name][state.localid2cacheid[tnid[mask]]]  # :191-193
        if (~mask).any():
            o[~mask] = tab[state.nid_map[tnid[~mask]]]            # :117,128,199-200
        out[name] = o
    return out, int((~mask).sum())


# --------------------------------------------------------------------------
# sampler (build-defined spec)
# --------------------------------------------------------------------------
def nodeflow_caps(batch, k, hops):
    caps = [0] * (hops + 1)
    caps[hops] = batch
    for l in range(hops - 1, -1, -1):
        caps[l] = caps[l + 1] * k
    return caps


def sample_nodeflow(indptr, indices, seeds, k, hops, seed, epoch, batch):
    """returns dict(node_mapping, layer_offsets, blocks=[(indptr, src)] layer-0 block first)"""
    indptr = _c(indptr, np.int64)
    indices = _c(indices, np.int32)
    seeds = _c(seeds, np.int64)
    caps = nodeflow_caps(len(seeds), k, hops)
    nm = np.zeros(sum(caps) + 1, dtype=np.int64)
    lo = np.zeros(hops + 2, dtype=np.int32)
    ioff = np.zeros(8, dtype=np.int64)
    soff = np.zeros(8, dtype=np.int64)
    ic = sc = 0
    for b in range(hops - 1, -1, -1):
        ioff[b], soff[b] = ic, sc
        ic += caps[b + 1] + 1
        sc += caps[b + 1] * k
    bip = np.zeros(ic + 1, dtype=np.int32)
    bsr = np.zeros(sc + 1, dtype=np.int32)
    eo = np.zeros(8, dtype=np.int32)
    rc = lib().pgc_sample_nodeflow(_p(indptr, _i64p), _p(indices, _i32p), _p(seeds, _i64p), len(seeds), k, hops,
                                   ctypes.c_uint64(seed), epoch, batch, _p(nm, _i64p), _p(lo, _i32p),
                                   _p(bip, _i32p), _p(ioff, _i64p), _p(bsr, _i32p), _p(soff, _i64p), _p(eo, _i32p))
    if rc != 0:
        raise RuntimeError("pgc_sample_nodeflow failed")
    offs = lo[:hops + 2].copy()
    blocks = []
    for b in range(hops):
        nd = offs[b + 2] - offs[b + 1]
        blocks.append((bip[ioff[b]:ioff[b] + nd + 1].copy(), bsr[soff[b]:soff[b] + eo[b]].copy()))
    return {"node_mapping": nm[:offs[hops + 1]].copy(), "layer_offsets": offs, "blocks": blocks}


def sample_epoch_rows(indptr, indices, seeds, batch_size, k, hops, seed, epoch, first_batch, n_batches,
                      keep_rows=False):
    indptr = _c(indptr, np.int64)
    indices = _c(indices, np.int32)
    seeds = _c(seeds, np.int64)
    cap = sum(nodeflow_caps(batch_size, k, hops))
    rows = np.full((n_batches, cap), -1, dtype=np.int64) if keep_rows else None
    tot = lib().pgc_sample_epoch(_p(indptr, _i64p), _p(indices, _i32p), _p(seeds, _i64p), len(seeds), batch_size, k,
                                 hops, ctypes.c_uint64(seed), epoch, first_batch, n_batches,
                                 _p(rows, _i64p) if keep_rows else None, cap)
    return int(tot), rows


# --------------------------------------------------------------------------
# aggregation + model (PaGraph/model/*.py)
# --------------------------------------------------------------------------
def spmm_fwd(indptr, src, h, n_dst, reduce="mean"):
    h = _c(h, np.float32)
    indptr = _c(indptr, np.int32)
    src = _c(src, np.int32)
    out = np.empty((n_dst, h.shape[1]), dtype=np.float32)
    if reduce == "max":
        lib().pgc_spmm_fwd_max(_p(indptr, _i32p), _p(src, _i32p), _p(h, _f32p), n_dst, h.shape[1], _p(out, _f32p))
        return out
    lib().pgc_spmm_fwd(_p(indptr, _i32p), _p(src, _i32p), _p(h, _f32p), n_dst, h.shape[1],
                       1 if reduce == "mean" else 0, _p(out, _f32p))
    return out


def spmm_bwd(indptr, src, grad_out, n_src, reduce="mean"):
    go = _c(grad_out, np.float32)
    indptr = _c(indptr, np.int32)
    src = _c(src, np.int32)
    gh = np.empty((n_src, go.shape[1]), dtype=np.float32)
    lib().pgc_spmm_bwd(_p(indptr, _i32p), _p(src, _i32p), _p(go, _f32p), go.shape[0], n_src, go.shape[1],
                       1 if reduce == "mean" else 0, _p(gh, _f32p))
    return gh


def spmm_bwd_max(indptr, src, grad_out, x, out):
    """backward of spmm_fwd(..., 'max'): x = the messages the forward aggregated [n_src, dim], out = its result"""
    go = _c(grad_out, np.float32)
    x = _c(x, np.float32)
    out = _c(out, np.float32)
    indptr = _c(indptr, np.int32)
    src = _c(src, np.int32)
    gh = np.empty_like(x)
    lib().pgc_spmm_bwd_max(_p(indptr, _i32p), _p(src, _i32p), _p(go, _f32p), _p(x, _f32p), _p(out, _f32p), go.shape[0],
                           x.shape[0], go.shape[1], _p(gh, _f32p))
    return gh


def _relu(x):
    return np.maximum(x, 0)


def cross_entropy(logits, labels, ignore_index=-100):
    """torch.nn.CrossEntropyLoss() of examples/profile/pa_gcn.py:80,101-104 in float64: mean over the rows whose
    label != ignore_index of logsumexp(x) - x[label]; also d loss / d logits."""
    x = np.asarray(logits, np.float64)
    lab = np.asarray(labels, np.int64)
    valid = lab != ignore_index
    m = x.max(1, keepdims=True)
    lse = (m + np.log(np.exp(x - m).sum(1, keepdims=True)))[:, 0]
    idx = np.where(valid, lab, 0)
    rows = lse - x[np.arange(x.shape[0]), idx]
    cnt = int(valid.sum())
    loss = rows[valid].sum() / cnt if cnt else float("nan")
    grad = np.exp(x - lse[:, None])
    grad[np.arange(x.shape[0]), idx] -= 1.0
    grad[~valid] = 0.0
    return loss, grad / max(cnt, 1)


def gcn_head(indptr, src, h, W, b, labels, ignore_index=-100, grad_scale=1.0, reduce="mean", keep=None, scale=1.0):
    """output head of the sampled GCN in float64: agg = reduce(dropout(h)[src]) (gcn_nssc.py:66-74), z = agg W^T + b
    (:18,58), CrossEntropyLoss (pa_gcn.py:80,101-104) and the gradients w.r.t. h, W, b, all times grad_scale.
    keep/scale: the dropout keep-mask over h's elements (dropout_mask) or None."""
    h = np.asarray(h, np.float64)
    hd = h if keep is None else np.where(keep, h * float(scale), 0.0)
    indptr = np.asarray(indptr, np.int64)
    n_dst = len(indptr) - 1
    deg = np.diff(indptr)
    dst = np.repeat(np.arange(n_dst), deg)
    agg = np.zeros((n_dst, h.shape[1]))
    np.add.at(agg, dst, hd[np.asarray(src, np.int64)])
    wgt = np.ones(n_dst)
    if reduce == "mean":
        wgt = np.where(deg > 0, 1.0 / np.maximum(deg, 1), 1.0)
        agg = agg * wgt[:, None]
    z = agg @ np.asarray(W, np.float64).T + (0 if b is None else np.asarray(b, np.float64))
    loss, dz = cross_entropy(z, labels, ignore_index)
    dz = dz * grad_scale
    dagg = dz @ np.asarray(W, np.float64)
    dW = dz.T @ agg
    db = dz.sum(0)
    dh = np.zeros_like(h)
    np.add.at(dh, np.asarray(src, np.int64), dagg[dst] * wgt[dst][:, None])
    if keep is not None:
        dh = np.where(keep, dh * float(scale), 0.0)
    return loss, z, dh, dW, db


def _act(z, concat, has_act):
    """NodeUpdate's tail (gcn_nssc.py:19-23, graphsage_nssc.py:25-29): skip-concat, else activation (ReLU), else none"""
    if concat:
        return np.concatenate([z, _relu(z)], axis=1)
    return _relu(z) if has_act else z


def _lin(state, prefix, x):
    return x @ np.asarray(state[prefix + ".weight"], np.float32).T + np.asarray(state[prefix + ".bias"], np.float32)


def gcn_model_forward(blocks, layer_sizes, frames, state, n_layers, preprocess=False, infer=False):
    """GCNSampling.forward / preprocess_forward (gcn_nssc.py:60-100) and GCNInfer.forward / preprocess_forward
    (:130-164) with NodeUpdate (:14-24), dropout off, activation = ReLU. Pinned by tests/golden/g7_*.
    blocks[i] = (indptr, src positions) of block i; frames[l] = {field: array} of NodeFlow layer l;
    state = {reference parameter name: array} ('layers.N.linear.weight', 'linear.weight', ...)."""
    reduce = "sum" if infer else "mean"                               # :72-73 vs :140-141
    L = len(layer_sizes)
    h = np.asarray(frames[0]["features"], np.float32)                 # :62 / :81
    if preprocess:
        z = _lin(state, "linear", h)                                  # :84
        h = _act(z, n_layers == 1, True)                              # :86-90
    n_upd = n_layers + 1 - (1 if preprocess else 0)                   # :45-58: len(self.layers)
    assert n_upd == L - 1, "one NodeUpdate per block"
    acts = []
    for i in range(n_upd):
        ip, sr = blocks[i]
        agg = spmm_fwd(ip, sr, h, layer_sizes[i + 1], reduce)         # :71-74 / :94-97
        if infer:
            agg = agg * np.asarray(frames[i + 1]["norm"], np.float32)  # :16-17 (test=True)
        z = _lin(state, f"layers.{i}.linear", agg)                    # :18
        last = i == n_upd - 1                                         # :58 output layer: no activation, no concat
        lid = i + (1 if preprocess else 0)                            # index in the reference's layer numbering
        h = z if last else _act(z, lid == n_layers - 1, True)         # :52,56 skip_start
        acts.append(h)
    return h, acts


def gcn_forward(nf, feats0, params, n_layers=1):
    """GCNSampling.forward, dropout off (gcn_nssc.py:60-77). params: list of (W[out,in], b[out]) per NodeUpdate;
    nf from sample_nodeflow; feats0 = layer-0 features. Thin wrapper over gcn_model_forward."""
    offs = nf["layer_offsets"]
    sizes = [offs[i + 1] - offs[i] for i in range(len(params) + 1)]
    state = {}
    for i, (W, b) in enumerate(params):
        state[f"layers.{i}.linear.weight"], state[f"layers.{i}.linear.bias"] = W, b
    frames = [{"features": feats0}] + [{} for _ in sizes[1:]]
    return gcn_model_forward(nf["blocks"], sizes, frames, state, n_layers)


def sage_model_forward(blocks, layer_sizes, frames, state, n_layers, aggregator="mean", preprocess=False):
    """GraphSageSampling.forward (graphsage_nssc.py:74-134) with NodeUpdate (:21-30), aggregators 'mean' (:98-101),
    'gcn' (:102-105) and 'pool' (:106-110), dropout off, activation = ReLU. Pinned by tests/golden/g8_*. Arguments as
    gcn_model_forward; under preprocess every layer's frame also holds 'neigh' (:77)."""
    reduce = {"mean": "mean", "gcn": "sum", "pool": "max"}[aggregator]                  # :98-110
    L = len(layer_sizes)                                              # nf.num_layers
    if preprocess:
        h = []
        for i in range(L):                                            # :76-87
            z = _lin(state, "fc_self", np.asarray(frames[i]["features"], np.float32)) + \
                _lin(state, "fc_neigh", np.asarray(frames[i]["neigh"], np.float32))
            h.append(_act(z, n_layers == 1, True))
    else:
        h = [np.asarray(frames[i]["features"], np.float32) for i in range(L)]   # :89-90
    n_upd = n_layers + 1 - (1 if preprocess else 0)
    for lid in range(n_upd):                                          # :92
        act = {}
        for i in range(lid, L - 1):                                   # :93
            ip, sr = blocks[i]
            neigh = spmm_fwd(ip, sr, h[i], layer_sizes[i + 1], reduce)
            z = _lin(state, f"layers.{lid}.fc_self", h[i + 1]) + _lin(state, f"layers.{lid}.fc_neigh", neigh)   # :24
            last = lid == n_upd - 1                                   # :71 output layer
            ref_lid = lid + (1 if preprocess else 0)
            act[i + 1] = z if last else _act(z, ref_lid == n_layers - 1, True)
        for i in range(lid + 1, L):                                   # :129-131
            h[i] = act[i]
    return h[L - 1]                                                   # :133


def sage_forward(nf, feats_by_layer, params, n_layers=1):
    """GraphSageSampling.forward, aggregator 'mean', dropout off. params: list of (W_self, b_self, W_neigh, b_neigh);
    feats_by_layer[l] = features of NodeFlow layer l. Thin wrapper over sage_model_forward."""
    offs = nf["layer_offsets"]
    sizes = [offs[i + 1] - offs[i] for i in range(len(feats_by_layer))]
    state = {}
    for i, (Ws, bs, Wn, bn) in enumerate(params):
        state[f"layers.{i}.fc_self.weight"], state[f"layers.{i}.fc_self.bias"] = Ws, bs
        state[f"layers.{i}.fc_neigh.weight"], state[f"layers.{i}.fc_neigh.bias"] = Wn, bn
    return sage_model_forward(nf["blocks"], sizes, [{"features": f} for f in feats_by_layer], state, n_layers, "mean")


# --------------------------------------------------------------------------
# cache-policy analysis (examples/opt_cache_hit.py, examples/count_vnum.py) — pinned by tests/golden/g9_*
# --------------------------------------------------------------------------
def count_nf_vnum(layers):
    """count_vnum.py:16-20: rows of one NodeFlow, every layer"""
    return int(sum(len(l) for l in layers))


def count_vertex_freq(layers, freq):
    """opt_cache_hit.py:22-24: `freq[nf.layer_parent_nid(lid)] += 1` — numpy's fancy-index add counts a vertex once
    per layer even when the layer lists it several times"""
    for ids in layers:
        freq[np.asarray(ids, np.int64)] += 1


def optimal_cache_hit(freq, cached):
    """opt_cache_hit.py:26-31"""
    num = int(freq.shape[0] * cached)
    total = np.sum(freq)
    sorted_freq = np.sort(freq)
    hit_time = np.sum(sorted_freq[-num:])
    return hit_time / total


# --------------------------------------------------------------------------
# partitioning (PaGraph/partition/dg.py, utils.py) — small-case Python restatements
# --------------------------------------------------------------------------
def _less(a, b):
    return a < b or (b != b and a == a)

def _aheapsort(v, t, lo, n):
    # t[lo .. lo+n-1], 1-based view a[i] = t[lo + i - 1]
    a = lambda i: t[lo + i - 1]
    def seta(i, x): t[lo + i - 1] = x
    l = n >> 1
    while l > 0:
        tmp = a(l); i = l; j = l << 1
        while j <= n:
            if j < n and _less(v[a(j)], v[a(j + 1)]): j += 1
            if _less(v[tmp], v[a(j)]):
                seta(i, a(j)); i = j; j += j
            else: break
        seta(i, tmp); l -= 1
    while n > 1:
        tmp = a(n); seta(n, a(1)); n -= 1
        i = 1; j = 2
        while j <= n:
            if j < n and _less(v[a(j)], v[a(j + 1)]): j += 1
            if _less(v[tmp], v[a(j)]):
                seta(i, a(j)); i = j; j += j
            else: break
        seta(i, tmp)

def numpy_scalar_argsort(v):
    """np.argsort(v) with the default, UNSTABLE kind as dg.py:31 calls it, on numpy's portable scalar path (numpy 2.2,
    numpy/_core/src/npysort/quicksort.cpp aquicksort_ / heapsort.cpp aheapsort_, restated from the algorithm): introsort —
    median-of-3 Hoare partitions while a range spans more than 16 elements (pr - pl > 15), insertion sort below (so up to
    16 partitions the sort is stable), heapsort when the depth budget 2 * floor(log2(n)) runs out. The tie order is part of
    dg's result (every first assignment ties). Pinned against numpy itself with its SIMD dispatch disabled
    (tests/test_oracle_golden.py) and, through dg(), by the G4 fixtures with P > 16."""
    v = [float(x) for x in v]
    num = len(v)
    t = list(range(num))
    if num < 2: return np.asarray(t, dtype=np.int64)
    pl, pr = 0, num - 1
    stack, depth = [], []
    cdepth = (num.bit_length() - 1) * 2
    while True:
        if cdepth < 0:
            _aheapsort(v, t, pl, pr - pl + 1)
        else:
            while pr - pl > 15:
                pm = pl + ((pr - pl) >> 1)
                if _less(v[t[pm]], v[t[pl]]): t[pm], t[pl] = t[pl], t[pm]
                if _less(v[t[pr]], v[t[pm]]): t[pr], t[pm] = t[pm], t[pr]
                if _less(v[t[pm]], v[t[pl]]): t[pm], t[pl] = t[pl], t[pm]
                vp = v[t[pm]]
                pi, pj = pl, pr - 1
                t[pm], t[pj] = t[pj], t[pm]
                while True:
                    pi += 1
                    while _less(v[t[pi]], vp): pi += 1
                    pj -= 1
                    while _less(vp, v[t[pj]]): pj -= 1
                    if pi >= pj: break
                    t[pi], t[pj] = t[pj], t[pi]
                pk = pr - 1
                t[pi], t[pk] = t[pk], t[pi]
                if pi - pl < pr - pi:
                    stack.append((pi + 1, pr)); pr = pi - 1
                else:
                    stack.append((pl, pi - 1)); pl = pi + 1
                cdepth -= 1
                depth.append(cdepth)
            for pi in range(pl + 1, pr + 1):
                vi = t[pi]; vp = v[vi]; pj = pi
                while pj > pl and _less(vp, v[t[pj - 1]]):
                    t[pj] = t[pj - 1]; pj -= 1
                t[pj] = vi
        if not stack: break
        pl, pr = stack.pop()
        cdepth = depth.pop()
    return np.asarray(t, dtype=np.int64)


def dg_partition(P, indptr, indices, V, train_nids, hops):
    """dg.py:59-103 restated with explicit loops (slow; small graphs only)."""
    def in_nb(n):
        return indices[indptr[n]:indptr[n + 1]]

    def in_nb_hop(nid):                                           # dg.py:18-27
        if hops == 1:
            return in_nb(nid)
        nids = []
        for _depth in range(hops):
            neighs = nids[-1] if nids else [nid]
            for n in neighs:
                nids.append(in_nb(n))
        return np.unique(np.hstack(nids))

    belongs = -np.ones(V, dtype=np.int8)
    r_belongs = [np.zeros(V, dtype=bool) for _ in range(P)]
    p_vnum = np.zeros(P, dtype=np.int64)
    r_vnum = np.zeros(P, dtype=np.int64)
    for nid in train_nids:
        nb = in_nb_hop(nid)
        com = np.ones(P, dtype=np.int64)                          # :47
        nbb = belongs[nb]
        bel = nbb[nbb != -1]
        pid, freq = np.unique(bel, return_counts=True)
        com[pid] += freq
        avg = V * 0.65 / P                                        # :54
        score = com * (-p_vnum + avg) / (r_vnum + 1)              # :55
        ids = numpy_scalar_argsort(score)[-2:]                    # :31 (default kind: stable only up to P = 16)
        if score[ids[0]] != score[ids[1]]:
            ind = ids[1]
        else:
            ind = ids[0] if p_vnum[ids[0]] < p_vnum[ids[1]] else ids[1]
        if belongs[nid] == -1:                                    # :76-83
            belongs[nid] = ind
            p_vnum[ind] += 1
            for u in np.append(nb, nid):
                if not r_belongs[ind][u]:
                    r_belongs[ind][u] = True
                    r_vnum[ind] += 1
    sub_v = [np.where(r_belongs[p])[0] for p in range(P)]
    sub_trainv = [np.where(belongs == p)[0] for p in range(P)]
    return sub_v, sub_trainv


def closure_subgraph(indptr, indices, V, train_nid, num_hops):
    """utils.py:9-52 get_sub_graph with DGL's sampler replaced by its spec
    (full-neighbour per-layer expansion, per-layer dedup). Returns
    (sub_indptr, sub_indices [CSR row=src col=dst], sub2full, subtrainid)."""
    import scipy.sparse as spsp
    train_nid = np.asarray(train_nid, dtype=np.int64)
    layer = train_nid
    srcs, dsts = [], []
    for _ in range(num_hops):
        s_all = [indices[indptr[v]:indptr[v + 1]].astype(np.int64) for v in layer]
        d_all = [np.full(indptr[v + 1] - indptr[v], v, dtype=np.int64) for v in layer]
        s = np.concatenate(s_all) if s_all else np.zeros(0, np.int64)
        d = np.concatenate(d_all) if d_all else np.zeros(0, np.int64)
        srcs.append(s); dsts.append(d)
        layer = np.unique(s)
    full_srcs = np.concatenate(srcs[::-1]); full_dsts = np.concatenate(dsts[::-1])   # :25-31
    sub2full = np.unique(np.concatenate((full_srcs, full_dsts)))  # :33
    full2sub = np.zeros(np.max(sub2full) + 1, dtype=np.int64)     # :34-35
    full2sub[sub2full] = np.arange(len(sub2full), dtype=np.int64)
    sub_srcs = full2sub[full_srcs]; sub_dsts = full2sub[full_dsts]  # :37-38
    vnum = len(sub2full)
    coo = spsp.coo_matrix((np.ones(len(sub_srcs), dtype=np.uint8), (sub_srcs, sub_dsts)), shape=(vnum, vnum))
    csr = coo.tocsr()                                             # :43
    csr.sort_indices()
    tnid = train_nid.copy()                                       # :48
    valid_t_max = np.max(sub2full); valid_t_min = np.min(tnid)    # :49-50
    tnid = np.where(tnid <= valid_t_max, tnid, valid_t_min)       # :51
    subtrainid = full2sub[np.unique(tnid)]                        # :52
    return csr.indptr.astype(np.int64), csr.indices.astype(np.int64), sub2full, subtrainid


# --------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------
def rmat_edges(seed, scale, first, n, a=0.45, b=0.22, c=0.22):
    q = lambda x: int(round(x * 2 ** 32)) & 0xFFFFFFFF
    src = np.empty(n, dtype=np.int64)
    dst = np.empty(n, dtype=np.int64)
    lib().pgc_rmat_edges(ctypes.c_uint64(seed), scale, q(a), q(b), q(c), first, n, _p(src, _i64p), _p(dst, _i64p))
    return src, dst


def random_features(seed, row0, rows, dim):
    out = np.empty((rows, dim), dtype=np.float32)
    lib().pgc_random_features(ctypes.c_uint64(seed), row0, rows, dim, _p(out, _f32p), dim)
    return out
