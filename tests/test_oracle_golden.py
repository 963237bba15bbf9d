"""CPU: the oracle (oracle/) against the golden vectors produced by the REFERENCE's
own Python (oracle/gen_golden.py) and against published known-answer vectors."""
import glob
import os

import numpy as np
import pytest


# Random123 v1.14 kat_vectors, philox4x32-10
PHILOX_KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,want", PHILOX_KAT)
def test_philox_known_answers(oracle, ctr, key, want):
    assert tuple(oracle.philox4x32_10(ctr, key)) == want


def _state_from_g1(oracle, z):
    st = oracle.CacheState(len(z["nid_map"]), z["nid_map"])
    tables = {"features": z["features_table"], "norm": z["norm_table"]}
    st.cache_fix_data(z["cached_nids"], tables, is_full=False)
    return st, tables


@pytest.mark.parametrize("F", [8, 600, 602])
def test_g1_fetch_data(oracle, golden_dir, F):
    z = np.load(os.path.join(golden_dir, f"g1_fetch_data_F{F}.npz"))
    st, tables = _state_from_g1(oracle, z)
    # G2: state after cache_fix_data
    assert np.array_equal(st.localid2cacheid, z["state_localid2cacheid"])
    assert np.array_equal(st.gpu_flag.astype(bool), z["state_gpu_flag"])
    assert st.cached_num == int(z["state_cached_num"])
    for i in range(int(z["num_layers"])):
        got = st.fetch_layer(z[f"layer{i}_nids"], tables)
        np_got, _ = oracle.fetch_layer_numpy(st, z[f"layer{i}_nids"], tables)
        for name in ("features", "norm"):
            assert got[name].shape == z[f"layer{i}_{name}"].shape
            assert np.array_equal(got[name], z[f"layer{i}_{name}"]), (i, name)   # pure copy: bit exact
            assert np.array_equal(np_got[name], z[f"layer{i}_{name}"])
    assert st.try_num == int(z["try_num"]) and st.miss_num == int(z["miss_num"])
    assert st.get_miss_rate() == float(z["miss_rate"])


@pytest.mark.parametrize("F", [8, 600, 602])
def test_g3_fetch_from_cache(oracle, golden_dir, F):
    z = np.load(os.path.join(golden_dir, f"g3_fetch_from_cache_F{F}.npz"))
    V = len(z["nid_map"])
    st = oracle.CacheState(V, z["nid_map"])
    tables = {"features": z["features_table"], "norm": z["norm_table"]}
    st.cache_fix_data(np.arange(V), tables, is_full=True)
    for i in range(int(z["num_layers"])):
        got = st.fetch_layer(z[f"layer{i}_nids"], tables)
        for name in ("features", "norm"):
            assert np.array_equal(got[name], z[f"layer{i}_{name}"])
    assert st.miss_num == 0


@pytest.mark.parametrize("tag", ["partial", "full"])
def test_g5_auto_cache(oracle, golden_dir, tag):
    z = np.load(os.path.join(golden_dir, f"g5_auto_cache_{tag}.npz"))
    V = len(z["nid_map"])
    total_dim = z["features_table"].shape[1] + 1
    avail = int(z["total_memory"]) - int(z["peak_allocated"]) - int(z["peak_cached"]) - 1024 ** 3
    cap = int(avail / (total_dim * 4))                       # storage.py:81-84
    assert cap == int(z["capability"])
    st = oracle.CacheState(V, z["nid_map"])
    ids, full = st.auto_cache_select(z["out_degrees"], cap)
    st.cache_fix_data(ids, {"features": z["features_table"], "norm": z["norm_table"]}, full)
    assert full == bool(z["full_cached"]) and st.cached_num == int(z["cached_num"])
    assert np.array_equal(st.gpu_flag.astype(bool), z["gpu_flag"])
    assert np.array_equal(st.localid2cacheid, z["localid2cacheid"])
    assert np.array_equal(st.cache["features"], z["cache_features"])
    assert np.array_equal(st.cache["norm"], z["cache_norm"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g4_*.npz"))),
                         ids=os.path.basename)
def test_g4_dg(oracle, path):
    z = np.load(path)
    P, V, hops = int(z["P"]), int(z["V"]), int(z["hops"])
    sub_v, sub_trainv = oracle.dg_partition(P, z["csc_indptr"], z["csc_indices"], V, z["train_nids"], hops)
    for p in range(P):
        assert np.array_equal(sub_trainv[p], z[f"sub_trainv_{p}"])
        assert np.array_equal(sub_v[p], z[f"sub_v_{p}"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g6_*.npz"))),
                         ids=os.path.basename)
def test_g6_closure(oracle, path):
    z = np.load(path)
    ip, ix, sub2full, subtrain = oracle.closure_subgraph(z["csc_indptr"], z["csc_indices"], int(z["V"]),
                                                         z["train_nids"], int(z["hops"]))
    assert np.array_equal(sub2full, z["sub2full"])
    assert np.array_equal(subtrain, z["subtrainid"])
    assert np.array_equal(ip, z["sub_indptr"]) and np.array_equal(ix, z["sub_indices"])


def test_vectorised_philox_matches_the_scalar_one(oracle):
    rng = np.random.default_rng(0)
    c = rng.integers(0, 2 ** 32, (50, 4), dtype=np.uint64)
    k = (0xDEADBEEF, 0x12345678)
    got = oracle.philox4x32_10_np(c[:, 0], c[:, 1], c[:, 2], c[:, 3], *k)
    for i in range(50):
        assert [int(g[i]) for g in got] == oracle.philox4x32_10(c[i], k)


def test_dropout_mask_spec(oracle):
    """keep rate, determinism, and that every (row, column) gets its own 16 bits"""
    keep, scale = oracle.dropout_mask(200, 600, oracle.dropout_threshold(0.5), 7, 1, 5)
    assert keep.shape == (200, 600) and abs(keep.mean() - 0.5) < 0.01 and scale == np.float32(2.0)
    again, _ = oracle.dropout_mask(200, 600, oracle.dropout_threshold(0.5), 7, 1, 5)
    assert np.array_equal(keep, again)
    other, _ = oracle.dropout_mask(200, 600, oracle.dropout_threshold(0.5), 7, 1, 6)
    assert 0.4 < (keep != other).mean() < 0.6
    cols = keep.astype(np.float64) - keep.mean()
    corr = (cols.T @ cols) / 200
    off = corr - np.diag(np.diag(corr))
    assert np.abs(off).max() < 0.12           # no two columns share their bits
    k1, s1 = oracle.dropout_mask(50, 64, oracle.dropout_threshold(0.1), 7, 1, 5)
    assert abs(k1.mean() - 0.9) < 0.03 and abs(float(s1) - 1 / 0.9) < 1e-3


# ---- G7 / G8: the reference's model classes (gcn_nssc.py, graphsage_nssc.py) -----------------------------
def _model_case(z):
    sizes = [int(x) for x in z["layer_sizes"]]
    blocks = [(z[f"blk{b}_indptr"], z[f"blk{b}_src"]) for b in range(len(sizes) - 1)]
    frames = [{k[len(f"layer{i}_"):]: z[k] for k in z.files if k.startswith(f"layer{i}_") and k != "layer_sizes"}
              for i in range(len(sizes))]
    state = {k[len("param:"):]: z[k] for k in z.files if k.startswith("param:")}
    return sizes, blocks, frames, state


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g7_*.npz"))),
                         ids=os.path.basename)
def test_g7_gcn_models_vs_reference(oracle, path):
    """oracle.gcn_model_forward == the reference's GCNSampling / GCNInfer (all four forward variants)"""
    z = np.load(path)
    sizes, blocks, frames, state = _model_case(z)
    got, _ = oracle.gcn_model_forward(blocks, sizes, frames, state, int(z["n_layers"]), bool(z["preprocess"]),
                                      infer=str(z["arch"]) == "gcn_infer")
    want = z["logits"]
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g8_*.npz"))),
                         ids=os.path.basename)
def test_g8_sage_models_vs_reference(oracle, path):
    """oracle.sage_model_forward == the reference's GraphSageSampling ('mean', 'gcn', preprocess)"""
    z = np.load(path)
    sizes, blocks, frames, state = _model_case(z)
    got = oracle.sage_model_forward(blocks, sizes, frames, state, int(z["n_layers"]), str(z["aggregator"]),
                                    bool(z["preprocess"]))
    want = z["logits"]
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_g9_cache_analysis_vs_reference(oracle, golden_dir):
    """count_vertex_freq / optimal_cache_hit (examples/opt_cache_hit.py:22-31) and count_nf_vnum
    (examples/count_vnum.py:16-20) restated in the oracle == the reference's functions on a fixed trace"""
    z = np.load(os.path.join(golden_dir, "g9_cache_analysis.npz"))
    freq = np.zeros(int(z["V"]), dtype=np.int64)
    vnum = 0
    for t in range(int(z["num_nodeflows"])):
        layers = [z[f"nf{t}_layer{i}"] for i in range(3)]
        oracle.count_vertex_freq(layers, freq)
        vnum += oracle.count_nf_vnum(layers)
    assert np.array_equal(freq, z["freq"]) and vnum == int(z["vnum"])
    for r in (0.05, 0.2, 0.5):
        assert oracle.optimal_cache_hit(freq, r) == float(z[f"opt_hit_{int(r * 100):02d}"])
