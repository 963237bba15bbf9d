"""CPU: structural invariants of the sampler spec on the oracle —
the properties DGL's NeighborSampler guarantees at the reference's call site
(examples/profile/pa_gcn.py:71-76): sampled sources are in-neighbours, min(k,deg)
distinct picks per destination, per-layer uniqueness, uniform selection."""
import numpy as np
import scipy.sparse as spsp
from hypothesis import given, settings, strategies as st


def _graph(rng, V, E):
    s = rng.integers(0, V, E); d = rng.integers(0, V, E)
    a = spsp.coo_matrix((np.ones(E, np.int8), (s, d)), shape=(V, V)).tocsc()
    a.sum_duplicates(); a.sort_indices()
    return a.indptr.astype(np.int64), a.indices.astype(np.int32)


@settings(max_examples=40, deadline=None)
@given(st.integers(20, 300), st.integers(1, 12), st.integers(1, 3), st.integers(0, 2 ** 31), st.integers(0, 5))
def test_nodeflow_invariants(V, k, hops, seed, epoch):
    from oracle import oracle
    rng = np.random.default_rng(seed % 1000)
    indptr, indices = _graph(rng, V, V * 6)
    seeds = rng.permutation(V)[:max(1, V // 4)].astype(np.int64)
    nf = oracle.sample_nodeflow(indptr, indices, seeds, k, hops, seed, epoch, 3)
    offs, nm = nf["layer_offsets"], nf["node_mapping"]
    assert offs[0] == 0 and offs[hops + 1] == len(nm)
    assert np.array_equal(nm[offs[hops]:offs[hops + 1]], seeds)            # rule 2: seed layer in seed order
    for l in range(hops):
        lay = nm[offs[l]:offs[l + 1]]
        assert np.all(np.diff(lay) > 0)                                    # rules 4,5: unique + ascending
        ip, src = nf["blocks"][l]
        dst = nm[offs[l + 1]:offs[l + 2]]
        assert len(ip) == len(dst) + 1 and ip[-1] == len(src)
        used = np.zeros(len(lay), bool)
        for p, v in enumerate(dst):
            nb = indices[indptr[v]:indptr[v + 1]]
            picked = lay[src[ip[p]:ip[p + 1]]]
            assert len(picked) == min(k, len(nb))                          # rule 3
            assert len(set(picked.tolist())) == len(picked)                # without replacement
            assert set(picked.tolist()) <= set(nb.tolist())                # every block edge exists in the CSC
            used[src[ip[p]:ip[p + 1]]] = True
        assert used.all()                                                  # layer = exactly the union of picks


def test_wide_fanout_invariants():
    """fan-out above 64 (the reference accepts any --num-neighbors, pa_gcn.py:146-147): the same rules on a dense graph
    where most vertices have more than k in-neighbours, and a uniformity check of Floyd's subset at k = 70 of 90"""
    from oracle import oracle
    rng = np.random.default_rng(5)
    V = 400
    indptr, indices = _graph(rng, V, V * 150)
    seeds = rng.permutation(V)[:60].astype(np.int64)
    for k in (65, 100, 130):
        nf = oracle.sample_nodeflow(indptr, indices, seeds, k, 1, 99, 0, 1)
        offs, nm = nf["layer_offsets"], nf["node_mapping"]
        lay = nm[offs[0]:offs[1]]
        ip, src = nf["blocks"][0]
        some_sampled = False
        for p, v in enumerate(seeds):
            nb = indices[indptr[v]:indptr[v + 1]]
            picked = lay[src[ip[p]:ip[p + 1]]]
            assert len(picked) == min(k, len(nb)) and len(set(picked.tolist())) == len(picked)
            assert set(picked.tolist()) <= set(nb.tolist())
            some_sampled |= len(nb) > k
        assert some_sampled
    deg, k, trials = 90, 70, 1500
    indptr1 = np.array([0, deg] + [deg] * deg, dtype=np.int64)
    indices1 = np.arange(1, deg + 1, dtype=np.int32)
    counts = np.zeros(deg + 1)
    for t in range(trials):
        nf = oracle.sample_nodeflow(indptr1, indices1, np.array([0]), k, 1, 4321, t, 0)
        counts[nf["node_mapping"][:nf["layer_offsets"][1]]] += 1
    exp = trials * k / deg
    var = trials * (k / deg) * (1 - k / deg)
    chi2 = ((counts[1:] - exp) ** 2 / var).sum() * (deg - 1) / deg
    assert chi2 < 140.0, chi2           # chi2(89 dof) 0.999 quantile ~ 135


def test_uniform_selection_chi2():
    """one vertex with 10 in-neighbours, k = 3: each neighbour picked with p = 0.3"""
    from oracle import oracle
    deg, k, trials = 10, 3, 6000
    indptr = np.array([0, deg] + [deg] * deg, dtype=np.int64)
    indices = np.arange(1, deg + 1, dtype=np.int32)
    counts = np.zeros(deg + 1)
    pairs = {}
    for t in range(trials):
        nf = oracle.sample_nodeflow(indptr, indices, np.array([0]), k, 1, 1234, t, 0)
        picks = nf["node_mapping"][:nf["layer_offsets"][1]]
        counts[picks] += 1
        key = tuple(picks.tolist())
        pairs[key] = pairs.get(key, 0) + 1
    exp = trials * k / deg
    chi2 = ((counts[1:] - exp) ** 2 / exp).sum()
    assert chi2 < 27.9, chi2            # chi2(9 dof) 0.999 quantile
    assert len(pairs) == 120            # all C(10,3) subsets occur
    e2 = trials / 120
    chi2s = sum((c - e2) ** 2 / e2 for c in pairs.values())
    assert chi2s < 175.0, chi2s         # chi2(119 dof) 0.999 quantile ~ 173


def test_different_keys_differ():
    from oracle import oracle
    rng = np.random.default_rng(0)
    indptr, indices = _graph(rng, 500, 20000)
    seeds = np.arange(100, dtype=np.int64)
    a = oracle.sample_nodeflow(indptr, indices, seeds, 2, 2, 1, 0, 0)["node_mapping"]
    same = oracle.sample_nodeflow(indptr, indices, seeds, 2, 2, 1, 0, 0)["node_mapping"]
    assert np.array_equal(a, same)
    for key in ((2, 0, 0), (1, 1, 0), (1, 0, 1)):
        b = oracle.sample_nodeflow(indptr, indices, seeds, 2, 2, *key)["node_mapping"]
        assert not np.array_equal(a, b)
