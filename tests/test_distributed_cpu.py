"""CPU, world_size 2 over gloo: the N>1 host path — partition ownership, step
equalisation, timing reduction and the gradient all-reduce (DDP) — without a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank); os.environ["WORLD_SIZE"] = str(world)
    from pagraph_amd import parallel
    from pagraph_amd.partition.dg import dg_raw
    parallel.init_process(rank, world, backend="gloo")
    # rank 0 partitions (sequential C++), everyone receives `belongs`
    V = 400
    rng = np.random.default_rng(5)
    s = rng.integers(0, V, 2000); d = rng.integers(0, V, 2000)
    import scipy.sparse as spsp
    csc = spsp.coo_matrix((np.ones(4000, np.int8), (np.concatenate([s, d]), np.concatenate([d, s]))), shape=(V, V)).tocsc()
    csc.sum_duplicates(); csc.sort_indices()
    train = np.arange(0, V, 2, dtype=np.int64)
    belongs = torch.full((V,), -7, dtype=torch.int8)
    if rank == 0:
        b, _, _, _ = dg_raw(world, csc.indptr.astype(np.int64), csc.indices.astype(np.int32), V, train, 1)
        belongs = torch.from_numpy(b.copy())
    parallel.broadcast_tensor(belongs, src=0)
    mine = torch.nonzero(belongs == rank).squeeze(1)
    # unequal partitions -> every rank runs MAX steps, wrapping around
    local_steps = (mine.numel() + 31) // 32 + (3 if rank == 1 else 0)
    steps = parallel.equalize_steps(local_steps)
    # DDP gradient all-reduce on a tiny dense model (the collective of pa_gcn.py:65,96)
    torch.manual_seed(0)
    model = torch.nn.Linear(8, 3)
    ddp = torch.nn.parallel.DistributedDataParallel(model)
    x = torch.full((4, 8), float(rank + 1))
    ddp(x).sum().backward()
    g = model.weight.grad.clone()
    tmax = parallel.max_over_ranks(1.0 + rank)
    tsum = parallel.sum_over_ranks(10.0 * (rank + 1))
    torch.save({"mine": mine, "steps": steps, "local": local_steps, "grad": g, "tmax": tmax, "tsum": tsum,
                "belongs": belongs}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_partition_steps_and_allreduce(tmp_path, hiplib):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt") for i in range(world)]
    assert torch.equal(r[0]["belongs"], r[1]["belongs"])
    allv = torch.cat([r[0]["mine"], r[1]["mine"]])
    assert allv.numel() == 200 and allv.unique().numel() == 200          # disjoint cover of the train set
    assert r[0]["steps"] == r[1]["steps"] == max(r[0]["local"], r[1]["local"])
    assert torch.allclose(r[0]["grad"], r[1]["grad"])                     # averaged over ranks
    assert torch.allclose(r[0]["grad"], torch.full((3, 8), 4 * 1.5))      # mean of 4*1 and 4*2
    assert r[0]["tmax"] == r[1]["tmax"] == 2.0 and r[0]["tsum"] == 30.0


def _store_worker(rank, world, port, ds, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"] = str(rank); os.environ["WORLD_SIZE"] = str(world); os.environ["LOCAL_RANK"] = str(rank)
    from pagraph_amd import data, parallel, server
    parallel.init_process(rank, world, backend="gloo")
    if rank != 0:
        # only local rank 0 may touch the dataset files: the others attach to its shared-memory tables
        def boom(*a, **k):
            raise AssertionError("rank %d loaded the dataset itself" % rank)
        data.get_graph_data = boom
        server.data.get_graph_data = boom
    store = server.load_store(ds, 'graphsage', preprocess=True, pin=False)
    f, n = store.ndata['features'], store.ndata['neigh']
    same_alloc = f.data_ptr() == n.data_ptr()
    dist.barrier()
    if rank == 0:
        f[3, 2] = 1234.5                    # one allocation per node: the write is seen by the other rank
    dist.barrier()
    torch.save({"same_alloc": same_alloc, "probe": float(f[3, 2]), "sum": float(f.double().sum()), "shape": tuple(f.shape),
                "leftovers": [p for p in os.listdir("/dev/shm") if p.startswith(f"pagraph_store_{port}_")]},
               os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_feature_store_is_one_shared_copy_per_node(tmp_path, hiplib):
    """every rank used to np.load + pin the whole table. Now local rank 0 publishes it in shared memory
    (the reference's graph-store server, pa_server.py:33-54) and the others attach; 'features' and 'neigh' of
    GraphSAGE --preprocess are one allocation; nothing is left in /dev/shm."""
    import scipy.sparse as spsp
    ds = tmp_path / "ds"
    ds.mkdir()
    rng = np.random.default_rng(2)
    V = 300
    spsp.save_npz(ds / "adj.npz", spsp.random(V, V, 0.02, format="coo", dtype=np.float32, random_state=1))
    feat = rng.random((V, 20), dtype=np.float32)
    np.save(ds / "feat.npy", feat)
    out = tmp_path / "out"
    out.mkdir()
    mp.spawn(_store_worker, args=(2, _free_port(), str(ds), str(out)), nprocs=2, join=True)
    r = [torch.load(out / f"s{i}.pt") for i in range(2)]
    want = feat.astype(np.float64).sum() - feat[3, 2] + 1234.5
    for x in r:
        assert x["same_alloc"] and x["shape"] == (V, 20) and x["leftovers"] == []
        assert x["probe"] == 1234.5 and abs(x["sum"] - want) < 1e-6 * abs(want)
