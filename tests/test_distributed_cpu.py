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
