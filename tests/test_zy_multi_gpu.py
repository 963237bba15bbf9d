"""Tests that need TWO GPUs (skipped on the one-GPU boxes): the first multi-GPU box should turn into a parity check, not a
diagnosis. Real RCCL, one rank per GPU:
  * GraphedTrainer's flat-gradient all-reduce — both shapes: eager collective between two graphs, collective captured in the
    step — against DDP's trajectory on the same seeds; replicas bit-identical after 30 steps
    (the asserts of test_zz_bench_lines.py::test_two_rank_graphed_allreduce_matches_ddp, which runs over gloo on one GPU);
  * bench.py --gpus 2 over nccl exactly as the driver launches it, at 300 K vertices: preflight, dg, closures, shared host
    table, equalised steps, all-reduce, per-rank blocks, loss evidence.
reference: /root/reference/examples/profile/pa_gcn.py:18-24 (init_process), :65 (DDP), :154-157 (mp.spawn per GPU)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (one RCCL rank per GPU)")]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rand_csc(rng, V, E):
    import scipy.sparse as spsp
    src, dst = rng.integers(0, V, E), rng.integers(0, V, E)
    keep = src != dst
    m = spsp.coo_matrix((np.ones(int(keep.sum()), np.uint8), (src[keep], dst[keep])), shape=(V, V)).tocsr()
    m.data[:] = 1
    return m


def _rccl_two_gpu_worker(rank, world, port, out_dir):
    """rank r on GPU r over RCCL: DDP eager vs GraphedTrainer with the eager all-reduce vs the in-graph all-reduce"""
    import torch.distributed as dist
    import torch.nn.functional as Fn
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", rank)
    from pagraph_amd.model import GCNSampling
    from pagraph_amd.optim import Adam
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    from pagraph_amd.storage import GraphCacheServer, HostFeatureStore
    from pagraph_amd.trainer import GraphedTrainer, MinibatchTrainer, cycle_batches
    rng = np.random.default_rng(5)
    V, Fdim, C, B = 4000, 32, 4, 250
    adj = _rand_csc(rng, V, 24000)
    g = DeviceGraph(adj, device=dev)
    feats = rng.standard_normal((V, Fdim)).astype(np.float32)
    labels = torch.from_numpy(rng.integers(0, C, V)).to(dev)
    train = np.arange(rank, V, 2 * world, dtype=np.int64)        # disjoint seeds per rank
    res = {}
    for mode in ("ddp", "graph-eager", "graph-ingraph"):
        store = HostFeatureStore({"features": torch.from_numpy(feats)})
        c = GraphCacheServer(store, V, torch.arange(V), rank, miss_mode="zerocopy" if mode == "ddp" else "async")
        c.init_field(["features"])
        c.auto_cache(g, ["features"], cache_ratio=0.5)
        torch.manual_seed(rank)                                   # different init per rank: the broadcast must fix it
        model = GCNSampling(Fdim, 8, C, 1, Fn.relu, 0.0).to(dev)
        need = model.required_inputs(3)
        smp = NeighborSampler(g, B, 2, neighbor_type='in', shuffle=True, num_hops=2, seed_nodes=train, prefetch=True,
                              seed=3 + rank, static=(mode != "ddp"), defer_transpose=(mode != "ddp"))
        if mode == "ddp":
            opt = torch.optim.Adam(model.parameters(), lr=1e-2)
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[rank])
            tr = MinibatchTrainer(net, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=need)
        else:
            opt = Adam(model.parameters(), lr=1e-2)
            tr = GraphedTrainer(model, torch.nn.CrossEntropyLoss(), opt, c, smp, labels, dev, need=need, world_size=world)
            if mode == "graph-eager":
                tr.allreduce_in_graph = False                     # graph A -> eager all-reduce (comm stream) -> graph B
            # graph-ingraph: left at None — the probe (every rank captures a tiny all-reduce, MIN-agreement, two checked
            # replays) must say yes over real RCCL
        out = []
        tr.on_step = lambda step, loss: out.append(loss.detach().clone())
        tr.run_steps(cycle_batches(smp, 30), 30)
        tr.synchronize()
        torch.cuda.synchronize()
        res[mode] = (torch.stack(out).cpu(), [p.detach().cpu().clone() for p in model.parameters()],
                     getattr(tr, "allreduce_in_graph", None))
        if mode != "ddp":
            c.shutdown_miss_queue()
    torch.save(res, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_gpus_rccl_graphed_allreduce_matches_ddp(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_rccl_two_gpu_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt") for i in range(2)]
    assert r[0]["graph-eager"][2] is False and r[1]["graph-eager"][2] is False
    assert r[0]["graph-ingraph"][2] is True and r[1]["graph-ingraph"][2] is True, "the in-graph all-reduce probe said no over RCCL"
    for mode in ("ddp", "graph-eager", "graph-ingraph"):          # replicas stay identical — bit for bit
        for a, b in zip(r[0][mode][1], r[1][mode][1]):
            assert torch.equal(a, b), mode
        assert torch.isfinite(r[0][mode][0]).all()
    for i in range(2):                                            # same trajectory as DDP, both step shapes
        for mode in ("graph-eager", "graph-ingraph"):
            assert torch.allclose(r[i]["ddp"][0], r[i][mode][0], rtol=3e-4, atol=3e-5), (mode, r[i]["ddp"][0], r[i][mode][0])
            for a, b in zip(r[i]["ddp"][1], r[i][mode][1]):
                assert torch.allclose(a, b, rtol=1e-3, atol=1e-4), mode
        # the two step shapes run the same kernels around the same sum: identical trajectories
        assert torch.allclose(r[i]["graph-eager"][0], r[i]["graph-ingraph"][0], rtol=1e-5, atol=1e-6)


def test_bench_two_gpus_over_rccl_as_the_driver_launches_it():
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5",
           "--vertices", "300000", "--edges", "3000000", "--skip-cpu-baseline", "--skip-opt-hit"]
    from conftest import run_group
    r = run_group(cmd, 900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-4000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["dist"]["world_size"] == 2 and line["dist"]["backend"] == "nccl"
    assert line["dist"]["ranks_share_a_gpu"] is False
    pf = line["dist"]["preflight"]
    assert pf["ok"] and pf["replicas_identical"] and pf["allreduce_in_graph"] is True
    assert line["config"]["allreduce_in_graph"] is True and line["config"]["miss_wait"] == "device"
    assert len(line["ranks"]) == 2 and {r_["device"].get("uuid", r_["gpu"]) for r_ in line["ranks"]}.__len__() == 2
    assert line["trained"]["finite"] and line["trained"]["param_update_l2"] > 0
    assert not line["misses_timed_out"] and line["value"] > 0 and line["ms_per_step"] > 0
