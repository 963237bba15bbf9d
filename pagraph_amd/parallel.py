"""Multi-GPU glue: one process per GPU, each owning one dg partition
(examples/profile/pa_gcn.py:27-41,65,154-157). The data path has no collective
— features and samples never cross GPUs — the only exchange is DDP's gradient
all-reduce (RCCL over xGMI) plus the small host-side agreements below."""
import os

import torch
import torch.distributed as dist


def init_process(rank=None, world_size=None, backend='nccl'):
    """pa_gcn.py:18-24, but rendezvous comes from the launcher's env when present"""
    rank = int(os.environ.get("RANK", 0)) if rank is None else rank
    world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else world_size
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29501')
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world_size)
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    torch.manual_seed(rank)
    return rank, world_size


def equalize_steps(local_steps, device=None):
    """dg balances partitions only approximately (dg.py:54-55), so ranks disagree on
    steps/epoch and DDP would hang. Every rank runs MAX steps; a rank
    that runs out of seeds wraps around to its first batches."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(local_steps)
    t = torch.tensor([int(local_steps)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def max_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_tensor(t, src=0):
    """rank `src` computed it (e.g. dg's `belongs`), everyone gets it"""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def wrapped_batches(num_local_batches, steps):
    """batch index per step for a rank with `num_local_batches` when every rank runs `steps`"""
    return [s % max(1, num_local_batches) for s in range(steps)]
