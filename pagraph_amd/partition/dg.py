"""dg partitioner — counterpart of PaGraph/partition/dg.py (same CLI flags and
output files). The sequential greedy assignment runs in C++ (pg_dg_partition),
the per-partition closure on the GPU (partition/utils.py)."""
import argparse
import ctypes
import os
import sys

import numpy as np
import scipy.sparse as spsp

from .. import _lib as L


def _csc_arrays(adj):
    csc = spsp.csc_matrix(adj)
    csc.sum_duplicates()
    csc.sort_indices()
    return (np.ascontiguousarray(csc.indptr, dtype=np.int64), np.ascontiguousarray(csc.indices, dtype=np.int32),
            csc.shape[0])


def default_threads():
    """host threads for the hops == 2 team: the CPUs this process may use (affinity and cgroup quota), at most 32"""
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cpus = min(cpus, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(32, cpus))


LAST_GPU_STATS = None       # pg_dg_gpu_stats_t of the last device-assisted run (diagnosis / bench.py's record)


def dg_raw(partition_num, indptr, indices, vnum, train_nids, hops, threads=None, device="auto", want_r_mask=True):
    """-> (belongs int8 [V], r_mask uint8 [P, V] (or None), p_vnum, r_vnum).

    device='auto' (default): the neighbour sets are built on the GPU (pg_dg_partition_gpu, round 6: hops 1 and 2, P <= 16)
    when one is there — `indptr` / `indices` may be CUDA tensors (the graph already in HBM) or host arrays (uploaded) — else
    on the host; 'cpu' forces the host code (pg_dg_partition_mt). Both give the same partition, bit for bit.
    threads: host threads of the hops == 2 builder team of the host code (default: every CPU the process may use)."""
    global LAST_GPU_STATS
    import torch
    lib = L.load()
    train = np.ascontiguousarray(train_nids, dtype=np.int64)
    belongs = np.empty(vnum, dtype=np.int8)
    r_mask = np.empty((partition_num, vnum), dtype=np.uint8) if want_r_mask else None
    p_vnum = np.zeros(partition_num, dtype=np.int64)
    r_vnum = np.zeros(partition_num, dtype=np.int64)
    vp = ctypes.c_void_p
    on_dev = torch.is_tensor(indptr) and indptr.is_cuda
    if device != "cpu" and (on_dev or torch.cuda.is_available()) and partition_num <= 16 and hops <= 2 and vnum < (1 << 28) \
            and (len(train) < 2 or bool(np.all(np.diff(train) > 0))):
        ip = indptr if on_dev else torch.as_tensor(np.ascontiguousarray(indptr, dtype=np.int64)).cuda()
        ix = indices if on_dev else torch.as_tensor(np.ascontiguousarray(indices, dtype=np.int32)).cuda()
        ip, ix = ip.to(torch.int64).contiguous(), ix.to(torch.int32).contiguous()
        st = L.PgDgGpuStats()
        vnum2 = np.zeros((2, partition_num), dtype=np.int64)
        with torch.cuda.device(ip.device):
            rc = lib.pg_dg_partition_gpu(vnum, L.ptr(ip), L.ptr(ix), vp(train.ctypes.data), len(train), partition_num, hops,
                                         vp(belongs.ctypes.data), vp(r_mask.ctypes.data) if want_r_mask else None,
                                         vp(vnum2.ctypes.data), ctypes.byref(st), L.stream_ptr())
        if rc == 0:
            p_vnum, r_vnum = vnum2[0].copy(), vnum2[1].copy()
            LAST_GPU_STATS = {n: getattr(st, n) for n, _ in st._fields_}
            return belongs, r_mask, p_vnum, r_vnum
        if rc != -4:                       # anything but PG_ERR_UNSUPPORTED is an error, not a reason to fall back
            L.check(rc, "pg_dg_partition_gpu")
    LAST_GPU_STATS = None
    if threads is None:
        threads = int(os.environ.get("PG_DG_THREADS", 0)) or default_threads()
    if torch.is_tensor(indptr):
        indptr, indices = indptr.cpu().numpy(), indices.cpu().numpy()
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    L.check(lib.pg_dg_partition_mt(vnum, vp(indptr.ctypes.data), vp(indices.ctypes.data), vp(train.ctypes.data),
                                   len(train), partition_num, hops, vp(belongs.ctypes.data),
                                   vp(r_mask.ctypes.data) if want_r_mask else None,
                                   vp(p_vnum.ctypes.data), vp(r_vnum.ctypes.data), int(threads)), "pg_dg_partition_mt")
    return belongs, r_mask, p_vnum, r_vnum


def dg(partition_num, adj, train_nids, hops):
    """dg.py:59-103 — returns (sub_v, sub_trainv): per partition the vertex set with
    redundancy and the assigned train vertices (both ascending)."""
    indptr, indices, vnum = _csc_arrays(adj)
    print('total vertices: {} | train vertices: {}'.format(vnum, len(train_nids)))
    belongs, r_mask, p_vnum, r_vnum = dg_raw(partition_num, indptr, indices, vnum, train_nids, hops)
    sub_v, sub_trainv = [], []
    for pid in range(partition_num):
        sub_trainv.append(np.where(belongs == pid)[0])
        p_v = np.where(r_mask[pid] != 0)[0]
        sub_v.append(p_v)
        assert p_v.shape[0] == r_vnum[pid]
        print('vertex# with self-reliance: ', r_vnum[pid])
        print('vertex# w/o  self-reliance: ', p_vnum[pid])
    return sub_v, sub_trainv


def main(argv=None):
    from .. import data
    from ..sampling import DeviceGraph
    from .utils import get_sub_graph
    parser = argparse.ArgumentParser(description='Partition')
    parser.add_argument("--dataset", type=str, default=None, help="dataset dir")
    parser.add_argument("--partition", type=int, default=2, help="num of partitions")
    parser.add_argument("--num-hops", type=int, default=1, help="num of hop neighbors required for a batch")
    args = parser.parse_args(argv)
    adj = spsp.load_npz(os.path.join(args.dataset, 'adj.npz'))
    train_mask, _, _ = data.get_masks(args.dataset)
    train_nids = np.nonzero(train_mask)[0].astype(np.int64)
    labels = data.get_labels(args.dataset)
    p_v, p_trainv = dg(args.partition, adj, train_nids, args.num_hops)
    g = DeviceGraph(adj, readonly=True)
    for pid, (pv, ptrainv) in enumerate(zip(p_v, p_trainv)):
        print('generating subgraph# {}...'.format(pid))
        subadj, sub2fullid, subtrainid = get_sub_graph(g, ptrainv, args.num_hops)
        sublabel = labels[sub2fullid[subtrainid]]
        data.save_partition(args.dataset, args.partition, pid, subadj, sub2fullid, subtrainid, sublabel)


if __name__ == '__main__':
    sys.exit(main())
