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


def dg_raw(partition_num, indptr, indices, vnum, train_nids, hops, threads=None):
    """-> (belongs int8 [V], r_mask uint8 [P, V], p_vnum, r_vnum). threads: host threads sharing the work inside
    each vertex when hops == 2 (default: every CPU the process may use; the result does not depend on it)"""
    lib = L.load()
    if threads is None:
        threads = int(os.environ.get("PG_DG_THREADS", 0)) or default_threads()
    train = np.ascontiguousarray(train_nids, dtype=np.int64)
    belongs = np.empty(vnum, dtype=np.int8)
    r_mask = np.empty((partition_num, vnum), dtype=np.uint8)
    p_vnum = np.zeros(partition_num, dtype=np.int64)
    r_vnum = np.zeros(partition_num, dtype=np.int64)
    vp = ctypes.c_void_p
    L.check(lib.pg_dg_partition_mt(vnum, vp(indptr.ctypes.data), vp(indices.ctypes.data), vp(train.ctypes.data),
                                   len(train), partition_num, hops, vp(belongs.ctypes.data), vp(r_mask.ctypes.data),
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
