"""hash partitioner — counterpart of PaGraph/partition/hash.py:15-70: shuffle the
train ids, cut into equal chunks, closure per chunk. `--seed` replaces the
reference's unseeded np.random.shuffle."""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as spsp


def hash_chunks(train_nid, partitions, seed=0):
    train_nid = np.array(train_nid, dtype=np.int64)
    np.random.default_rng(seed).shuffle(train_nid)
    chunk = int(len(train_nid) / partitions)                     # hash.py:44
    out = []
    for pid in range(partitions):
        lo = chunk * pid
        hi = len(train_nid) if pid == partitions - 1 else lo + chunk   # hash.py:47-50
        out.append(train_nid[lo:hi])
    return out


def main(argv=None):
    from .. import data
    from ..sampling import DeviceGraph
    from .utils import get_sub_graph
    parser = argparse.ArgumentParser(description='Hash')
    parser.add_argument("--dataset", type=str, default=None, help="path to the dataset folder")
    parser.add_argument("--num-hops", type=int, default=1, help="num hops for the extended graph")
    parser.add_argument("--partition", type=int, default=2, help="partition number")
    parser.add_argument("--seed", type=int, default=0)
    args = parser.parse_args(argv)
    adj = spsp.load_npz(os.path.join(args.dataset, 'adj.npz'))
    g = DeviceGraph(adj, readonly=True)
    train_mask, _, _ = data.get_masks(args.dataset)
    train_nid = np.nonzero(train_mask)[0].astype(np.int64)
    labels = data.get_labels(args.dataset)
    for pid, part_nid in enumerate(hash_chunks(train_nid, args.partition, args.seed)):
        subadj, sub2fullid, subtrainid = get_sub_graph(g, part_nid, args.num_hops)
        sublabel = labels[sub2fullid[subtrainid]]
        data.save_partition(args.dataset, args.partition, pid, subadj, sub2fullid, subtrainid, sublabel)


if __name__ == '__main__':
    sys.exit(main())
