"""Vertices loaded per epoch — counterpart of the reference's examples/count_vnum.py:16-47 (same flags + --gpu):
the number of NodeFlow rows (all layers, count_nf_vnum :16-20) one epoch of neighbour sampling over the train
vertices references, i.e. the feature rows a cache-less trainer would load.
  python examples/count_vnum.py --dataset DIR [--gpu 0]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def count_nf_vnum(nf):
    """count_vnum.py:16-20 (padding ids of fixed-shape NodeFlows excluded)"""
    vnum = 0
    for lid in range(nf.num_layers):
        ids = nf.layer_parent_nid(lid)
        vnum += int((ids >= 0).sum()) if getattr(nf, 'padded', False) else ids.size(0)
    return vnum


def main(args, out=print):
    import pagraph_amd.data as data
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    torch.cuda.set_device(args.gpu)
    g = DeviceGraph(data.get_struct(args.dataset), readonly=True)
    train_mask, _, _ = data.get_masks(args.dataset)
    train_nid = np.nonzero(train_mask)[0].astype(np.int64)
    num_hops = args.n_layers if args.preprocess else args.n_layers + 1
    sampler = NeighborSampler(g, args.batch_size, args.num_neighbors, neighbor_type='in', shuffle=True, num_workers=16,
                              num_hops=num_hops, seed_nodes=train_nid, prefetch=False)
    totals = []
    for epoch in range(args.n_epochs):
        epoch_load_vnum = 0
        for nf in sampler:
            epoch_load_vnum += count_nf_vnum(nf)
        out('Epoch loaded vertex#: ', epoch_load_vnum)
        totals.append(epoch_load_vnum)
    return totals


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='Eval')
    parser.add_argument("--dataset", type=str, default=None, help="path to the dataset folder")
    parser.add_argument("--feat-size", type=int, default=600, help='input feature size')
    parser.add_argument("--n-layers", type=int, default=1, help="number of hidden gcn layers")
    parser.add_argument("--preprocess", dest='preprocess', action='store_true')
    parser.set_defaults(preprocess=False)
    parser.add_argument("--n-epochs", type=int, default=10, help="number of training epochs")
    parser.add_argument("--batch-size", type=int, default=6000, help="batch size")
    parser.add_argument("--num-neighbors", type=int, default=2, help="number of neighbors to be sampled")
    parser.add_argument("--gpu", type=int, default=0)
    main(parser.parse_args())
