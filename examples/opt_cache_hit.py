"""Oracle vs degree-policy cache hit rate for a partition — counterpart of the reference's
examples/opt_cache_hit.py (same flags + --gpu/--cache-ratio/--partitions).
  python examples/opt_cache_hit.py --dataset DIR [--gpu 0] [--cache-ratio 0.2]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='Eval')
    parser.add_argument("--dataset", type=str, default=None, help="path to the dataset folder")
    parser.add_argument("--feat-size", type=int, default=600)
    parser.add_argument("--n-layers", type=int, default=1)
    parser.add_argument("--preprocess", dest='preprocess', action='store_true')
    parser.add_argument("--n-epochs", type=int, default=10)
    parser.add_argument("--batch-size", type=int, default=6000)
    parser.add_argument("--num-neighbors", type=int, default=2)
    parser.add_argument("--gpu", type=int, default=0)
    parser.add_argument("--cache-ratio", type=float, default=0.2)     # opt_cache_hit.py:58
    parser.add_argument("--partition", type=int, default=0)
    parser.add_argument("--partitions", type=int, default=1)
    args = parser.parse_args()
    import pagraph_amd.data as data
    from pagraph_amd import analysis
    from pagraph_amd.sampling import DeviceGraph, NeighborSampler
    torch.cuda.set_device(args.gpu)
    adj, _ = data.get_sub_train_graph(args.dataset, args.partition, args.partitions)
    train_nid = data.get_sub_train_nid(args.dataset, args.partition, args.partitions)
    g = DeviceGraph(adj, readonly=True)
    num_hops = args.n_layers if args.preprocess else args.n_layers + 1
    sampler = NeighborSampler(g, args.batch_size, args.num_neighbors, neighbor_type='in', shuffle=True,
                              num_hops=num_hops, seed_nodes=train_nid, prefetch=True)
    for epoch in range(args.n_epochs):
        freq, loaded = analysis.access_frequency(sampler)
        print('Oracle cache hit rate: ', analysis.optimal_cache_hit(freq, args.cache_ratio))
        print('Degree-policy cache hit rate: ', analysis.degree_cache_hit(freq, g.out_degrees(), args.cache_ratio))
        print('Vertices loaded this epoch: ', loaded)
