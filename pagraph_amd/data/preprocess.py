"""Dataset preparation CLI — counterpart of PaGraph/data/preprocess.py:116-183 (same flags):
edge list -> adj.npz, random features / labels / 65-10-25 split. `--gen-rmat V E` additionally
replaces the external PaRMAT step (README.md:36-41) with the seeded GPU RMAT generator."""
import argparse
import os
import sys

import numpy as np
import scipy.sparse


def pp2adj(filepath, is_direct=True, delimiter='\t', outfile=None):
    """preprocess.py:10-46"""
    pp = np.loadtxt(filepath, delimiter=delimiter)
    src_node = pp[:, 0].astype(np.int64)
    dst_node = pp[:, 1].astype(np.int64)
    min_nid = min(np.min(src_node), np.min(dst_node))
    vnum = max(np.max(src_node), np.max(dst_node)) - min_nid + 1
    src_node -= min_nid
    dst_node -= min_nid
    if not is_direct:
        src_node, dst_node = np.concatenate((src_node, dst_node)), np.concatenate((dst_node, src_node))
    coo_adj = scipy.sparse.coo_matrix((np.ones(len(src_node), dtype=np.int64), (src_node, dst_node)), shape=(vnum, vnum))
    if outfile is not None:
        scipy.sparse.save_npz(outfile, coo_adj)
    return coo_adj


def main(argv=None):
    parser = argparse.ArgumentParser(description='Preprocess')
    parser.add_argument("--dataset", type=str, default=None, help="dataset dir")
    parser.add_argument("--ppfile", type=str, default=None, help='point-to-point graph filename')
    parser.add_argument("--directed", dest="directed", action='store_true')
    parser.add_argument("--gen-rmat", type=int, nargs=2, metavar=('V', 'E'), default=None,
                        help="generate a symmetric RMAT graph with V vertices and E undirected edges on the GPU")
    parser.add_argument("--gen-feature", dest='gen_feature', action='store_true')
    parser.add_argument("--feat-size", type=int, default=600)
    parser.add_argument("--gen-label", dest='gen_label', action='store_true')
    parser.add_argument("--class-num", type=int, default=60)
    parser.add_argument("--gen-set", dest='gen_set', action='store_true')
    args = parser.parse_args(argv)
    if not os.path.exists(args.dataset):
        print('{}: No such a dataset folder'.format(args.dataset))
        sys.exit(-1)
    adj_file = os.path.join(args.dataset, 'adj.npz')
    if args.gen_rmat is not None:
        from . import synthetic as syn
        V, E = args.gen_rmat
        ip, ix = syn.rmat_graph(V, E)
        csc = scipy.sparse.csc_matrix((np.ones(ix.numel(), np.int8), ix.cpu().numpy(), ip.cpu().numpy()), shape=(V, V))
        scipy.sparse.save_npz(adj_file, csc.tocoo())
        vnum = V
    elif args.ppfile is not None:
        print('Generating adj matrix in: {}...'.format(adj_file))
        vnum = pp2adj(os.path.join(args.dataset, args.ppfile), is_direct=args.directed, outfile=adj_file).shape[0]
    else:
        vnum = scipy.sparse.load_npz(adj_file).shape[0]
    if args.gen_feature:
        from . import synthetic as syn
        import torch
        feat = torch.empty((vnum, args.feat_size), dtype=torch.float32)
        syn.fill_random_features(feat)
        np.save(os.path.join(args.dataset, 'feat.npy'), feat.numpy())
    if args.gen_label:
        from . import synthetic as syn
        np.save(os.path.join(args.dataset, 'labels.npy'), syn.random_labels(vnum, args.class_num).numpy())
    if args.gen_set:
        from . import synthetic as syn
        tr, va, te = syn.split_dataset(vnum)
        np.save(os.path.join(args.dataset, 'train.npy'), tr.numpy())
        np.save(os.path.join(args.dataset, 'val.npy'), va.numpy())
        np.save(os.path.join(args.dataset, 'test.npy'), te.numpy())
    print('Done.')


if __name__ == '__main__':
    main()
