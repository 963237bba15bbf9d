"""Test accuracy of saved checkpoints — counterpart of the reference's examples/eval.py:13-46 (same flags):
ONE full-neighbour NodeFlow over the test vertices (expand_factor = V, eval.py:20-26), every layer's features
(and `norm`) loaded, GCNInfer (sum aggregation scaled by norm, gcn_nssc.py:103-164) or GraphSageSampling evaluated
once per checkpoint `<ckpt>/<arch>_<epoch>` (written by examples/profile/pa_gcn.py --ckpt DIR).
  python examples/eval.py --gpu 0 --dataset DIR --arch gcn-nssc --ckpt DIR --start 0 --end 10 --interval 1"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def gnneval(args, infer_model, g, store, labels, test_nid, dev, out=print):
    from pagraph_amd.sampling import full_neighbor_nodeflow
    num_hops = args.n_layers if args.preprocess else args.n_layers + 1
    nf = full_neighbor_nodeflow(g, test_nid, num_hops)                  # eval.py:20-26
    ids = nf._node_mapping.tousertensor().cpu()
    accs = {}
    for ckpt in range(args.start, args.end, args.interval):
        path = os.path.join(args.ckpt, args.arch + '_' + str(ckpt))
        if not os.path.exists(path):
            continue
        state = torch.load(path, map_location='cpu')
        params = list(state.values()) if isinstance(state, dict) else [p.data for p in state.parameters()]
        for infer_param, param in zip(infer_model.parameters(), params):    # eval.py:33-34: positional copy
            infer_param.data.copy_(param)
        infer_model.to(dev).eval()
        with torch.no_grad():
            for i in range(nf.num_layers):                               # nf.copy_from_parent (eval.py:40)
                o0, o1 = nf._layer_offsets[i], nf._layer_offsets[i + 1]
                nf._node_frames[i] = {k: t[ids[o0:o1]].to(dev) for k, t in store.ndata.items()}
            pred = infer_model(nf)
            batch_labels = labels[nf.layer_parent_nid(-1).cpu()].to(dev)
            num_acc = (pred.argmax(dim=1) == batch_labels).sum().cpu().item()
        accs[ckpt] = num_acc / len(test_nid)
        out("[{}]: Test Accuracy {:.4f}".format(ckpt, accs[ckpt]))
    return accs


def main(args):
    import pagraph_amd.data as data
    from pagraph_amd import server
    from pagraph_amd.model import GCNInfer, GraphSageSampling
    from pagraph_amd.sampling import DeviceGraph
    torch.cuda.set_device(args.gpu or 0)
    dev = torch.device('cuda', args.gpu or 0)
    labels = torch.LongTensor(data.get_labels(args.dataset))
    n_classes = len(np.unique(labels.numpy()))
    _, _, test_mask = data.get_masks(args.dataset)
    test_nid = np.nonzero(test_mask)[0].astype(np.int64)
    g = DeviceGraph(data.get_struct(args.dataset), readonly=True, device=dev)
    if args.arch == 'gcn-nssc':
        store = server.load_store(args.dataset, 'gcn', args.preprocess, pin=False)
        infer_model = GCNInfer(args.feat_size, 32, n_classes, args.n_layers, F.relu, args.preprocess)
    elif args.arch == 'gs-nssc':
        store = server.load_store(args.dataset, 'graphsage', args.preprocess, pin=False)
        infer_model = GraphSageSampling(args.feat_size, 16, n_classes, args.n_layers, F.relu, 0, 'mean', args.preprocess)
    else:
        print('Unknown arch')
        sys.exit(-1)
    return gnneval(args, infer_model, g, store, labels, test_nid, dev)


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='GCNInfer')
    parser.add_argument("--gpu", type=int, default=None, help="gpu id. such as 0 or 1 or 2")
    parser.add_argument("--dataset", type=str, default=None, help="path to the dataset folder")
    parser.add_argument("--arch", type=str, default='gcn-nssc', help='model arch')
    parser.add_argument("--feat-size", type=int, default=602, help='input feature size')
    parser.add_argument("--n-layers", type=int, default=1, help="number of hidden gcn layers")
    parser.add_argument("--start", type=int, default=0, help="eval epoch start")
    parser.add_argument("--interval", type=int, default=5, help="eval epoch interval")
    parser.add_argument("--end", type=int, default=60, help='eval epoch end (not include)')
    parser.add_argument("--ckpt", type=str, default='checkpoint', help="checkpoint dir")
    parser.add_argument("--preprocess", dest='preprocess', action='store_true')
    parser.set_defaults(preprocess=False)
    main(parser.parse_args())
