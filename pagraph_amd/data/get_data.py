"""Dataset / partition file readers — same function names, arguments and on-disk
layout as PaGraph/data/get_data.py:8-103 (README.md:18-26, dg.py:156-171):

  <dataset>/adj.npz            scipy sparse, (V,V), row = src, col = dst
  <dataset>/feat.npy           optional fp32 [V, F]; else U[0,1) with F = 600 (get_data.py:24-27)
  <dataset>/labels.npy, train.npy, val.npy, test.npy
  <dataset>/<P>naive/subadj_<i>.npz, sub_trainid_<i>.npy, sub_train2fullid_<i>.npy, sub_label_<i>.npy
"""
import os

import numpy as np
import scipy.sparse


def get_graph_data(dataname, mmap=False):
    """get_data.py:8-27. mmap=True maps feat.npy instead of reading it (the feature provider copies it straight
    into its pinned / shared table: no second full-size copy in host memory)"""
    adj = scipy.sparse.load_npz(os.path.join(dataname, 'adj.npz'))
    try:
        feat = np.load(os.path.join(dataname, 'feat.npy'), mmap_mode='r' if mmap else None)
    except FileNotFoundError:
        print('random generate feat...')
        import torch
        feat = torch.rand((adj.shape[0], 600))
    return adj, feat


def get_sub_train_graph(dataname, idx, partitions):
    dataname = os.path.join(dataname, '{}naive'.format(partitions))
    adj = scipy.sparse.load_npz(os.path.join(dataname, 'subadj_{}.npz'.format(idx)))
    train2fullid = np.load(os.path.join(dataname, 'sub_train2fullid_{}.npy'.format(idx)))
    return adj, train2fullid


def get_struct(dataname):
    return scipy.sparse.load_npz(os.path.join(dataname, 'adj.npz'))


def get_masks(dataname):
    return (np.load(os.path.join(dataname, 'train.npy')),
            np.load(os.path.join(dataname, 'val.npy')),
            np.load(os.path.join(dataname, 'test.npy')))


def get_sub_train_nid(dataname, idx, partitions):
    dataname = os.path.join(dataname, '{}naive'.format(partitions))
    return np.load(os.path.join(dataname, 'sub_trainid_{}.npy'.format(idx)))


def get_labels(dataname):
    return np.load(os.path.join(dataname, 'labels.npy'))


def get_sub_train_labels(dataname, idx, partitions):
    dataname = os.path.join(dataname, '{}naive'.format(partitions))
    return np.load(os.path.join(dataname, 'sub_label_{}.npy'.format(idx)))


def save_partition(dataname, partitions, idx, subadj, sub2fullid, subtrainid, sublabel):
    """writer for the layout above (dg.py:144-171 / hash.py:38-70)"""
    pdir = os.path.join(dataname, '{}naive'.format(partitions))
    os.makedirs(pdir, exist_ok=True)
    scipy.sparse.save_npz(os.path.join(pdir, 'subadj_{}.npz'.format(idx)), subadj)
    np.save(os.path.join(pdir, 'sub_trainid_{}.npy'.format(idx)), subtrainid)
    np.save(os.path.join(pdir, 'sub_train2fullid_{}.npy'.format(idx)), sub2fullid)
    np.save(os.path.join(pdir, 'sub_label_{}.npy'.format(idx)), sublabel)
