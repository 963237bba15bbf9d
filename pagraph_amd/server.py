"""Feature provider — what server/pa_server.py:15-61 publishes into the DGL shared-memory
store, built in-process: `features` (feat.npy or U[0,1) fallback) and, for GCN,
`norm = 1/in_degree` (pa_server.py:43, inf for isolated vertices exactly as there) plus the
optional one-hop preprocessing X' = norm * (A^T X) (pa_server.py:45-52)."""
import numpy as np
import scipy.sparse as spsp
import torch

from . import data
from .storage import HostFeatureStore


def load_store(dataset, model='gcn', preprocess=False, pin=True):
    coo_adj, feat = data.get_graph_data(dataset)
    features = torch.as_tensor(np.asarray(feat), dtype=torch.float32)
    fields = {}
    if model == 'gcn':
        csc = spsp.csc_matrix(coo_adj)
        csc.sum_duplicates()
        in_deg = torch.from_numpy(np.diff(csc.indptr).astype(np.float32))
        norm = (1. / in_deg).unsqueeze(1)
        if preprocess:
            print('Preprocessing features...')
            # update_all(copy_src, sum) then * norm: row v = norm[v] * sum_{u->v} X[u]
            ones = spsp.csc_matrix((np.ones(csc.nnz, np.float32), csc.indices, csc.indptr), shape=csc.shape)
            features = torch.from_numpy(np.asarray(ones.T @ features.numpy(), dtype=np.float32)) * norm
        fields['norm'] = norm
        fields['features'] = features
    elif model == 'graphsage':
        if preprocess:
            print('preprocessing: warning: jusy copy')
            fields['neigh'] = features
        fields['features'] = features
    else:
        raise ValueError(model)
    return HostFeatureStore(fields, pin=pin)
