"""helpers shared by the GPU test files (importable from spawned workers: tests/ is on sys.path via conftest)"""
import os

import numpy as np
import scipy.sparse as spsp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    """a TCP port nobody listens on right now (a fixed rendezvous port can sit in TIME_WAIT after the previous run)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rand_csc(rng, V, E, powerlaw=True):
    if powerlaw:
        w = 1.0 / np.arange(1, V + 1) ** 0.9; w /= w.sum()
        s = rng.choice(V, E, p=w); d = rng.choice(V, E, p=w)
    else:
        s = rng.integers(0, V, E); d = rng.integers(0, V, E)
    a = spsp.coo_matrix((np.ones(2 * E, np.int8), (np.concatenate([s, d]), np.concatenate([d, s]))), shape=(V, V)).tocsr()
    a.data[:] = 1
    return a
