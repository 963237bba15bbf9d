"""The DGL builtins the reference's models pass to block_compute
(`import dgl.function as fn`: gcn_nssc.py:72-73,140-141; graphsage_nssc.py:99-110)."""


class copy_src:
    def __init__(self, src, out):
        self.src, self.out = src, out


class _Reduce:
    op = None

    def __init__(self, msg, out):
        self.msg, self.out = msg, out


class mean(_Reduce):
    op = "mean"


class sum(_Reduce):  # noqa: A001  (name mirrors dgl.function.sum)
    op = "sum"


class max(_Reduce):  # noqa: A001  (name mirrors dgl.function.max; graphsage_nssc.py:108, the 'pool' aggregator)
    op = "max"
