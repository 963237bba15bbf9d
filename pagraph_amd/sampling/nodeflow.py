"""Device-resident graph + NodeFlow: the slice of DGL 0.4.1's object protocol
that the reference's hot path touches:

  storage.py:171-173,202,208-216   nf._node_mapping.tousertensor(), nf._layer_offsets,
                                   nf.num_layers, nf._node_frames[i], nf.layer_parent_nid(i)
  gcn_nssc.py:64-76                nf.layers[i].data[...], nf.block_compute(i, msg, red, apply)
  pa_gcn.py:36,89; storage.py:100  DGLGraph(adj, readonly=True), g.out_degrees()
"""
import numpy as np
import scipy.sparse as spsp
import torch

from .. import _lib as L
from ..ops import RowSource, block_aggregate


class DeviceGraph:
    """Counterpart of `DGLGraph(adj, readonly=True)` (examples/profile/pa_gcn.py:36):
    `adj` is scipy sparse with row = src, col = dst (README.md:20), so the
    in-neighbours of v are column v — the CSC arrays, kept in HBM:
      indptr  int64 [V+1]    indices int32 [nnz] (ascending inside a column)."""

    def __init__(self, adj, readonly=True, device=None):
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        csc = spsp.csc_matrix(adj)
        csc.sum_duplicates()
        csc.sort_indices()
        self.num_nodes = csc.shape[0]
        if self.num_nodes >= 2 ** 31:
            raise L.PgError("a partition must have < 2^31 vertices")
        self.indptr_h = torch.from_numpy(csc.indptr.astype(np.int64))
        self.indices_h = torch.from_numpy(csc.indices.astype(np.int32))
        self.indptr = self.indptr_h.to(self.device)
        self.indices = self.indices_h.to(self.device)
        self._out_deg = None

    @classmethod
    def from_csc(cls, indptr, indices, num_nodes, device=None):
        """adopt CSC tensors that already live on the device (bench.py builds them there)"""
        g = cls.__new__(cls)
        g.device = indptr.device if device is None else torch.device(device)
        g.num_nodes = int(num_nodes)
        g.indptr = indptr.to(g.device, torch.int64).contiguous()
        g.indices = indices.to(g.device, torch.int32).contiguous()
        g.indptr_h = g.indices_h = None
        g._out_deg = None
        return g

    def number_of_nodes(self):
        return self.num_nodes

    def number_of_edges(self):
        return int(self.indices.numel())

    def in_degrees(self):
        return self.indptr[1:] - self.indptr[:-1]

    def out_degrees(self):
        """out-degree = row count of adj = occurrences as a source (storage.py:100)"""
        if self._out_deg is None:
            self._out_deg = torch.bincount(self.indices.long(), minlength=self.num_nodes)
        return self._out_deg


class _UserTensor:
    def __init__(self, t):
        self._t = t

    def tousertensor(self):
        return self._t


class _LayerView:
    def __init__(self, frames, i):
        self._frames, self._i = frames, i

    @property
    def data(self):
        if self._frames[self._i] is None:
            self._frames[self._i] = {}
        return self._frames[self._i]


class _Layers:
    def __init__(self, nf):
        self._nf = nf

    def __getitem__(self, i):
        n = self._nf.num_layers
        return _LayerView(self._nf._node_frames, i % n)


class _NodeBatch:
    def __init__(self, data):
        self.data = data


class NodeFlow:
    """node_mapping: int64 [R] local ids, layer 0 (inputs) first, last layer = seeds.
    layer_offsets: python ints [num_layers+1]. Block i (layer i -> i+1) is CSR by
    destination: blk_indptr[i] int32 [|L(i+1)|+1], blk_src[i] int32 [edges] = position of the
    source vertex inside layer i."""

    def __init__(self, node_mapping, layer_offsets, blk_indptr, blk_src):
        self._node_mapping = _UserTensor(node_mapping)
        self._layer_offsets = [int(x) for x in layer_offsets]
        self.num_layers = len(self._layer_offsets) - 1
        self.num_blocks = self.num_layers - 1
        self.blk_indptr = blk_indptr
        self.blk_src = blk_src
        # source-major copy of a block (sampler option `transpose`): tptr int32 [|L(i)|+1], tdst int32 [edges] =
        # positions of the destinations inside layer i+1, ascending per source; None where not built
        self.blk_tptr = [None] * self.num_blocks
        self.blk_tdst = [None] * self.num_blocks
        self.blk_theavy = [None] * self.num_blocks   # [count, hub sources...] (more than PG_HEAVY_ROW edges)
        # {block index: aggregated rows} for blocks whose FIRST aggregation (of the raw feature rows) was run ahead of the step
        # by the trainer (GraphedTrainer.early_aggregate); the models then call apply_block instead of block_compute
        self._pre_agg = None
        self._node_frames = [None] * self.num_layers
        self.layers = _Layers(self)
        self.padded = False      # True: fixed-shape layout, ids < 0 are padding (sampler static=True)

    def actual_sizes(self):
        """(layer sizes, block edge counts) of a padded NodeFlow — synchronises with the sampler"""
        slot = self._slot
        slot.ready.synchronize()
        z = slot.sizes.tolist()
        from .. import _lib as L
        return z[:self.num_layers], z[L.PG_MAX_LAYERS:L.PG_MAX_LAYERS + self.num_blocks]

    def layer_size(self, i):
        i %= self.num_layers
        return self._layer_offsets[i + 1] - self._layer_offsets[i]

    def layer_parent_nid(self, i):
        i %= self.num_layers
        return self._node_mapping.tousertensor()[self._layer_offsets[i]:self._layer_offsets[i + 1]]

    def block_size(self, i):
        return int(self.blk_src[i].numel())

    def apply_block(self, i, agg, out_field, apply_node_func=None):
        """block_compute(i, ...) whose reduce has ALREADY been computed into `agg` [|L(i+1)|, dim] — by GraphedTrainer, ahead
        of the step (`_pre_agg`: an aggregation of raw features depends on no parameter): store it as layer i+1's
        `out_field` and run the node UDF"""
        dst = self.layers[i + 1].data
        dst[out_field] = agg
        if apply_node_func is not None:
            dst.update(apply_node_func(_NodeBatch(dst)))

    def block_compute(self, i, message_func, reduce_func, apply_node_func=None, dropout=None):
        """DGL's nf.block_compute for the builtin pair copy_src + mean|sum
        (gcn_nssc.py:71-74,139-142; graphsage_nssc.py:98-111): aggregate layer i's
        `src` field over block i into layer i+1's `out` field, then run the node UDF on layer i+1.
        `dropout` (an ops.DropoutSpec, not in DGL) applies the model's dropout to the source field inside
        the aggregation kernel."""
        src_field = message_func.src
        assert reduce_func.msg == message_func.out, "reduce must consume the message field"
        h = self.layers[i].data[src_field]
        agg = block_aggregate(self.blk_indptr[i], self.blk_src[i], h, self.layer_size(i + 1), reduce_func.op,
                              dropout=dropout, transpose=(self.blk_tptr[i], self.blk_tdst[i], self.blk_theavy[i]))
        dst = self.layers[i + 1].data
        dst[reduce_func.out] = agg
        if apply_node_func is not None:
            dst.update(apply_node_func(_NodeBatch(dst)))
