"""2-layer sampled GCN with skip-concat — same constructor, parameter names
(`layers.N.linear.{weight,bias}`, `linear.*` under preprocess) and maths as
PaGraph/model/gcn_nssc.py:6-164, running on pagraph_amd's NodeFlow: the
aggregation is the HIP SpMM (pg_spmm.hip; layer 0 straight from the feature cache when
the cacher hands over an ops.RowSource), the skinny dense step the fp32-MFMA kernels of
pg_dense.hip (ops.linear), the training output layer + loss one kernel (pg_head.hip).
Pinned against the reference's classes by tests/golden/g7_*."""
import torch
import torch.nn as nn

from .. import function as fn
from .. import ops
from ._dropout import FusedDropoutMixin


class NodeUpdate(nn.Module):
    """gcn_nssc.py:6-24"""

    def __init__(self, in_feats, out_feats, activation=None, test=False, concat=False):
        super().__init__()
        self.linear = nn.Linear(in_feats, out_feats)
        self.activation = activation
        self.concat = concat
        self.test = test

    def forward(self, node):
        h = node.data['h']
        if self.test:
            h = h * node.data['norm']
        relu = self.activation in (torch.relu, torch.nn.functional.relu)
        if self.concat and relu:                 # dense step + skip-concat fused in one kernel
            h = ops.linear(h, self.linear, ops.ACT_CONCAT)
        elif self.activation is not None and relu and not self.concat:
            h = ops.linear(h, self.linear, ops.ACT_RELU)
        else:
            h = ops.linear(h, self.linear)
            if self.concat:
                h = torch.cat((h, self.activation(h)), dim=1)
            elif self.activation:
                h = self.activation(h)
        return {'activation': h}


def _stack(in_feats, n_hidden, n_classes, n_layers, activation, preprocess, test):
    """layer list of gcn_nssc.py:45-58 (training) / :114-128 (inference)"""
    layers = nn.ModuleList()
    if not preprocess:
        layers.append(NodeUpdate(in_feats, n_hidden, activation, test=test, concat=(n_layers == 1)))
    for i in range(1, n_layers):
        layers.append(NodeUpdate(n_hidden, n_hidden, activation, test=test, concat=(i == n_layers - 1)))
    layers.append(NodeUpdate(2 * n_hidden, n_classes, test=test))
    return layers


class _GCNBase(FusedDropoutMixin, nn.Module):
    reducer = fn.mean
    uses_norm = False
    # every NodeUpdate (and the preprocess transform) is applied once per forward: each parameter receives exactly one
    # gradient contribution per step (what ops.defer_partials needs)
    deferrable_parameters = True

    def required_inputs(self, num_layers):
        """{layer: [fields]} this model reads from the NodeFlow frames (for fetch_data(need=...)):
        training GCN touches only layer 0's 'features' (gcn_nssc.py:64,81); inference also multiplies
        every destination layer by 'norm' (gcn_nssc.py:16-17)."""
        need = {0: ['features']}
        if self.uses_norm:
            for l in range(1, num_layers):
                need[l] = ['norm']
            need = {l: need.get(l, []) for l in range(num_layers)}
        return need

    def virtual_inputs(self, num_layers):
        """{layer: [fields]} this model only ever AGGREGATES (never reads row by row): the cacher may hand those over
        as ops.RowSource — the rows stay in the cache / the staged miss block and the layer-0 aggregation reads them
        there (gcn_nssc.py:64-74 fused with storage.py:176-204). Without preprocessing that is layer 0's 'features';
        with it the raw features feed a dense transform first (:81-84)."""
        if self.preprocess or len(self.layers) < 2:
            return {}
        return {0: ['features']}

    def early_aggregations(self, num_layers, step_value):
        """[(block, field, reduce name, DropoutSpec or None)]: the aggregations of RAW feature rows this model starts with —
        they depend on no parameter, so a trainer may run them ahead of the step (GraphedTrainer.early_aggregate) and hand
        the results over as nf._pre_agg. (seed, layer tag) of the masks are the model's; the step value is the caller's
        count of what the model's own counter will hold when the batch is computed (then the masks are the in-step path's).
        [] when the model's first operation on its input is not such an aggregation."""
        if self.preprocess or len(self.layers) < 2 or self.uses_norm:
            return []
        drop = self._early_drop_spec(0, step_value)
        if drop is False:
            return []
        return [(0, 'features', self.reducer(msg='m', out='h').op, drop)]

    def _input_transform(self, nf):
        """gcn_nssc.py:80-90: dense transform of the raw features before any aggregation"""
        h = nf.layers[0].data['features']
        if getattr(self, 'dropout', None):
            h = self.dropout(h)
        h = ops.linear(h, self.linear)
        if self.n_layers == 1:
            return torch.cat((h, self.activation(h)), dim=1)
        return self.activation(h)

    def _propagate(self, nf, h):
        for i, layer in enumerate(self.layers):
            if self._apply_pre_aggregated(nf, i, layer):  # aggregated ahead of the step, dropout included
                h = nf.layers[i + 1].data.pop('activation')
                continue
            drop = None
            if getattr(self, 'dropout', None) and not self.preprocess:
                drop = self._drop_spec(i, h)             # dropout inside the aggregation kernel ...
                if drop is None:
                    h = self._dropout_or_raise(h)        # ... or nn.Dropout where that cannot be done
            nf.layers[i].data['h'] = h
            nf.block_compute(i, fn.copy_src(src='h', out='m'), self.reducer(msg='m', out='h'), layer, dropout=drop)
            h = nf.layers[i + 1].data.pop('activation')
        return h

    def _apply_pre_aggregated(self, nf, i, layer):
        pre = getattr(nf, '_pre_agg', None)
        if i != 0 or self.preprocess or not pre or 0 not in pre:
            return False
        nf.apply_block(0, pre[0], 'h', layer)
        return True

    def forward(self, nf):
        self._bump_drop_step()
        if self.preprocess:
            return self._propagate(nf, self._input_transform(nf))
        return self._propagate(nf, nf.layers[0].data['features'])

    def forward_loss(self, nf, labels, n_valid, grad_seed=None, ignore_index=-100, want_logits=False):
        """CrossEntropyLoss(self(nf), labels) with the output layer, the loss and their gradients in ONE kernel
        (ops.gcn_head): everything up to the last block runs as in forward(), then the last aggregation
        (+ dropout), the output NodeUpdate (gcn_nssc.py:58: plain linear) and the loss are fused. Returns the
        loss (or (loss, logits)), or None when the fused kernel does not apply (inference model, > 64
        classes / hidden columns, CPU tensors) — the caller then uses forward() and its loss function.
        n_valid: device int32[1] = number of labels != ignore_index."""
        last = self.layers[-1]
        if self.uses_norm or last.test or last.concat or last.activation is not None or not labels.is_cuda:
            return None
        # the head's static envelope BEFORE the layers below it run (a head that declines afterwards made the caller's
        # fall-back forward(nf) run those layers a second time — inside a captured step, every step)
        drop_active = bool(self.training and getattr(self, 'dropout', None) and not self.preprocess and self.dropout.p > 0)
        if (labels.dtype != torch.int64 or labels.numel() == 0
                or not ops.head_fits(last.linear.in_features, 0, last.linear.out_features, (last.linear.weight,), drop_active)):
            return None
        self._bump_drop_step()
        h = self._input_transform(nf) if self.preprocess else nf.layers[0].data['features']
        n = len(self.layers)
        for i, layer in enumerate(self.layers[:-1]):
            if self._apply_pre_aggregated(nf, i, layer):
                h = nf.layers[i + 1].data.pop('activation')
                continue
            drop = None
            if getattr(self, 'dropout', None) and not self.preprocess:
                drop = self._drop_spec(i, h)
                if drop is None:
                    h = self._dropout_or_raise(h)
            nf.layers[i].data['h'] = h
            nf.block_compute(i, fn.copy_src(src='h', out='m'), self.reducer(msg='m', out='h'), layer, dropout=drop)
            h = nf.layers[i + 1].data.pop('activation')
        i = n - 1
        drop = None
        if getattr(self, 'dropout', None) and not self.preprocess and self.training:
            drop = self._drop_spec(i, h)
            if drop is None:
                h = self.dropout(h)
        out = None
        if torch.is_tensor(h):
            out = ops.gcn_head(nf.blk_indptr[i], nf.blk_src[i], h, last.linear, labels, n_valid, grad_seed, ignore_index,
                               self.reducer(msg='m', out='h').op, drop,
                               (nf.blk_tptr[i], nf.blk_tdst[i], nf.blk_theavy[i]), want_logits)
        if out is None:
            # only the run-time tensor could say no (strides): the output layer finishes unfused from the state the layers
            # below left, instead of handing the caller a None it answers with a second run of the whole model
            nf.layers[i].data['h'] = h
            nf.block_compute(i, fn.copy_src(src='h', out='m'), self.reducer(msg='m', out='h'), last, dropout=drop)
            logits = nf.layers[i + 1].data.pop('activation')
            loss = ops.cross_entropy(logits, labels, ignore_index)
            out = (loss, logits) if want_logits else loss
        return out


class GCNSampling(_GCNBase):
    """gcn_nssc.py:27-100 — mean aggregation, dropout before every aggregation"""

    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation, dropout, preprocess=False):
        super().__init__()
        self.preprocess = preprocess
        self.n_layers = n_layers
        self.dropout = nn.Dropout(p=dropout) if dropout != 0 else None
        self._init_fused_dropout()
        if preprocess:
            self.linear = nn.Linear(in_feats, n_hidden)
            self.activation = activation
        self.layers = _stack(in_feats, n_hidden, n_classes, n_layers, activation, preprocess, test=False)


class GCNInfer(_GCNBase):
    """gcn_nssc.py:103-164 — sum aggregation scaled by `norm` inside NodeUpdate(test=True)"""
    reducer = fn.sum
    uses_norm = True

    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation, preprocess=False):
        super().__init__()
        self.preprocess = preprocess
        self.n_layers = n_layers
        self._init_fused_dropout()
        if preprocess:
            self.linear = nn.Linear(in_feats, n_hidden)
            self.activation = activation
        self.layers = _stack(in_feats, n_hidden, n_classes, n_layers, activation, preprocess, test=True)
