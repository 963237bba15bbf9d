"""Sampled GraphSAGE (mean / gcn / pool aggregators) — constructor, parameter names
(`layers.N.fc_self`, `layers.N.fc_neigh`) and dataflow of
PaGraph/model/graphsage_nssc.py:6-134 on pagraph_amd's NodeFlow.  'pool' (the
constructor's default, :38) is DGL's fn.max reducer (:106-110) = PG_REDUCE_MAX."""
import torch
import torch.nn as nn

from .. import function as fn
from .. import ops
from ._dropout import FusedDropoutMixin


class NodeUpdate(nn.Module):
    """graphsage_nssc.py:6-30"""

    def __init__(self, in_feats, out_feats, activation=None, concat=False):
        super().__init__()
        self.fc_neigh = nn.Linear(in_feats, out_feats)
        self.fc_self = nn.Linear(in_feats, out_feats)
        self.activation = activation
        self.concat = concat
        gain = nn.init.calculate_gain('relu')
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)

    def forward(self, node):
        relu = self.activation in (torch.relu, torch.nn.functional.relu)
        if self.concat and relu:                 # both GEMMs, the add and the skip-concat in one kernel
            h = ops.linear2(node.data['h'], self.fc_self, node.data['neigh'], self.fc_neigh, ops.ACT_CONCAT)
        elif relu:
            h = ops.linear2(node.data['h'], self.fc_self, node.data['neigh'], self.fc_neigh, ops.ACT_RELU)
        else:
            h = ops.linear2(node.data['h'], self.fc_self, node.data['neigh'], self.fc_neigh)
            if self.concat:
                h = torch.cat((h, self.activation(h)), dim=1)
            elif self.activation:
                h = self.activation(h)
        return {'activation': h}


_REDUCERS = {'mean': fn.mean, 'gcn': fn.sum, 'pool': fn.max}


class GraphSageSampling(FusedDropoutMixin, nn.Module):
    @property
    def deferrable_parameters(self):
        """model layer `lid` is applied to every block >= lid: with n_layers == 1 (pa_gs.py:134) a parameter receives at
        most TWO gradient contributions per step — what ops.defer_partials / pg_adam_step fold into the
        optimiser's launch. Deeper stacks (three and more uses) and the preprocess variant keep the separate sums."""
        return self.n_layers == 1 and not self.preprocess

    def __init__(self, in_feats, n_hidden, n_classes, n_layers, activation=None, dropout=0.,
                 aggregator_type='pool', preprocess=False):
        super().__init__()
        self.preprocess = preprocess
        self.n_layers = n_layers
        self.dropout = nn.Dropout(dropout)
        self._init_fused_dropout()
        self.activation = activation
        self.aggregator_type = aggregator_type
        self.layers = nn.ModuleList()
        # 'lstm' (graphsage_nssc.py:58-71): the reference builds one nn.LSTM per block — same modules here, so that a
        # state_dict moves either way — but its forward cannot run (see forward)
        lstm = aggregator_type == 'lstm'
        self.reducer = nn.ModuleList()
        if preprocess:
            self.fc_self = nn.Linear(in_feats, n_hidden)
            self.fc_neigh = nn.Linear(in_feats, n_hidden)
        else:
            if lstm:
                self.reducer.append(nn.LSTM(in_feats, in_feats, batch_first=True))
            self.layers.append(NodeUpdate(in_feats, n_hidden, activation, concat=(n_layers == 1)))
        for i in range(1, n_layers):
            if lstm:
                self.reducer.append(nn.LSTM(n_hidden, n_hidden, batch_first=True))
            self.layers.append(NodeUpdate(n_hidden, n_hidden, activation, concat=(i == n_layers - 1)))
        if lstm:
            self.reducer.append(nn.LSTM(2 * n_hidden, 2 * n_hidden, batch_first=True))
        self.layers.append(NodeUpdate(2 * n_hidden, n_classes))

    def required_inputs(self, num_layers):
        """every layer's 'features' (+ 'neigh' under preprocess) is read (graphsage_nssc.py:75-90)"""
        f = ['features', 'neigh'] if self.preprocess else ['features']
        return {l: list(f) for l in range(num_layers)}

    def virtual_inputs(self, num_layers):
        """Every layer's raw 'features' may stay un-materialised (ops.RowSource): they are read by model layer 0 only —
        as the source rows of a block's aggregation (graphsage_nssc.py:92-111: pg_spmm_fwd_rows) and as the self term
        fc_self(h) of the block's destinations (:24: pg_linear_fwd (rows in place) / pg_linear_bwd_w (rows in place), which need at most 64
        hidden units); from model layer 1 on a layer's 'h' is the previous activation. With a wider hidden layer only
        layer 0 (a source of block 0 and nothing else) stays virtual. Not under preprocess (every layer goes through
        fc_self / fc_neigh first, :76-87)."""
        if self.preprocess:
            return {}
        if self.layers[0].fc_self.out_features > 64:
            return {0: ['features']}
        return {l: ['features'] for l in range(num_layers)}

    def early_aggregations(self, num_layers, step_value):
        """[(block, field, reduce name, DropoutSpec or None)] (see _GCNBase.early_aggregations): model layer 0 aggregates
        the raw 'features' of EVERY block's source layer (graphsage_nssc.py:92-111) — none of them depends on a parameter"""
        if self.preprocess or self.aggregator_type not in _REDUCERS:
            return []
        out = []
        for i in range(num_layers - 1):
            drop = self._early_drop_spec(i, step_value)          # (tag lid * 16 + i with lid = 0)
            if drop is False:
                return []
            out.append((i, 'features', _REDUCERS[self.aggregator_type]('m', 'neigh').op, drop))
        return out

    def forward_loss(self, nf, labels, n_valid, grad_seed=None, ignore_index=-100, want_logits=False):
        """CrossEntropyLoss(self(nf), labels) with the LAST model layer — the output NodeUpdate fc_neigh(neigh) + fc_self(h),
        applied to the seeds' block only (graphsage_nssc.py:92-131 with lid = n_layers) — its aggregation (+ dropout), the
        loss and all their gradients in ONE kernel (ops.sage_head; round 4): nine launches of the replayed step become one.
        Returns the loss, or None where the fused kernel does not apply (preprocess variant, 'pool' / 'lstm', more than 64
        input columns or classes, logits wanted) — the caller then runs forward() and its loss function."""
        last = self.layers[-1]
        if (self.preprocess or want_logits or self.aggregator_type not in ('mean', 'gcn') or last.concat
                or last.activation is not None or not labels.is_cuda or len(self.layers) < 2):
            return None
        # the head's envelope is checked BEFORE the layers below it run: they pop every frame's 'features', so a head that
        # declined afterwards left the caller's fall-back forward(nf) a consumed NodeFlow (--n-hidden 32)
        drop_active = bool(self.training and 0.0 < self.dropout.p)
        if (labels.dtype != torch.int64 or labels.numel() == 0
                or not ops.head_fits(last.fc_neigh.in_features, last.fc_self.in_features, last.fc_neigh.out_features,
                                     (last.fc_neigh.weight, last.fc_self.weight), drop_active)):
            return None
        L = nf.num_layers
        self._forward_layers(nf, upto=len(self.layers) - 1)
        lid, i = len(self.layers) - 1, L - 2
        src_h, self_h = nf.layers[i].data['h'], nf.layers[i + 1].data['h']
        red = _REDUCERS[self.aggregator_type]
        loss = None
        drop = self._drop_spec(lid * 16 + i, src_h) if self.training else None
        if torch.is_tensor(src_h) and torch.is_tensor(self_h):
            h_in = src_h
            if drop is None and drop_active:
                h_in = self._dropout_or_raise(src_h)
            loss = ops.sage_head(nf.blk_indptr[i], nf.blk_src[i], h_in, last.fc_neigh, self_h, last.fc_self, labels, n_valid,
                                 grad_seed, ignore_index, red('m', 'neigh').op, drop,
                                 (nf.blk_tptr[i], nf.blk_tdst[i], nf.blk_theavy[i]))
        if loss is None:
            # what only the run-time tensors can say (strides, a row source instead of a tensor): the last layer finishes
            # UNFUSED from the state the layers below left — never `return None` with the frames consumed
            d = nf.layers[i].data
            if drop is None:
                d['h'] = self._dropout_or_raise(d.pop('h'))
            nf.block_compute(i, fn.copy_src(src='h', out='m'), red('m', 'neigh'), last, dropout=drop)
            loss = ops.cross_entropy(nf.layers[L - 1].data.pop('activation'), labels, ignore_index)
        return loss

    def forward(self, nf):
        h = self._forward_layers(nf, upto=len(self.layers))
        return h

    def _forward_layers(self, nf, upto):
        """model layers [0, upto) of forward(); upto == len(self.layers): the whole model, returns the seeds' output"""
        L = nf.num_layers
        self._bump_drop_step()
        if self.preprocess:
            # graphsage_nssc.py:75-87: every layer carries 'features' and a pre-aggregated 'neigh'
            for i in range(L):
                d = nf.layers[i].data
                h = self.dropout(d.pop('features'))
                h = ops.linear(h, self.fc_self) + ops.linear(d.pop('neigh'), self.fc_neigh)
                d['h'] = torch.cat((h, self.activation(h)), dim=1) if self.n_layers == 1 else self.activation(h)
        else:
            for i in range(L):
                d = nf.layers[i].data
                d['h'] = d.pop('features')
        if self.aggregator_type == 'lstm':
            # graphsage_nssc.py:112-121 hands block_compute a reduce UDF declared `_reducer(self, nodes)`; DGL calls reduce
            # UDFs with the node batch alone, so the reference's own forward stops here with this very TypeError
            raise TypeError("_reducer() missing 1 required positional argument: 'nodes'")
        if self.aggregator_type not in _REDUCERS:
            raise KeyError('Aggregator type {} not recognized.'.format(self.aggregator_type))     # :126-127
        red = _REDUCERS[self.aggregator_type]
        # graphsage_nssc.py:92-131: model layer `lid` is applied to every block i >= lid, so the
        # self term of a destination always has the same depth as its neighbour term.
        pre = getattr(nf, '_pre_agg', None) or {}
        for lid, layer in enumerate(self.layers):
            if lid >= upto:
                return True                             # (forward_loss takes over: the last layer runs inside the loss head)
            for i in range(lid, L - 1):
                if lid == 0 and i in pre:               # block i's aggregation of the raw rows ran ahead of the step
                    nf.apply_block(i, pre[i], 'neigh', layer)
                    continue
                d = nf.layers[i].data
                # the dropped 'h' is read by this aggregation only (layer i's self term was consumed by block
                # i - 1 already), so the mask can be applied inside the kernel (tag: one per (lid, i) call site)
                drop = self._drop_spec(lid * 16 + i, d['h'])
                if drop is None:
                    d['h'] = self._dropout_or_raise(d.pop('h'))
                nf.block_compute(i, fn.copy_src(src='h', out='m'), red('m', 'neigh'), layer, dropout=drop)
            for i in range(lid + 1, L):
                d = nf.layers[i].data
                d['h'] = d.pop('activation')
        return nf.layers[L - 1].data.pop('h')
