"""nn.Dropout of the sampled models folded into the aggregation kernel that consumes it.

The reference applies `self.dropout(h)` to a layer's input right before `nf.block_compute`
(PaGraph/model/gcn_nssc.py:66-69, graphsage_nssc.py:86-89). Here the model hands the aggregation a
DropoutSpec instead (ops.DropoutSpec / pg_dropout_t): the kernel draws the keep-mask from a counter-based
RNG keyed by (torch.initial_seed(), layer, rank, step). `step` lives in a device buffer bumped once per
training forward, so a replayed hipGraph gets a new mask each step. Inputs the kernel cannot take
(CPU tensors, dim % 4 != 0) and eval mode go through nn.Dropout unchanged."""
import torch

from .. import ops


class FusedDropoutMixin:
    def _init_fused_dropout(self):
        # non-persistent: the state_dict keeps the reference's keys
        self.register_buffer('_drop_step', torch.zeros(1, dtype=torch.int64), persistent=False)
        self._drop_seed = None
        self.fuse_dropout = True
        # True only WHILE a caller that advances _drop_step itself runs the forward (GraphedTrainer's deferred step lets the
        # optimiser's launch do it: one kernel less per replayed step) — see drop_step_external().
        self._drop_step_external = False
        # True: the counter already holds the value the NEXT forward uses (an external owner bumps it AFTER each step);
        # False: it holds the value the last forward used (this class's own bump-then-use order).
        self._drop_step_primed = False

    def _bump_drop_step(self):
        if self.training and self._drop_step.is_cuda and not self._drop_step_external:
            # (after an external owner the counter is one ahead already: bumping again skips one value, which is harmless
            # — the masks only have to differ from step to step)
            self._drop_step.add_(1)
            self._drop_step_primed = False

    def externalise_drop_step(self):
        """make the counter hold the value the next forward uses (one eager increment, only when it does not already)
        and return it; the caller then advances it once AFTER every step. Does not switch the model's own bump off:
        wrap the forward in drop_step_external() for that."""
        if not self._drop_step_primed:
            self._drop_step.add_(1)
            self._drop_step_primed = True
        return self._drop_step

    def drop_step_external(self):
        """context manager: inside it the model's forward leaves the counter alone (the caller advances it); outside,
        every training forward bumps it as before — a model handed from a GraphedTrainer to an eager loop keeps drawing
        a fresh mask per step (the hand-over used to be permanent and silently froze the mask)."""
        return _ExternalDropStep(self)

    def _drop_spec(self, layer, h):
        """DropoutSpec for aggregating `h` as the input of block `layer`, or None (use nn.Dropout)"""
        mod = getattr(self, 'dropout', None)
        if not (self.fuse_dropout and self.training and isinstance(mod, torch.nn.Dropout) and 0.0 < mod.p < 1.0):
            return None
        if not (self._drop_step.is_cuda and ops.DropoutSpec.fusable(h)):
            return None
        if self._drop_seed is None:
            self._drop_seed = torch.initial_seed()
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        return ops.DropoutSpec(mod.p, self._drop_seed, (rank << 8) | (layer & 0xFF), self._drop_step)

    def _early_drop_spec(self, tag, step_value):
        """_drop_spec for an aggregation somebody runs ahead of the step with the step value as an immediate
        (pg_dropout_t.step_value): None = dropout inactive, False = active but not fusable (the caller must not run ahead)"""
        mod = getattr(self, 'dropout', None)
        if not (self.training and isinstance(mod, torch.nn.Dropout) and mod.p > 0.0):
            return None
        if not (self.fuse_dropout and mod.p < 1.0 and self._drop_step.is_cuda):
            return False
        if self._drop_seed is None:
            self._drop_seed = torch.initial_seed()
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        return ops.DropoutSpec(mod.p, self._drop_seed, (rank << 8) | (tag & 0xFF), None, step_value)

    def _dropout_or_raise(self, h):
        """nn.Dropout for an input the aggregation kernel could not take the mask for; rows that were never
        materialised (ops.RowSource) have no tensor to drop from: identity when dropout is inactive, else refuse"""
        mod = getattr(self, 'dropout', None)
        if isinstance(h, ops.RowSource):
            if self.training and isinstance(mod, torch.nn.Dropout) and mod.p > 0.0:
                raise ops.L.PgError("dropout on un-materialised feature rows needs the fused mask (fuse_dropout=True, "
                                    "dim % 4 == 0); fetch the layer densely instead (virtual=None)")
            return h
        return mod(h) if mod is not None else h


class _ExternalDropStep:
    def __init__(self, model):
        self.model = model

    def __enter__(self):
        self.prev = self.model._drop_step_external
        self.model._drop_step_external = True
        return self.model._drop_step

    def __exit__(self, *exc):
        self.model._drop_step_external = self.prev
        return False
