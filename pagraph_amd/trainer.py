"""The hot loop of examples/profile/pa_gcn.py:82-103 (and pa_gs.py) as a reusable
object, with the load/compute overlap the north star asks for:

  main stream   : model forward / backward / optimizer of batch k
  load stream   : cacher.fetch_data (gather kernel, miss path) + label lookup of batch k+1
  sampler stream: NeighborSampler's own stream, one more batch ahead

The host enqueues the compute of batch k first (asynchronous), then prepares batch
k+1 — so the only host-blocking part (the staged miss path's wait for the miss list
and its CPU row gather) runs while the GPU computes batch k.
"""
import time

import torch


def _record_stream(nf, stream):
    ts = [nf._node_mapping.tousertensor()] + list(nf.blk_indptr) + list(nf.blk_src)
    for fr in nf._node_frames:
        if fr:
            ts += list(fr.values())
    for t in ts:
        if t.is_cuda:
            t.record_stream(stream)


class Prepared:
    __slots__ = ("nf", "label", "event")


class MinibatchTrainer:
    def __init__(self, model, loss_fcn, optimizer, cacher, sampler, labels, device, overlap=True):
        self.model, self.loss_fcn, self.optimizer = model, loss_fcn, optimizer
        self.cacher, self.sampler, self.labels = cacher, sampler, labels
        self.device = device
        self.overlap = overlap
        self.load_stream = torch.cuda.Stream(device=device) if overlap else None
        self.on_step = None          # callback(step_in_epoch, loss_tensor)
        self.after_first_step = None  # callback() — pa_gcn.py:99-100 (auto_cache)
        self._first_done = False

    # -- 'gpu-load' (pa_gcn.py:87-91) ----------------------------------------
    def prepare(self, nf):
        p = Prepared()
        p.nf = nf
        if self.load_stream is None:
            with torch.autograd.profiler.record_function('gpu-load'):
                self.cacher.fetch_data(nf)
                p.label = self.labels[nf.layer_parent_nid(-1)]
            p.event = None
            return p
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(self.load_stream):
            with torch.autograd.profiler.record_function('gpu-load'):
                self.cacher.fetch_data(nf)
                p.label = self.labels[nf.layer_parent_nid(-1)]
            p.event = torch.cuda.Event()
            p.event.record(self.load_stream)
        _record_stream(nf, main)
        p.label.record_stream(main)
        return p

    # -- 'gpu-compute' (pa_gcn.py:92-97) --------------------------------------
    def compute(self, p):
        if p.event is not None:
            torch.cuda.current_stream(self.device).wait_event(p.event)
        with torch.autograd.profiler.record_function('gpu-compute'):
            pred = self.model(p.nf)
            loss = self.loss_fcn(pred, p.label)
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
        return loss

    def _next(self, it):
        if self.load_stream is None:
            return next(it, None)
        with torch.cuda.stream(self.load_stream):   # the sampler's hand-off copies land on the load stream
            return next(it, None)

    def run_steps(self, it, steps=None):
        """drive `steps` minibatches from iterator `it` (None = until exhausted); returns #steps"""
        done = 0
        nf = self._next(it)
        if nf is None:
            return 0
        cur = self.prepare(nf)
        while cur is not None:
            loss = self.compute(cur)
            done += 1
            if not self._first_done:
                self._first_done = True
                if self.after_first_step is not None:
                    torch.cuda.current_stream(self.device).synchronize()
                    self.after_first_step()
            nxt = None
            if steps is None or done < steps:
                nf = self._next(it)
                if nf is not None:
                    nxt = self.prepare(nf)
            if self.on_step is not None:
                self.on_step(done, loss)
            cur = nxt
        return done

    def run_epoch(self):
        """one pass over the sampler; returns (steps, seconds) with a device sync on both sides
        (the reference times without syncing, pa_gcn.py:84,105 — the sync makes the number honest)"""
        torch.cuda.synchronize(self.device)
        t0 = time.time()
        n = self.run_steps(iter(self.sampler))
        torch.cuda.synchronize(self.device)
        return n, time.time() - t0


def cycle_batches(sampler, steps):
    """iterator over exactly `steps` NodeFlows, wrapping into the next epoch when the
    sampler is exhausted (multi-GPU step equalisation, parallel.equalize_steps)"""
    produced = 0
    while produced < steps:
        got = False
        for nf in sampler:
            got = True
            yield nf
            produced += 1
            if produced >= steps:
                return
        if not got:
            return
