"""Data-parallel training over the GPUs of one node: one process per GPU, resident
replicas, bucketed gradient all-reduce on RCCL (torch.distributed backend "nccl" on
ROCm) overlapped with the hand-written backward.

Replaces ``nn.DataParallel(model)`` (train.py:970), whose per-step parameter broadcast,
input scatter, logits gather and gradient reduce onto GPU 0 all disappear:
  * parameters are broadcast once from rank 0 at construction;
  * each rank runs forward/backward on its shard of the batch with LOCAL BatchNorm
    statistics -- nn.DataParallel replicas normalise with their own shard too;
  * the backward pass reports every finished parameter gradient (GradSink.on_ready);
    gradients live in one flat arena laid out in registration order, which backward fills
    from the END (decoder/ASPP first, stem last), so a bucket is a contiguous tail slice
    that is handed to an asynchronous all-reduce(SUM) as soon as its last gradient is
    written.  Only the final (stem + layer1) bucket is exposed;
  * gradients are SUMMED, not averaged: the criterion divides by the GLOBAL sum of class
    weights (utils/loss.py ``group=``), which reproduces the reference's loss on the
    gathered full batch.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets default to 32 MB so each
ring step moves a few MB per link -- large enough to run at link rate, small enough that
the first bucket starts within a few ms of backward starting.
"""
import contextlib

import torch
import torch.distributed as dist
import torch.nn as nn

from .optim import arena_of


class DistributedDataParallelHIP(nn.Module):
    def __init__(self, module, process_group=None, bucket_mb=32.0, broadcast=True, time_waits=False):
        super().__init__()
        self.time_waits = bool(time_waits)       # bench.py: measure the exposed wait per step with ONE reusable event pair
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        params = [p for p in module.parameters()]
        self.arena = arena_of(params)
        if broadcast and self.world > 1:
            dist.broadcast(self.arena.data, src=dist.get_global_rank(process_group, 0) if process_group else 0,
                           group=process_group)
            for b in module.buffers():
                dist.broadcast(b, src=0, group=process_group)
        # buckets: contiguous arena ranges, cut from the end (backward order)
        limit = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []           # (start, end) element ranges, in all-reduce launch order
        self.bucket_of = {}
        end = self.arena.numel
        count = 0
        members = []
        for i in range(len(params) - 1, -1, -1):
            members.append(params[i])
            count = end - self.arena.offsets[i]
            if count >= limit or i == 0:
                b = len(self.buckets)
                self.buckets.append((self.arena.offsets[i], end))
                for p in members:
                    self.bucket_of[id(p)] = b
                members = []
                end = self.arena.offsets[i]
        self._need = [0] * len(self.buckets)
        for p in params:
            if p.requires_grad:
                self._need[self.bucket_of[id(p)]] += 1
        self._left = list(self._need)
        self._works = []
        self._sync = True
        # diagnostics for bench.py: what was all-reduced and how long the compute stream stood still for it
        self.stats = dict(bytes_allreduced=0, allreduces=0, steps=0, exposed_wait_ms=0.0)
        self._wait_pair = None      # (start, end) events of the previous step's wait, folded into stats at the next one
        module._iswm_on_ready = self._on_ready

    def forward(self, *args, **kwargs):
        self._left = list(self._need)
        return self.module(*args, **kwargs)

    @contextlib.contextmanager
    def no_sync(self):
        """gradient accumulation: backward passes inside this context only accumulate locally; the first backward outside
        it all-reduces the accumulated sum (same contract as torch's DistributedDataParallel.no_sync)"""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _on_ready(self, p):
        if self.world == 1 or not self._sync:
            return
        b = self.bucket_of[id(p)]
        self._left[b] -= 1
        if self._left[b] == 0:
            s, e = self.buckets[b]
            self._works.append(dist.all_reduce(self.arena.grad[s:e], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))
            self.stats["bytes_allreduced"] += (e - s) * 4
            self.stats["allreduces"] += 1

    def finish_grad_sync(self):
        """Block the compute stream until every bucket's all-reduce has landed (call before
        optimizer.step(); ``attach`` does it automatically)."""
        if self.world > 1 and self._sync:
            # torch's DDP raises when a bucket is left half-filled (a requires_grad parameter that got no gradient this
            # step): its all-reduce would never be launched and the replicas would silently diverge
            stuck = [b for b in range(len(self.buckets)) if 0 < self._left[b] < self._need[b]]
            if stuck:
                raise RuntimeError("DistributedDataParallelHIP: bucket(s) %s received only part of their gradients in this "
                                   "backward pass (a parameter with requires_grad=True took no part in the loss); freeze it "
                                   "with requires_grad_(False) before wrapping the model" % stuck)
        timed = self.time_waits and bool(self._works) and self.arena.grad.is_cuda
        if timed:
            self._fold_wait()
            if self._wait_pair is None:
                self._wait_pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), [False])
            self._wait_pair[0].record()
        for w in self._works:
            w.wait()
        if timed:
            self._wait_pair[1].record()
            self._wait_pair[2][0] = True
        if self._works:
            self.stats["steps"] += 1
        self._works = []

    def _fold_wait(self):
        """add the previous step's measured wait to the totals (the pair is re-recorded afterwards: nothing accumulates
        in a long training run)"""
        if self._wait_pair is not None and self._wait_pair[2][0]:
            a, b, live = self._wait_pair
            b.synchronize()
            self.stats["exposed_wait_ms"] += a.elapsed_time(b)
            live[0] = False

    def comm_stats(self):
        """totals since construction (call after a device synchronize): bytes and all-reduce launches, and the time the
        compute stream spent between reaching the optimizer step and the last bucket landing (exposed, not overlapped)"""
        self._fold_wait()
        out = dict(self.stats)
        out["buckets"] = len(self.buckets)
        out["backend"] = dist.get_backend(self.group) if dist.is_initialized() else "none"
        out["world_size"] = self.world
        return out

    def attach(self, optimizer):
        optimizer.register_step_pre_hook(lambda *a, **k: self.finish_grad_sync())
        return optimizer
