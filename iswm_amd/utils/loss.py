"""Criteria of the training step on the fused HIP loss kernel -- counterpart of the
reference's utils/loss.py (FocalLoss :14-35, create_loss :37-39) and of the criterion
built by setup_criterion, train.py:454-459 (nn.CrossEntropyLoss(weight, ignore_index=255)).

``criterion(logits, labels)`` keeps the reference's call shape (train.py:1046): logits
NCHW fp32 on the GPU, labels [B,H,W] int64 (or uint8, as the dataset produces them).
One kernel pass computes the loss terms AND the unnormalised gradient; backward only
rescales it.  Under data parallelism ``group`` makes the weighted-mean normaliser global
(sum of class weights over ALL ranks' pixels), which is what the reference's
gathered-logits loss under nn.DataParallel computes (train.py:970,1045-1046).
"""
import math

import torch
import torch.nn as nn

from .. import ops


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, weight, ignore_index, alpha, gamma, mode, group):
        loss, sums, grad = ops.loss_fwd(logits, labels, weight, ignore_index, alpha, gamma, mode)
        npix = labels.numel()
        if group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(group)
            if mode == ops.MODE_WCE:
                dist.all_reduce(sums, group=group)              # global sum(w*nll), sum(w)
                loss = (sums[0] / sums[1]).reshape(1)
            else:
                if mode == ops.MODE_FOCAL_MEAN:
                    npix = npix * world
                dist.all_reduce(sums, group=group)
                loss = (sums[0] / npix if mode == ops.MODE_FOCAL_MEAN else sums[0]).reshape(1)
        ctx.save_for_backward(grad, sums)
        ctx.mode, ctx.npix = mode, npix
        return loss.reshape(())

    @staticmethod
    def backward(ctx, upstream):
        grad, sums = ctx.saved_tensors
        up = upstream.reshape(1).contiguous().to(torch.float32)
        ops.loss_bwd_scale(grad, sums, up, ctx.mode, ctx.npix)
        return grad, None, None, None, None, None, None, None


def _check(logits, labels):
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4):
        raise ValueError("loss expects fp32 CUDA logits [B,C,H,W] (there is no CPU fallback)")
    if labels.dtype not in (torch.int64, torch.uint8):
        labels = labels.long()
    return labels.to(logits.device)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(weight=w, ignore_index=255, reduction='mean') as train.py:454-459
    builds it ('ce_loss': weight=None; 'IWce_loss': weight=[1, sqrt(N_black/N_white)])."""

    def __init__(self, weight=None, ignore_index=255, reduction='mean', group=None):
        super().__init__()
        if reduction != 'mean':
            raise NotImplementedError("the training path uses reduction='mean'")
        self.register_buffer('weight', None if weight is None else weight.detach().float().clone())
        self.ignore_index = ignore_index
        self.group = group

    def forward(self, inputs, targets):
        targets = _check(inputs, targets)
        w = None if self.weight is None else self.weight.to(inputs.device)
        return _LossFn.apply(inputs, targets, w, self.ignore_index, 1.0, 0.0, ops.MODE_WCE, self.group)


class FocalLoss(nn.Module):
    """FocalLoss(alpha=1, gamma=0, size_average=True, ignore_index=255, weight=None) --
    reference utils/loss.py:14-35 (mean runs over ALL pixels, ignored ones included)."""

    def __init__(self, alpha=1, gamma=0, size_average=True, ignore_index=255, weight=None, group=None):
        super(FocalLoss, self).__init__()
        self.alpha = alpha
        self.gamma = gamma
        self.ignore_index = ignore_index
        self.size_average = size_average
        self.weight = weight
        self.group = group

    def forward(self, inputs, targets):
        targets = _check(inputs, targets)
        w = None if self.weight is None else self.weight.to(device=inputs.device, dtype=torch.float32)
        mode = ops.MODE_FOCAL_MEAN if self.size_average else ops.MODE_FOCAL_SUM
        return _LossFn.apply(inputs, targets, w, self.ignore_index, float(self.alpha), float(self.gamma), mode,
                             self.group)


def create_loss(loss_type="focal", temporal_loss="none", temporal_weight=0.5, **kwargs):
    """reference utils/loss.py:37-39"""
    return FocalLoss(**kwargs)


def calculate_class_weights(loader, group=None):
    """[1, sqrt(N_black / N_white)] over one pass of the loader -- train.py:388-410.

    Under data parallelism every rank iterates its own shard of the train set (DistributedSampler); the reference counts
    the WHOLE set, and the criterion's global normaliser assumes one common weight vector, so the two pixel counts are
    summed over the process group (int64 all-reduce) before the ratio is taken: every rank gets the same weights."""
    black = white = 0
    for batch in loader:
        labels = batch['mask'] if isinstance(batch, dict) else batch[1]
        black += int((labels == 0).sum())
        white += int((labels == 1).sum())
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
            cnt = torch.tensor([black, white], dtype=torch.int64, device=dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
            black, white = int(cnt[0]), int(cnt[1])
    except ImportError:
        pass
    return torch.tensor([1.0, math.sqrt(black / white)], dtype=torch.float32)
