"""CPU restatement of the reference's losses and class-weight rule
(TEST INFRASTRUCTURE).  Written as explicit per-pixel math -- not a call to
``F.cross_entropy`` -- so the fused HIP kernel has a formula to be checked
against; autograd differentiates it for the gradient check.
"""
import math

import torch


def _per_pixel(logits, labels, weight, ignore_index):
    """w[y]*nll and w[y] per pixel, 0 where y == ignore_index."""
    b, c, h, w = logits.shape
    lse = torch.logsumexp(logits, dim=1)                       # [B,H,W]
    valid = labels != ignore_index
    y = torch.where(valid, labels, torch.zeros_like(labels))
    zy = logits.gather(1, y.unsqueeze(1)).squeeze(1)
    nll = lse - zy
    wy = torch.ones(c, dtype=logits.dtype) if weight is None else weight
    zero = torch.zeros_like(nll)
    # torch.where (not a multiply by 0): an ignored pixel contributes exactly 0 and
    # receives exactly 0 gradient even when the focal factor's derivative is inf at
    # ce == 0 (gamma < 1) -- that is what F.cross_entropy's ignore_index does.
    return torch.where(valid, wy[y] * nll, zero), torch.where(valid, wy[y], zero)


def weighted_ce(logits, labels, weight=None, ignore_index=255):
    """nn.CrossEntropyLoss(weight=w, ignore_index=255, reduction='mean') as used by
    setup_criterion, train.py:454-459:  sum_i w[y_i]*nll_i / sum_i w[y_i] over
    pixels with y_i != 255."""
    wnll, wpix = _per_pixel(logits, labels, weight, ignore_index)
    return wnll.sum() / wpix.sum()


def focal_loss(logits, labels, alpha=1.0, gamma=0.0, size_average=True,
               ignore_index=255, weight=None):
    """FocalLoss.forward, utils/loss.py:23-35: ce_i = w[y_i]*nll_i (0 if ignored),
    pt = exp(-ce), f = alpha*(1-pt)**gamma*ce, mean over ALL B*H*W (or sum)."""
    ce, _ = _per_pixel(logits, labels, weight, ignore_index)
    pt = torch.exp(-ce)
    f = alpha * (1 - pt) ** gamma * ce
    return f.mean() if size_average else f.sum()


def class_weights(labels):
    """calculate_class_weights, train.py:388-410: [1, sqrt(N_black/N_white)]."""
    black = int((labels == 0).sum())
    white = int((labels == 1).sum())
    return torch.tensor([1.0, math.sqrt(black / white)], dtype=torch.float32)
