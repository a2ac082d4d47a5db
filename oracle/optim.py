"""CPU restatement of the reference's optimizer / LR-schedule step
(TEST INFRASTRUCTURE).  The reference builds stock torch optimizers with the
*default* learning rate (``--lr`` only reaches the scheduler's eta_min),
train.py:421-452; the arithmetic restated here is torch 2.x's single-tensor
update rule for each of them.
"""
import math

import torch


class OracleSGD:
    """torch.optim.SGD(params, momentum=0.9, weight_decay=wd, nesterov=True) with
    torch's default lr=1e-3 -- setup_optimizer, train.py:426-432."""

    def __init__(self, params, lr=1e-3, momentum=0.9, weight_decay=1e-4, nesterov=True):
        self.params = list(params)
        self.lr, self.momentum, self.wd, self.nesterov = lr, momentum, weight_decay, nesterov
        self.buf = [None] * len(self.params)

    @torch.no_grad()
    def step(self):
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            g = p.grad.add(p, alpha=self.wd) if self.wd != 0 else p.grad.clone()
            if self.buf[i] is None:
                self.buf[i] = g.clone()
            else:
                self.buf[i].mul_(self.momentum).add_(g)
            g = g.add(self.buf[i], alpha=self.momentum) if self.nesterov else self.buf[i]
            p.add_(g, alpha=-self.lr)


class OracleAdam:
    """torch.optim.Adam / AdamW defaults (lr 1e-3, betas (0.9,0.999), eps 1e-8) with
    weight_decay=wd -- setup_optimizer, train.py:433-442.  ``decoupled`` selects
    AdamW's p *= 1 - lr*wd instead of Adam's g += wd*p."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4,
                 decoupled=False):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, betas[0], betas[1], eps, weight_decay
        self.decoupled = decoupled
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for p, m, v in zip(self.params, self.m, self.v):
            if p.grad is None:
                continue
            g = p.grad
            if self.decoupled:
                p.mul_(1 - self.lr * self.wd)
            elif self.wd != 0:
                g = g.add(p, alpha=self.wd)
            m.lerp_(g, 1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(m, denom, value=-self.lr / bc1)


def cosine_lr(base_lr, t, t_max, eta_min):
    """Closed form of CosineAnnealingLR(T_max=total_itrs, eta_min=lr*0.01) after t
    scheduler steps -- setup_scheduler, train.py:446-452 (stepped per iteration,
    train.py:1103)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / t_max)) / 2
