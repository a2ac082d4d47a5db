"""PolyLR -- counterpart of the reference's utils/scheduler.py:3-11 (exported by utils/__init__.py:3; train.py itself
uses CosineAnnealingLR, train.py:446-452).  Pure host arithmetic: the fused optimizers read their learning rate from
`param_groups` every step, so any torch LR scheduler drives them."""
from torch.optim.lr_scheduler import _LRScheduler, StepLR  # noqa: F401  (StepLR re-exported like the reference module)


class PolyLR(_LRScheduler):
    """lr = max(base_lr * (1 - iter / max_iters) ** power, min_lr)"""

    def __init__(self, optimizer, max_iters, power=0.9, last_epoch=-1, min_lr=1e-6):
        self.power = power
        self.max_iters = max_iters  # avoid zero lr
        self.min_lr = min_lr
        super(PolyLR, self).__init__(optimizer, last_epoch)

    def get_lr(self):
        return [max(base_lr * (1 - self.last_epoch / self.max_iters) ** self.power, self.min_lr)
                for base_lr in self.base_lrs]
