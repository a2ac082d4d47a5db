"""Flat-arena optimizers on the fused HIP update kernels.

setup_optimizer (train.py:421-444) builds torch.optim.SGD(momentum=0.9, nesterov=True,
weight_decay) / Adam / AdamW over ``model.parameters()`` with torch's DEFAULT lr (1e-3;
``--lr`` only reaches the scheduler's eta_min -- reproduced).  These
classes take the same arguments and are ``torch.optim.Optimizer`` subclasses, so
CosineAnnealingLR (train.py:446-452), ``state_dict()`` and ``zero_grad()`` work unchanged --
but all parameters live in ONE fp32 allocation (ParamArena), gradients in a second and
optimizer state in a third, so a step is a single kernel launch over 40-59 M elements and
the gradient arena can be all-reduced in large contiguous buckets with no copies.
"""
import math

import torch

from . import ops


def _align4(n):
    return (n + 3) // 4 * 4


class ParamArena:
    """Re-homes a list of parameters into one flat fp32 buffer (keeping each parameter's
    shape and strides, e.g. channels_last conv weights) and pre-creates matching views of
    a flat gradient buffer that the hand-written backward writes into directly."""

    def __init__(self, params):
        params = [p for p in params]
        if not params:
            raise ValueError("no parameters")
        dev = params[0].device
        for p in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("ParamArena needs fp32 parameters on one device")
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            self.offsets.append(off)
            off += _align4(p.numel())
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                view = torch.as_strided(self.data, p.shape, p.stride(), o)
                view.copy_(p.data)
                p.data = view
                p._iswm_grad_view = torch.as_strided(self.grad, p.shape, p.stride(), o)
                p.grad = None

    def view_of(self, flat, p_index):
        p, o = self.params[p_index], self.offsets[p_index]
        return torch.as_strided(flat, p.shape, p.stride(), o)

    def zero_grad(self):
        if self.grad.is_cuda:
            ops.call("iswm_fill_zero", ops._p(self.grad), self.grad.numel() * 4, ops._stream())
        else:
            self.grad.zero_()
        for p in self.params:
            p.grad = None


def arena_of(params):
    """Reuse an existing arena if these parameters were already flattened together."""
    params = list(params)
    a = getattr(params[0], "_iswm_arena", None)
    if a is not None and len(a.params) == len(params) and all(x is y for x, y in zip(a.params, params)):
        return a
    a = ParamArena(params)
    for p in params:
        p._iswm_arena = a
    return a


class _FusedBase(torch.optim.Optimizer):
    def _setup(self):
        if len(self.param_groups) != 1:
            raise NotImplementedError("one parameter group (as train.py:424 passes model.parameters())")
        self.arena = arena_of(self.param_groups[0]["params"])
        dev = self.arena.data.device
        # step hyper-parameters (lr, Adam bias corrections) reach the kernel through device memory so that the launch stays
        # capturable; the pinned staging area is a RING of slots, each guarded by the event of the copy that last read it:
        # the DMA reads pinned memory when the GPU gets to it, not at enqueue, and the host runs steps ahead of the GPU
        self._nslots = 8
        self._hyper_host = torch.zeros((self._nslots, 4), dtype=torch.float32)
        if dev.type == "cuda":
            self._hyper_host = self._hyper_host.pin_memory()
        self._hyper_events = [None] * self._nslots
        self._hyper_slot = 0
        self._hyper_dev = torch.zeros(4, dtype=torch.float32, device=dev)

    def _push_hyper(self, *vals):
        i = self._hyper_slot
        self._hyper_slot = (i + 1) % self._nslots
        ev = self._hyper_events[i]
        if ev is not None:
            ev.synchronize()                   # the copy that read this slot nslots steps ago has run
        for k, v in enumerate(vals):
            self._hyper_host[i, k] = v
        self._hyper_dev.copy_(self._hyper_host[i], non_blocking=True)
        if self._hyper_dev.is_cuda:
            ev = ev or torch.cuda.Event()
            ev.record()
            self._hyper_events[i] = ev

    def _segments(self):
        """[(start, end)) element ranges of the arena to update: torch.optim skips parameters without a gradient (frozen
        layers keep their weights, momentum and moments untouched), so only runs of parameters that have one are stepped"""
        a = self.arena
        segs, start = [], None
        for i, p in enumerate(a.params):
            live = p.grad is not None
            if live and start is None:
                start = a.offsets[i]
            if not live and start is not None:
                segs.append((start, a.offsets[i]))
                start = None
        if start is not None:
            segs.append((start, a.numel))
        return segs

    def load_state_dict(self, state_dict):
        """accepts the state of the stock torch optimizer of the same family (the reference's checkpoints,
        train.py:567-582): its param_groups lack this class's own keys"""
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            for k, v in self.defaults.items():
                g.setdefault(k, v)

    def zero_grad(self, set_to_none=True):
        self.arena.zero_grad()

    def _grads_ready(self):
        for p in self.arena.params:
            if p.grad is None:
                continue            # arena.grad was zeroed; a parameter without a gradient keeps g == 0
            if p.grad.data_ptr() != p._iswm_grad_view.data_ptr():
                p._iswm_grad_view.copy_(p.grad)   # gradient produced outside the arena (plain autograd)


class FusedSGD(_FusedBase):
    """torch.optim.SGD semantics (g += wd*p; buf = mu*buf + g; p -= lr*(g + mu*buf) if nesterov)."""

    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        if dampening != 0.0:
            raise NotImplementedError("dampening")
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        defaults = dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                        nesterov=nesterov)
        super().__init__(params, defaults)
        self._setup()
        self._buf = torch.zeros_like(self.arena.data)
        for i, p in enumerate(self.arena.params):
            self.state[p]["momentum_buffer"] = self.arena.view_of(self._buf, i)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._grads_ready()
        self._push_hyper(g["lr"])
        for a, b in self._segments():
            ops.sgd_step(self.arena.data[a:b], self.arena.grad[a:b], self._buf[a:b], self._hyper_dev, g["momentum"],
                         g["weight_decay"], g["nesterov"])
        ops.weights_changed()
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for i, p in enumerate(self.arena.params):
            view = self.arena.view_of(self._buf, i)
            mb = self.state[p].get("momentum_buffer")
            if mb is not None and mb.data_ptr() != view.data_ptr():
                view.copy_(mb)
            self.state[p]["momentum_buffer"] = view


class FusedAdam(_FusedBase):
    """torch.optim.Adam (decoupled=False) / AdamW (decoupled=True), amsgrad off."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled)
        super().__init__(params, defaults)
        self._setup()
        self._m = torch.zeros_like(self.arena.data)
        self._v = torch.zeros_like(self.arena.data)
        self._t = 0
        for i, p in enumerate(self.arena.params):
            self.state[p]["step"] = torch.tensor(0.0)
            self.state[p]["exp_avg"] = self.arena.view_of(self._m, i)
            self.state[p]["exp_avg_sq"] = self.arena.view_of(self._v, i)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        self._grads_ready()
        segs = self._segments()
        if not segs:
            return loss             # nothing has a gradient: torch's Adam leaves every step counter alone too
        self._t += 1
        b1, b2 = g["betas"]
        self._push_hyper(g["lr"], 1.0 - b1 ** self._t, 1.0 - b2 ** self._t)
        for a, b in segs:
            ops.adam_step(self.arena.data[a:b], self.arena.grad[a:b], self._m[a:b], self._v[a:b], self._hyper_dev, b1, b2,
                          g["eps"], g["weight_decay"], g["decoupled"])
        ops.weights_changed()
        for p in self.arena.params:
            if p.grad is not None:          # a frozen parameter's counter stays where it was (as in torch.optim.Adam)
                self.state[p]["step"] += 1
        return loss

    def load_state_dict(self, state_dict):
        """a stock torch.optim.Adam(W) checkpoint has NO state entry for a parameter that never received a gradient (a frozen
        backbone): such parameters keep zero moments and a zero counter, and the arena views are reinstalled for all"""
        super().load_state_dict(state_dict)
        t_max = 0
        for i, p in enumerate(self.arena.params):
            st = self.state[p]
            for key, flat in (("exp_avg", self._m), ("exp_avg_sq", self._v)):
                view = self.arena.view_of(flat, i)
                t = st.get(key)
                if t is None:
                    view.zero_()
                elif t.data_ptr() != view.data_ptr():
                    view.copy_(t)
                st[key] = view
            step = st.get("step")
            if step is None:
                st["step"] = torch.tensor(0.0)
            else:
                t_max = max(t_max, int(step))
        self._t = t_max             # one bias correction for the whole arena: the stepped parameters share a counter


class FusedAdamW(FusedAdam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=True)


def cosine_lr(base_lr, t, t_max, eta_min):
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / t_max)) / 2
