"""Execution layer of the HIP backend: modules with a hand-written forward AND
backward over NHWC activations.

The reference is a tree of stock ``torch.nn`` modules driven by autograd
(network/utils.py:16-25).  Here every module keeps the reference's name, children
and parameters (so ``state_dict`` keys match) but implements

    fwd(x, save)  -> y          NHWC in, NHWC out; ``save`` keeps what bwd needs
    bwd(dy, sink) -> dx         hand-written backward; parameter gradients are
                                written straight into ``p.grad`` via ``sink``

and the whole model is ONE node in torch's autograd graph (``_Bridge``), so
``loss.backward()`` / ``optimizer.step()`` in train.py work unchanged while no
torch op ever touches an activation.  All compute is libiswm_hip.so kernels.
"""
import os

import torch
import torch.nn as nn

from .. import ops


# Test hook: when set to a dict, cba_fwd records {bn_module: (out > 0)} for every fused ReLU so a
# parity test can hand the CPU oracle the exact sign pattern this path used (tests/test_hip_modules.py).
MASK_RECORDER = None
# ... and {MaxPool2d module: uint8 NHWC tap index (kh * 3 + kw) chosen per output element}: two fp32 implementations pick different
# elements of a near-tie, which re-routes that window's gradient (same kind of discontinuity as a flipped ReLU)
POOL_RECORDER = None


def pad4(c):
    return (c + 3) // 4 * 4


def pad_cin(c):
    """channel count a conv's INPUT buffer is padded to: groups of 4 (16-byte loads) always; wide inputs
    that are not a multiple of the K chunk (the decoder's 304 = 48 + 256) go up to the next multiple of 32
    so the tap-uniform / bf16x6 kernels apply -- 5 % more MACs on zero channels buys a 1.5x faster kernel"""
    if c % 32 == 0 or c < 128:
        return pad4(c)
    return (c + 31) // 32 * 32


def dense_flat(t):
    """1-D view over the dense memory of a parameter-shaped tensor (contiguous or
    channels_last)."""
    if t.dim() == 4 and not t.is_contiguous():
        v = t.permute(0, 2, 3, 1)
        assert v.is_contiguous(), "parameter is neither contiguous nor channels_last"
        return v.reshape(-1)
    assert t.is_contiguous()
    return t.reshape(-1)


class GradSink:
    """Where parameter gradients go.  ``target(p)`` returns a tensor shaped/strided
    like ``p`` for a kernel to overwrite; ``done(p)`` folds it into ``p.grad``
    (a no-op on the usual zero_grad(set_to_none=True) path) and fires the
    data-parallel ready hook."""

    def __init__(self, on_ready=None):
        self.on_ready = on_ready
        self._pending = {}

    def target(self, p):
        if p.grad is None:
            view = getattr(p, "_iswm_grad_view", None)
            p.grad = view if view is not None else torch.empty_like(p)
            return p.grad
        tmp = torch.empty_like(p)
        self._pending[id(p)] = tmp
        return tmp

    def done(self, p):
        tmp = self._pending.pop(id(p), None)
        if tmp is not None:
            ops.add_inplace(dense_flat(p.grad), dense_flat(tmp))
        if self.on_ready is not None:
            self.on_ready(p)


class HipModule(nn.Module):
    """Base: public ``forward`` takes/returns NCHW like the reference module and
    bridges to fwd/bwd."""

    def fwd(self, x, save):
        raise NotImplementedError

    def bwd(self, dy, sink):
        raise NotImplementedError

    def out_channels_of(self, cin):
        """logical channel count of the output (to strip channel padding)."""
        return None

    def forward(self, x):
        return run_module(self, x)


class _Bridge(torch.autograd.Function):
    """One autograd node for a whole HipModule call (NCHW tensors outside)."""

    @staticmethod
    def forward(ctx, mod, cout, x, *params):
        xh = ops.nchw_to_nhwc(x)
        yh = mod.fwd(xh, True)
        ctx.mod, ctx.cin, ctx.need_dx = mod, x.shape[1], x.requires_grad
        return ops.nhwc_to_nchw(yh, cout or yh.shape[3])

    @staticmethod
    def backward(ctx, dy):
        mod = ctx.mod
        dyh = ops.nchw_to_nhwc(dy.contiguous())
        sink = GradSink(getattr(mod, "_iswm_on_ready", None))
        dxh = mod.bwd(dyh, sink)
        dx = ops.nhwc_to_nchw(dxh, ctx.cin) if (ctx.need_dx and dxh is not None) else None
        return (None, None, dx) + (None,) * (len(ctx.needs_input_grad) - 3)


def run_module(mod, x):
    if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
        raise ValueError("iswm_amd modules take a 4-D fp32 CUDA NCHW tensor (no CPU fallback); got %s" %
                         (x.shape if torch.is_tensor(x) else type(x),))
    params = [p for p in mod.parameters() if p.requires_grad]
    cout = mod.out_channels_of(x.shape[1])
    if torch.is_grad_enabled() and (x.requires_grad or params):
        return _Bridge.apply(mod, cout, x, *params)
    yh = mod.fwd(ops.nchw_to_nhwc(x), False)
    return ops.nhwc_to_nchw(yh, cout or yh.shape[3])


# ---------------------------------------------------------------------------------------
# leaf modules: same classes/keys as torch.nn so isinstance checks in the reference's
# _init_weight (network/_deeplab.py:63-69) and state_dict round-trips keep working
# ---------------------------------------------------------------------------------------
class Conv2d(HipModule, nn.Conv2d):
    """nn.Conv2d(groups=1) whose weight lives in channels_last memory format, i.e.
    physically OHWI -- the layout the implicit-GEMM kernels read directly."""

    def __init__(self, *args, **kwargs):
        nn.Conv2d.__init__(self, *args, **kwargs)
        if self.groups != 1 or self.padding_mode != "zeros":
            raise NotImplementedError("HIP conv supports groups=1, zero padding")
        if self.kernel_size[0] != self.kernel_size[1] or self.dilation[0] != self.dilation[1] or \
                self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1]:
            raise NotImplementedError("HIP conv supports square kernels/strides/dilations")
        with torch.no_grad():
            self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        self._saved = None

    # -- operand views -----------------------------------------------------------------
    @property
    def cin_p(self):
        return pad_cin(self.in_channels)

    @property
    def cout_p(self):
        return pad4(self.out_channels)

    def needs_pack(self):
        return self.cin_p != self.in_channels or self.cout_p != self.out_channels

    def ohwi(self):
        """OHWI weight for the kernels; zero-padded copy when Cin/Cout % 4 != 0
        (the 3-channel stem, the num_classes-wide classifier)."""
        w = self.weight
        v = w.permute(0, 2, 3, 1)
        if not self.needs_pack():
            return v if v.is_contiguous() else v.contiguous()
        return ops.pad_weights(w, self.cout_p, self.cin_p)

    def packed2(self, kind):
        """this weight pre-packed for the planes kernels (csrc/conv_mfma_pl2.hip) by the WeightPacker, or None"""
        e = getattr(self, "_iswm_wpk", None)
        if e is None or not e["live"] or e["epoch"] != ops.WEIGHTS_EPOCH or e["version"] != self.weight._version or \
                e["ptr"] != self.weight.data_ptr():
            return None
        return e["buf"].get(kind + 2)

    def packed(self, kind):
        """this weight pre-packed for the bf16x6 forward (0) / data-gradient (1) kernel by the model's WeightPacker,
        or None (the kernel wrappers then pack it themselves).  Valid only inside the forward/backward window it was
        made for and only while the parameter is untouched."""
        e = getattr(self, "_iswm_wpk", None)
        if e is None or not e["live"] or e["epoch"] != ops.WEIGHTS_EPOCH or e["version"] != self.weight._version or \
                e["ptr"] != self.weight.data_ptr():
            return None
        return e["buf"][kind]

    def bias_p(self):
        if self.bias is None:
            return None
        if self.cout_p == self.out_channels:
            return self.bias
        return ops.pad_weights(self.bias.view(-1, 1, 1, 1), self.cout_p, 1).view(-1)

    def geometry(self, x):
        if x.shape[3] != self.cin_p:
            raise ValueError("conv expects %d (padded) input channels, got %d" % (self.cin_p, x.shape[3]))
        g = ops.ConvGeom(x, self.cout_p, self.kernel_size[0], self.kernel_size[1], self.stride[0],
                         self.padding[0], self.dilation[0])
        g.alg_cin, g.alg_cout = self.in_channels, self.out_channels
        return g

    def write_wgrad(self, x, dy, g, sink):
        """weight gradient straight into p.grad's memory when it is OHWI-dense."""
        p = self.weight
        if not p.requires_grad:
            return
        buf = sink.target(p)
        v = buf.permute(0, 2, 3, 1)
        if not self.needs_pack() and v.is_contiguous():
            ops.conv2d_wgrad(x, dy, g, v)
        else:
            ops.unpad_weights(ops.conv2d_wgrad(x, dy, g), buf)
        sink.done(p)

    # -- standalone conv (+bias), e.g. the final 1x1 classifier ---------------------------
    def fwd(self, x, save, out=None):
        g = self.geometry(x)
        y, _, _ = ops.conv2d_fwd(x, self.ohwi(), g, bias=self.bias_p(), out=out, wpk=self.packed(0), wpk2=self.packed2(0))
        self._saved = (x, g) if save else None
        return y

    def bwd(self, dy, sink, need_dx=True, dx=None, accumulate=False):
        x, g = self._saved
        self._saved = None
        if self.bias is not None and self.bias.requires_grad:
            ops.unpad_weights(ops.colsum(dy).view(-1, 1, 1, 1), sink.target(self.bias).view(-1, 1, 1, 1))
            sink.done(self.bias)
        self.write_wgrad(x, dy, g, sink)
        if not need_dx:
            return None
        return ops.conv2d_dgrad(dy, self.ohwi(), g, tuple(x.shape), dx, accumulate, wpk=self.packed(1), wpk2=self.packed2(1))

    def out_channels_of(self, cin):
        return self.out_channels


class DepthwiseConv2d(HipModule, nn.Conv2d):
    """nn.Conv2d(C, C, k, groups=C): the depthwise half of AtrousSeparableConvolution (network/_deeplab.py:103).
    The parameter keeps torch's [C,1,KH,KW] layout; the activation may be a zero-padded buffer wider than C."""

    def __init__(self, *args, **kwargs):
        nn.Conv2d.__init__(self, *args, **kwargs)
        if self.groups != self.in_channels or self.out_channels != self.in_channels or self.padding_mode != "zeros":
            raise NotImplementedError("HIP depthwise conv needs groups == in_channels == out_channels, zero padding")
        if self.kernel_size[0] != self.kernel_size[1] or self.dilation[0] != self.dilation[1] or \
                self.stride[0] != self.stride[1] or self.padding[0] != self.padding[1]:
            raise NotImplementedError("HIP depthwise conv supports square kernels/strides/dilations")
        self._saved = None

    def geometry(self, x):
        c = x.shape[3]
        if c < self.in_channels or c % 4 != 0:
            raise ValueError("depthwise conv over %d channels got a %d-channel buffer" % (self.in_channels, c))
        return ops.ConvGeom(x, c, self.kernel_size[0], self.kernel_size[1], self.stride[0], self.padding[0],
                            self.dilation[0])

    def fwd(self, x, save, out=None):
        g = self.geometry(x)
        y = ops.dwconv2d_fwd(x, self.weight.contiguous(), g, self.bias, out)
        self._saved = (x, g) if save else None
        return y

    def bwd(self, dy, sink, need_dx=True, dx=None, accumulate=False):
        x, g = self._saved
        self._saved = None
        if self.bias is not None and self.bias.requires_grad:
            ops.unpad_weights(ops.colsum(dy).view(-1, 1, 1, 1), sink.target(self.bias).view(-1, 1, 1, 1))
            sink.done(self.bias)
        if self.weight.requires_grad:
            buf = sink.target(self.weight)
            if buf.is_contiguous():
                ops.dwconv2d_wgrad(x, dy, g, self.in_channels, buf)
            else:
                buf.copy_(ops.dwconv2d_wgrad(x, dy, g, self.in_channels))
            sink.done(self.weight)
        if not need_dx:
            return None
        return ops.dwconv2d_dgrad(dy, self.weight.contiguous(), g, tuple(x.shape), dx, accumulate)

    def out_channels_of(self, cin):
        return self.out_channels


class BatchNorm2d(nn.BatchNorm2d):
    pass


class ReLU(nn.ReLU):
    pass


class ReLU6(nn.ReLU6):
    """nn.ReLU6 (MobileNetV2's activation): fused into the BatchNorm pass like ReLU (clamp to [0, 6])"""
    pass


class Dropout(HipModule, nn.Dropout):
    """nn.Dropout(p) (network/_deeplab.py:165) with a Philox counter mask."""

    def __init__(self, *a, **k):
        nn.Dropout.__init__(self, *a, **k)
        self._calls = 0
        self._mask = None

    def fwd(self, x, save):
        if not self.training or self.p == 0.0:
            self._mask = None
            return x
        self._calls += 1
        y, mask = ops.dropout_fwd(x, self.p, torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, self._calls)
        self._mask = mask if save else None
        return y

    def bwd(self, dy, sink):
        if self._mask is None:
            return dy
        dx = ops.dropout_bwd(dy.contiguous(), self._mask, self.p)
        self._mask = None
        return dx


# ---------------------------------------------------------------------------------------
# conv -> BatchNorm -> (+residual) -> ReLU, the unit every stage of the net is made of
# ---------------------------------------------------------------------------------------
def cba_fwd(conv, bn, relu, x, save, residual=None, out=None, out_fmt=None):
    """Returns (out, ctx).  Training-mode BN statistics come from the conv epilogue's
    per-tile partial sums (no extra pass over y).  The output is written pre-split (ops.Planes) when a convolution
    will consume it: `out` (a buffer slice) decides, else out_fmt ("planes" / "f32"), else the channel count."""
    sep = None
    if isinstance(conv, SeparableBase):         # depthwise first, then the pointwise conv carries the fused BN
        sep, conv = conv, conv.body[1]
        x = sep.body[0].fwd(x, save)
    training = bn.training
    if bn.momentum is None or not bn.track_running_stats or not bn.affine:
        raise NotImplementedError("HIP BatchNorm2d supports affine=True, momentum!=None, running stats")
    if isinstance(conv, DepthwiseConv2d):       # depthwise conv -> BN (MobileNetV2): statistics from a column pass
        g = conv.geometry(x)
        y = conv.fwd(x, save)
        partials, tiles = None, (0, 0)
        if training:
            partials, nt, tr = ops.colstat(y)
            tiles = (nt, tr)
        return _cba_finish(conv, bn, relu, x, y, g, partials, tiles, training, save, residual, out, sep, True, out_fmt)
    g = conv.geometry(x)
    if conv.bias is None:
        y, partials, tiles = ops.conv2d_fwd(x, conv.ohwi(), g, want_stats=training, wpk=conv.packed(0), wpk2=conv.packed2(0))
    else:
        # a biased conv in front of a BatchNorm (only reachable through convert_to_separable_conv on a biased
        # conv): the fused epilogue statistics do not include the bias, so take them in a separate column pass
        y, _, _ = ops.conv2d_fwd(x, conv.ohwi(), g, bias=conv.bias_p())
        partials, tiles = None, (0, 0)
        if training:
            partials, nt, tr = ops.colstat(y)
            tiles = (nt, tr)
    return _cba_finish(conv, bn, relu, x, y, g, partials, tiles, training, save, residual, out, sep, False, out_fmt)


def _cba_finish(conv, bn, relu, x, y, g, partials, tiles, training, save, residual, out, sep, dw, out_fmt=None):
    if training:
        count = y.shape[0] * y.shape[1] * y.shape[2]
        if count <= 1:
            # same contract as torch (network/_deeplab.py:130-141 needs batch >= 2)
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" %
                             (tuple(y.shape),))
        coef = ops.bn_finalize(partials, tiles[0], count, tiles[1], bn.weight, bn.bias, bn.running_mean, bn.running_var,
                               bn.momentum, bn.eps)
        if not getattr(bn, "_iswm_nbt_fused", False):
            bn.num_batches_tracked.add_(1)     # models built by modeling.* bump all counters in ONE op
    else:
        coef = ops.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    if out_fmt is None:
        want_planes = ops.planes_on() and y.shape[3] % 64 == 0
    else:
        want_planes = out_fmt == "planes" and ops.planes_on() and y.shape[3] % 64 == 0
    o = ops.bn_apply(y, coef, relu, residual, out, planes=want_planes)
    if MASK_RECORDER is not None and relu:
        of = ops.as_f32(o)
        MASK_RECORDER[bn] = (of > 0) & (of < 6) if relu == 6 and relu is not True else (of > 0)
    ctx = None
    if save:
        ctx = dict(x=x, y=y, out=o, coef=coef, g=g, relu=relu, training=training, res=residual is not None, sep=sep, dw=dw)
    return o, ctx


_BN3_FUSE = os.environ.get("ISWM_BN3_FUSE", "1") != "0"      # tuning switch: 0 = residual stages reduce themselves


def _bn_stats_request(up):
    """`up` = ctx of the stage that produced this stage's input, when ITS BatchNorm backward is the only consumer of the dx this
    stage returns: the planes data gradient then takes that backward's reduction pass in its epilogue (ops.BnStats)."""
    if up is None or not ops.planes_on() or up.get("bn_stats") is not None:
        return None
    if ops._relu_code(up["relu"]) not in (0, 1) or not ops._BN_MASK_FROM_Y:
        return None
    if up.get("res"):
        # a residual stage (bn3 + identity + ReLU): the pattern is read from its saved output planes and the data gradient
        # stores dx masked -- see ops.BnStats.mask
        if not (_BN3_FUSE and ops._relu_code(up["relu"]) == 1 and ops.is_planes(up["out"]) and up["training"]):
            return None
        return ops.BnStats(up["y"], up["coef"], up["relu"], mask=up["out"])
    return ops.BnStats(up["y"], up["coef"], up["relu"])


def cba_bwd_bn(conv, bn, ctx, dout, sink, dy_out=None, need_dx=True):
    """the BatchNorm (+ activation) backward of a stage: parameter gradients into the sink, returns (dy, dres) with dy the
    gradient of the raw conv output -- pre-split when the conv's gradient kernels take planes; dy_out: write it there (a Planes
    slice of a wider buffer: the fused ASPP data gradient reads the four branches' gradients from one tensor)"""
    y, o = ctx["y"], ctx["out"]
    sep = ctx.get("sep")
    if sep is not None:
        conv = sep.body[1]
    gw, gb = bn.weight, bn.bias
    dgamma = sink.target(gw) if gw.requires_grad else torch.empty_like(gw)
    dbeta = sink.target(gb) if gb.requires_grad else torch.empty_like(gb)
    # planes only when a planes kernel will read them: the data gradient, or the weight gradient of a pre-split input (the stem
    # has neither: its gradient stays fp32 instead of being split here and joined again for the weight gradient)
    dyp = (not ctx.get("dw")) and ops.planes_conv_ok(conv.cin_p, conv.cout_p, 1) and (need_dx or ops.is_planes(ctx["x"]))
    st = ctx.pop("bn_stats", None)
    if st is not None and st.partials is not None and st.masked:
        dout = ops.as_f32(dout)
        dy, _ = ops.bn_backward(dout, None, y, ctx["coef"], gw, False, ctx["training"], dgamma, dbeta, want_dres=False,
                                dy=dy_out, dy_planes=dyp, stats=st)
        dres = dout if ctx["res"] else None
    else:
        dy, dres = ops.bn_backward(ops.as_f32(dout), o if ctx["relu"] else None, y, ctx["coef"], gw, ctx["relu"], ctx["training"],
                                   dgamma, dbeta, want_dres=ctx["res"], dy=dy_out, dy_planes=dyp, stats=st)
    if gw.requires_grad:
        sink.done(gw)
    if gb.requires_grad:
        sink.done(gb)
    return dy, dres


def cba_bwd(conv, bn, ctx, dout, sink, need_dx=True, dx=None, accumulate=False, up=None):
    """Returns (dx, dres): dres is the gradient of the residual input (if any).  up: see _bn_stats_request."""
    x, g = ctx["x"], ctx["g"]
    sep = ctx.get("sep")
    dy, dres = cba_bwd_bn(conv, bn, ctx, dout, sink, need_dx=need_dx or sep is not None)
    if sep is not None:
        conv = sep.body[1]
    if ctx.get("dw"):
        return conv.bwd(dy, sink, need_dx, dx, accumulate), dres
    conv.write_wgrad(x, dy, g, sink)
    if conv.bias is not None and conv.bias.requires_grad:
        ops.unpad_weights(ops.colsum(dy).view(-1, 1, 1, 1), sink.target(conv.bias).view(-1, 1, 1, 1))
        sink.done(conv.bias)
    if sep is not None:
        dmid = ops.conv2d_dgrad(dy, conv.ohwi(), g, tuple(x.shape), wpk=conv.packed(1), wpk2=conv.packed2(1))
        return sep.body[0].bwd(dmid, sink, need_dx, dx, accumulate), dres
    if need_dx:
        req = _bn_stats_request(up) if (up is not None and tuple(up["y"].shape) == tuple(x.shape)) else None
        dx = ops.conv2d_dgrad(dy, conv.ohwi(), g, tuple(x.shape), dx, accumulate, wpk=conv.packed(1), wpk2=conv.packed2(1),
                              bn_stats=req)
        if req is not None and req.partials is not None:
            up["bn_stats"] = req
    else:
        dx = None
    return dx, dres


def fuse_batch_counters(root):
    """Re-home every BatchNorm's ``num_batches_tracked`` into one int64 tensor so a training forward
    bumps all of them with a single op (113 tiny launches per step for ResNet-101 otherwise).  The
    buffers stay registered under their usual state_dict keys (they become views)."""
    bns = [m for m in root.modules() if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
    if not bns:
        return None
    flat = torch.stack([m.num_batches_tracked.detach().reshape(()) for m in bns]).contiguous()
    for i, m in enumerate(bns):
        m._buffers["num_batches_tracked"] = flat[i]
        m._iswm_nbt_fused = True
    return flat


_BATCH_PACK = os.environ.get("ISWM_BATCH_PACK", "1") != "0"     # 0: every conv call packs its own weight (tuning switch)


class WeightPacker(object):
    """Packs the weights of every (unpadded) Conv2d under `root` for the bf16x6 kernels in ONE launch per forward
    pass -- the exact 3-way split in MFMA fragment order, csrc/conv_mfma_x6.hip k_pack_weights_batch -- instead of
    two tiny launches per conv per step.  Buffers and the device job table persist; they are rebuilt when a
    parameter moved.  `begin()` at the start of a model forward makes the buffers live, `end()` retires them."""

    def __init__(self, root):
        self.convs = [m for m in root.modules() if isinstance(m, Conv2d) and not m.needs_pack()]
        self.key, self.jobs, self.njobs, self.blocks, self.entries = None, None, 0, 0, []

    def _build(self, dev):
        import ctypes
        from .. import _lib
        lib = _lib.load()
        jobs, entries, first = [], [], 0
        keep = []
        for m in self.convs:
            w = m.weight
            v = w.permute(0, 2, 3, 1)
            if not (w.is_cuda and v.is_contiguous()):
                continue
            taps = m.kernel_size[0] * m.kernel_size[1]
            bufs = {}
            # with pre-split activations the planes kernels take the conv (kinds 2 / 3); the fp32-input fragment order
            # (kinds 0 / 1) is packed only for the direction the planes kernels do not cover
            kinds = []
            for kind in (0, 1):
                if ops.planes_on() and lib.iswm_packed_weight_bytes(m.out_channels, taps, m.in_channels, kind + 2):
                    kinds.append(kind + 2)
                else:
                    kinds.append(kind)
            for kind in kinds:
                nb = lib.iswm_packed_weight_bytes(m.out_channels, taps, m.in_channels, kind)
                if nb == 0:
                    continue
                buf = torch.empty((nb // 4,), dtype=torch.float32, device=dev)
                bufs[kind] = buf
                jobs.append((w.data_ptr(), buf.data_ptr(), m.out_channels, taps, m.in_channels, kind, first, 0))
                first += lib.iswm_pack_job_blocks(m.out_channels, taps, m.in_channels, kind)
            if bufs:
                e = dict(live=False, epoch=-1, version=-1, ptr=w.data_ptr(), buf={k: bufs.get(k) for k in range(4)})
                m._iswm_wpk = e
                entries.append((m, e))
                keep.append(bufs)

        class _Job(ctypes.Structure):
            _fields_ = [("w", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("Cout", ctypes.c_int), ("taps", ctypes.c_int),
                        ("Cin", ctypes.c_int), ("kind", ctypes.c_int), ("first_block", ctypes.c_int),
                        ("reserved", ctypes.c_int)]
        arr = (_Job * max(1, len(jobs)))()
        for i, j in enumerate(jobs):
            arr[i] = _Job(*j)
        self.jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev) if jobs else None
        self.njobs, self.blocks, self.entries = len(jobs), first, entries

    def begin(self):
        from .. import _lib
        math = _lib.load().iswm_get_conv_math()
        if math < 1 or not ops._USE_PACKED or not self.convs or not _BATCH_PACK:
            return
        key = (math, ops.planes_on()) + tuple(m.weight.data_ptr() for m in self.convs)      # bf16x6 packs 3 planes, bf16 one
        if key != self.key:
            self._build(self.convs[0].weight.device)
            self.key = key
        if self.njobs == 0:
            return
        ops.call("iswm_pack_weights_batch", ops._p(self.jobs), self.njobs, self.blocks, ops._stream())
        for m, e in self.entries:
            e["live"], e["epoch"], e["version"] = True, ops.WEIGHTS_EPOCH, m.weight._version

    def end(self):
        for _, e in self.entries:
            e["live"] = False


_CLS_FUSE = os.environ.get("ISWM_CLS_FUSE", "1") != "0"      # tuning switch: 0 = the classifier as a conv of its own


def cls_fusable(conv, bn, relu, cls):
    """[Conv2d(.., 256, k) -> BatchNorm2d -> ReLU] followed by the 1x1 classifier Conv2d(256, num_classes <= 4, 1): the tail of both
    DeepLab heads (network/_deeplab.py:44-52, 84-90).  The classifier is then folded into the stage's BatchNorm passes
    (csrc/bn_classify.hip).  Not while a test records ReLU patterns (it needs the stage's stored output)."""
    return (_CLS_FUSE and MASK_RECORDER is None and relu is True and type(conv) is Conv2d and conv.out_channels == 256 and
            conv.bias is None and type(cls) is Conv2d and cls.in_channels == 256 and cls.out_channels <= 4 and
            tuple(cls.kernel_size) == (1, 1) and tuple(cls.stride) == (1, 1) and tuple(cls.padding) == (0, 0) and
            bn.momentum is not None and bn.track_running_stats and bn.affine)


def cba_cls_fwd(conv, bn, cls, x, save):
    """conv -> BatchNorm -> ReLU -> 1x1 classifier with the activation never stored: returns (logits [N,H,W,4], ctx)"""
    g = conv.geometry(x)
    training = bn.training
    y, partials, tiles = ops.conv2d_fwd(x, conv.ohwi(), g, want_stats=training, wpk=conv.packed(0), wpk2=conv.packed2(0))
    if training:
        count = y.shape[0] * y.shape[1] * y.shape[2]
        if count <= 1:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(y.shape),))
        coef = ops.bn_finalize(partials, tiles[0], count, tiles[1], bn.weight, bn.bias, bn.running_mean, bn.running_var,
                               bn.momentum, bn.eps)
        if not getattr(bn, "_iswm_nbt_fused", False):
            bn.num_batches_tracked.add_(1)
    else:
        coef = ops.bn_eval_coeffs(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    wc4 = ops.pad_weights(cls.weight, 4, 256).view(4, 256)
    b4 = None if cls.bias is None else ops.pad_weights(cls.bias.view(-1, 1, 1, 1), 4, 1).view(-1)
    logits = ops.bn_apply_classify(y, coef, wc4, b4)
    ctx = dict(x=x, y=y, coef=coef, g=g, training=training, wc4=wc4, cls=cls, cls_fused=True) if save else None
    return logits, ctx


def cba_cls_bwd(conv, bn, ctx, dlogit, sink, need_dx=True, dx=None, accumulate=False):
    """backward of cba_cls_fwd: classifier bias / weight gradients, the stage's BatchNorm backward fed by dlogit, then the
    conv's weight and data gradients as in cba_bwd"""
    cls, x, y, g = ctx["cls"], ctx["x"], ctx["y"], ctx["g"]
    dlogit = ops.as_f32(dlogit)
    if cls.bias is not None and cls.bias.requires_grad:
        ops.unpad_weights(ops.colsum(dlogit).view(-1, 1, 1, 1), sink.target(cls.bias).view(-1, 1, 1, 1))
        sink.done(cls.bias)
    gw, gb = bn.weight, bn.bias
    dgamma = sink.target(gw) if gw.requires_grad else torch.empty_like(gw)
    dbeta = sink.target(gb) if gb.requires_grad else torch.empty_like(gb)
    dyp = ops.planes_conv_ok(conv.cin_p, conv.cout_p, 1)
    dy, dwc4 = ops.bn_backward_classify(dlogit, ctx["wc4"], y, ctx["coef"], gw, ctx["training"], dgamma, dbeta, dyp)
    if gw.requires_grad:
        sink.done(gw)
    if gb.requires_grad:
        sink.done(gb)
    if cls.weight.requires_grad:
        ops.unpad_weights(dwc4.view(4, 1, 1, 256), sink.target(cls.weight))
        sink.done(cls.weight)
    conv.write_wgrad(x, dy, g, sink)
    if not need_dx:
        return None
    return ops.conv2d_dgrad(dy, conv.ohwi(), g, tuple(x.shape), dx, accumulate, wpk=conv.packed(1), wpk2=conv.packed2(1))


class SeparableBase(HipModule):
    """marker base of _deeplab.AtrousSeparableConvolution (body = [DepthwiseConv2d, pointwise Conv2d]) so that the
    conv -> BN -> ReLU stage logic here can recognise it without importing _deeplab"""
    pass


class HipSequential(HipModule, nn.Sequential):
    """nn.Sequential whose children are grouped into fused stages:
    [Conv2d, BatchNorm2d, ReLU?] -> cba;  Conv2d alone -> conv(+bias);  Dropout;  HipModule."""

    def _stages(self):
        mods = list(self.children())
        st, i = [], 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, (Conv2d, SeparableBase, DepthwiseConv2d)) and i + 1 < len(mods) and \
                    isinstance(mods[i + 1], nn.BatchNorm2d):
                act = mods[i + 2] if i + 2 < len(mods) else None
                relu = 6 if isinstance(act, nn.ReLU6) else isinstance(act, nn.ReLU)
                st.append(("cba", m, mods[i + 1], relu))
                i += 3 if relu else 2
            elif isinstance(m, Conv2d):
                if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
                    relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
                    st.append(("cba", m, mods[i + 1], relu))
                    i += 3 if relu else 2
                else:
                    st.append(("conv", m))
                    i += 1
            elif isinstance(m, HipModule):
                st.append(("mod", m))
                i += 1
            else:
                raise NotImplementedError("no HIP implementation for %s inside a Sequential" % type(m).__name__)
        return st

    def fwd(self, x, save, out=None, out_fmt=None):
        st = self._stages()
        ctxs = []
        for k, s in enumerate(st):
            last = k == len(st) - 1
            if ctxs and ctxs[-1] == "FOLDED" and last:
                break                                   # the classifier conv was folded into the stage in front of it
            if (s[0] == "cba" and k == len(st) - 2 and st[k + 1][0] == "conv" and out is None and
                    cls_fusable(s[1], s[2], s[3], st[k + 1][1])):
                x, c = cba_cls_fwd(s[1], s[2], st[k + 1][1], x, save)
                ctxs.append(c)
                ctxs.append("FOLDED")
                continue
            if s[0] == "cba":
                # the stage's output is pre-split when the next stage is a convolution; the last stage follows `out`
                # / out_fmt (default fp32: the caller is not a conv unless it says so)
                if last:
                    fmt = out_fmt if out is None else None
                    if out is None and out_fmt is None:
                        fmt = "f32"
                else:
                    fmt = "planes" if st[k + 1][0] in ("cba", "conv") else "f32"
                x, c = cba_fwd(s[1], s[2], s[3], x, save, out=out if last else None, out_fmt=fmt)
                ctxs.append(c)
            elif s[0] == "conv":
                x = s[1].fwd(x, save, out=out if last else None)
                ctxs.append(None)
            else:
                x = s[1].fwd(x, save)
                ctxs.append(None)
        self._saved = (st, ctxs) if save else None
        return x

    def bwd(self, dy, sink, need_dx=True, dx=None, accumulate=False):
        st, ctxs = self._saved
        self._saved = None
        for k in range(len(st) - 1, -1, -1):
            s = st[k]
            first = k == 0
            if ctxs[k] == "FOLDED":
                continue
            if s[0] == "cba" and ctxs[k] is not None and ctxs[k].get("cls_fused"):
                dy = cba_cls_bwd(s[1], s[2], ctxs[k], dy, sink, need_dx or not first, dx if first else None, accumulate and first)
                continue
            if s[0] == "cba":
                dy, _ = cba_bwd(s[1], s[2], ctxs[k], dy, sink, need_dx or not first, dx if first else None,
                                accumulate and first)
            elif s[0] == "conv":
                dy = s[1].bwd(dy, sink, need_dx or not first, dx if first else None, accumulate and first)
            else:
                dy = s[1].bwd(dy, sink)
        return dy

    def out_channels_of(self, cin):
        c = None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                c = m.out_channels
        return c
