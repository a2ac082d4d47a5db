"""DeepLabV3 / V3+ heads and ASPP on the HIP execution layer -- counterpart of the
reference's network/_deeplab.py (DeepLabHeadV3Plus :33-69, DeepLabHead :71-93,
ASPPConv :121-128, ASPPPooling :130-141, ASPP :143-172): same constructor arguments,
children and state_dict keys.

MI355X-specific structure:
  * torch.cat never runs: each ASPP branch's BN/ReLU pass writes its 256 channels
    straight into its slice of one [N,H,W,1280] NHWC buffer, and the decoder's
    [48 | 256] concat is filled the same way (low-level projection + the bilinear
    upsample of the ASPP output write the two slices of a [N,H,W,304] buffer);
  * the image-pooling branch (global mean -> 1x1 -> BN -> ReLU -> bilinear from 1x1,
    i.e. a broadcast) costs one reduction pass and one broadcast into its slice;
  * in backward the five branches' input gradients accumulate in place into one
    buffer (dgrad kernels with accumulate=1) instead of five tensors plus four adds.
"""
import ctypes

import torch
import torch.nn as nn

from . import _hip
from .utils import _SimpleSegmentationModel
from .. import ops

__all__ = ["DeepLabV3"]


class DeepLabV3(_SimpleSegmentationModel):
    """DeepLabV3(backbone, classifier) -- reference :16-31."""
    pass


def _init_weight(self):
    # reference :63-69 (kaiming_normal_ default fan_in; BN gamma 1, beta 0)
    for m in self.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class ASPPConv(_hip.HipSequential):
    def __init__(self, in_channels, out_channels, dilation):
        modules = [
            _hip.Conv2d(in_channels, out_channels, 3, padding=dilation, dilation=dilation, bias=False),
            _hip.BatchNorm2d(out_channels),
            _hip.ReLU(inplace=True)
        ]
        super(ASPPConv, self).__init__(*modules)


class ASPPPooling(_hip.HipSequential):
    def __init__(self, in_channels, out_channels):
        super(ASPPPooling, self).__init__(
            nn.AdaptiveAvgPool2d(1),
            _hip.Conv2d(in_channels, out_channels, 1, bias=False),
            _hip.BatchNorm2d(out_channels),
            _hip.ReLU(inplace=True))

    def fwd(self, x, save, out=None):
        n, h, w, c = x.shape
        pooled = ops.gap_fwd(x)                                            # [N,1,1,C]
        v, ctx = _hip.cba_fwd(self[1], self[2], True, pooled, save, out_fmt="f32")   # BN over N x 1 x 1 (needs N >= 2)
        if out is None:
            out = ops.new_act(n, h, w, v.shape[3], x.device)
        ops.bcast_fwd(v, out)                                              # bilinear from 1x1 == broadcast
        self._saved = (ctx, (n, h, w, c)) if save else None
        return out

    def bwd(self, dy, sink, need_dx=True, dx=None, accumulate=False):
        ctx, shape = self._saved
        self._saved = None
        dv = ops.bcast_bwd(dy)
        dpooled, _ = _hip.cba_bwd(self[1], self[2], ctx, dv, sink)
        if dx is None:
            dx = ops.new_act(*shape, dy.device)
            accumulate = False
        return ops.gap_bwd(dpooled, dx, accumulate)


class ASPP(_hip.HipModule):
    def __init__(self, in_channels, atrous_rates):
        super(ASPP, self).__init__()
        out_channels = 256
        modules = []
        modules.append(_hip.HipSequential(
            _hip.Conv2d(in_channels, out_channels, 1, bias=False),
            _hip.BatchNorm2d(out_channels),
            _hip.ReLU(inplace=True)))
        rate1, rate2, rate3 = tuple(atrous_rates)
        modules.append(ASPPConv(in_channels, out_channels, rate1))
        modules.append(ASPPConv(in_channels, out_channels, rate2))
        modules.append(ASPPConv(in_channels, out_channels, rate3))
        modules.append(ASPPPooling(in_channels, out_channels))
        self.convs = nn.ModuleList(modules)
        self.project = _hip.HipSequential(
            _hip.Conv2d(5 * out_channels, out_channels, 1, bias=False),
            _hip.BatchNorm2d(out_channels),
            _hip.ReLU(inplace=True),
            _hip.Dropout(0.1),)
        self._saved = None

    def fwd(self, x, save, out=None):
        n, h, w, _ = x.shape
        oc = self.project[0].in_channels // len(self.convs)
        # the five branches write their slices of one buffer, pre-split for the projection conv
        if ops.planes_on() and (oc * len(self.convs)) % 64 == 0:
            cat = ops.new_planes(n, h, w, oc * len(self.convs), x.device)
        else:
            cat = ops.new_act(n, h, w, oc * len(self.convs), x.device)
        fused = self._fwd_fused(x, save, cat, oc)
        for i, conv in enumerate(self.convs):
            if fused is None or i >= len(fused):
                conv.fwd(x, save, out=cat[..., i * oc:(i + 1) * oc])
        self._saved = (tuple(x.shape), oc, fused) if save else None
        return self.project.fwd(cat, save)

    # ---- the parallel conv branches (convs[0..3]: 1x1 and the three atrous 3x3) as ONE launch each way (ops.aspp_fwd / aspp_dgrad:
    # rows sorted by in-bounds tap set, one tile table over all branches, the data gradient a single GEMM over the 28 taps)
    def _branch_convs(self):
        out = []
        for m in list(self.convs)[:-1]:
            st = m._stages() if isinstance(m, _hip.HipSequential) else []
            if len(st) != 1 or st[0][0] != "cba" or not isinstance(st[0][1], _hip.Conv2d) or not st[0][3] or st[0][3] == 6:
                return None
            c = st[0][1]
            k = c.kernel_size[0]
            if c.bias is not None or c.stride[0] != 1 or k % 2 == 0 or c.padding[0] != c.dilation[0] * (k - 1) // 2 or c.needs_pack():
                return None
            out.append((c, st[0][2]))
        if not out or any(c.in_channels != out[0][0].in_channels or c.out_channels != out[0][0].out_channels for c, _ in out):
            return None
        return out

    def _fwd_fused(self, x, save, cat, oc):
        """runs the conv branches through ops.aspp_fwd and finishes each (BatchNorm, ReLU into its slice of `cat`); returns the
        per-branch stage contexts, or None when the fused kernel does not cover this module / input"""
        if not (ops.planes_on() and ops.is_planes(cat) and x.shape[3] % 64 == 0):
            return None
        if not ops.is_planes(x):
            x = ops.split_planes(x)           # a stand-alone ASPP over an fp32 map (inside the network the backbone hands over planes)
        br = self._branch_convs()
        if br is None or len(br) > 4 or br[0][0].out_channels != oc:
            return None
        ksize = [c.kernel_size[0] for c, _ in br]
        dil = [c.dilation[0] for c, _ in br]
        training = br[0][1].training
        if any(bn.training != training for _, bn in br):
            return None
        wpks = []
        for c, _ in br:
            w2 = c.packed2(0)
            if w2 is None:
                g = c.geometry(x)
                d = g.desc(ops.pgeom(x)[4], oc)
                w2 = torch.empty((ops._pl2_bytes(d, 0) // 4,), dtype=torch.float32, device=x.t.device)
                ops.call("iswm_conv2d_pl2_pack_weights", ctypes.byref(d), 0, ops._p(c.ohwi()), ops._p(w2), ops._stream())
            wpks.append(w2)
        res = ops.aspp_fwd(x, ksize, dil, oc, wpks, training)
        if res is None:
            return None
        ys, parts, tiles = res
        ctxs = []
        for i, (c, bn) in enumerate(br):
            _, ctx = _hip._cba_finish(c, bn, True, x, ys[i], c.geometry(x), parts[i] if parts else None,
                                      (tiles, ops.ASPP_TILE_ROWS), training, save, None, cat[..., i * oc:(i + 1) * oc], None, False)
            ctxs.append(ctx)
        return ctxs

    def bwd(self, dy, sink):
        xshape, oc, fused = self._saved
        self._saved = None
        dcat = self.project.bwd(dy, sink)
        dx, first = None, 0
        if fused is not None:
            dx = self._bwd_fused(dcat, sink, oc, fused, xshape)
            first = len(fused)
        for i, conv in enumerate(self.convs):
            if i < first:
                continue
            d = conv.bwd(dcat[..., i * oc:(i + 1) * oc], sink, True, dx, dx is not None)
            dx = d if dx is None else dx
        return dx

    def _bwd_fused(self, dcat, sink, oc, ctxs, xshape):
        br = self._branch_convs()
        n, h, w, cin = xshape
        nb = len(ctxs)
        # the four BatchNorm backwards write their dy (gradient of the raw conv outputs) into ONE planes buffer
        dyc = ops.new_planes(n, h, w, nb * oc, dcat.device)
        for i, (c, bn) in enumerate(br):
            dy_i, _ = _hip.cba_bwd_bn(c, bn, ctxs[i], dcat[..., i * oc:(i + 1) * oc], sink, dy_out=dyc[..., i * oc:(i + 1) * oc])
            c.write_wgrad(ctxs[i]["x"], dy_i, ctxs[i]["g"], sink)
        wpks = []
        for c, _ in br:
            w2 = c.packed2(1)
            if w2 is None:
                g = c.geometry(ctxs[0]["x"])
                d = g.desc(cin, nb * oc)
                w2 = torch.empty((ops._pl2_bytes(d, 1) // 4,), dtype=torch.float32, device=dcat.device)
                ops.call("iswm_conv2d_pl2_pack_weights", ctypes.byref(d), 1, ops._p(c.ohwi()), ops._p(w2), ops._stream())
            wpks.append(w2)
        dx = ops.aspp_dgrad(dyc, [c.kernel_size[0] for c, _ in br], [c.dilation[0] for c, _ in br], cin, oc, wpks)
        if dx is None:                                   # (cannot happen when the forward plan existed; keep the path honest)
            for i, (c, _) in enumerate(br):
                dx = ops.conv2d_dgrad(dyc[..., i * oc:(i + 1) * oc], c.ohwi(), ctxs[i]["g"], xshape, dx, dx is not None,
                                      wpk=c.packed(1), wpk2=c.packed2(1))
        return dx

    def out_channels_of(self, cin):
        return self.project[0].out_channels


class DeepLabHeadV3Plus(_hip.HipModule):
    def __init__(self, in_channels, low_level_channels, num_classes, aspp_dilate=[6, 12, 18]):
        super(DeepLabHeadV3Plus, self).__init__()
        self.project = _hip.HipSequential(
            _hip.Conv2d(low_level_channels, 48, 1, bias=False),
            _hip.BatchNorm2d(48),
            _hip.ReLU(inplace=True),
        )
        self.aspp = ASPP(in_channels, aspp_dilate)
        self.classifier = _hip.HipSequential(
            _hip.Conv2d(304, 256, 3, padding=1, bias=False),
            _hip.BatchNorm2d(256),
            _hip.ReLU(inplace=True),
            _hip.Conv2d(256, 256, 3, padding=1, bias=False),
            _hip.BatchNorm2d(256),
            _hip.ReLU(inplace=True),
            _hip.Conv2d(256, num_classes, 1)
        )
        self.num_classes = num_classes
        self._saved = None
        self._init_weight()

    _init_weight = _init_weight

    def fwd(self, feature, save):
        low, hi = feature['low_level'], feature['out']
        n, hl, wl, _ = low.shape
        c_low = self.project[0].out_channels
        c_aspp = self.aspp.project[0].out_channels
        c_cat = c_low + c_aspp
        c_buf = self.classifier[0].cin_p            # 304 -> 320: zero channels keep the K axis a multiple of 32
        if ops.planes_on() and c_buf % 64 == 0:
            cat = ops.new_planes(n, hl, wl, c_buf, low.device)
        else:
            cat = ops.new_act(n, hl, wl, c_buf, low.device)
        if c_buf > c_cat:
            ops.zero_channels(cat, c_cat)
        self.project.fwd(low, save, out=cat[..., :c_low])
        a = self.aspp.fwd(hi, save)
        ops.bilinear_fwd(a, hl, wl, out=cat[..., c_low:c_cat])
        self._saved = (tuple(a.shape), c_low, c_cat) if save else None
        return self.classifier.fwd(cat, save)

    def bwd(self, dy, sink):
        (n, ha, wa, ca), c_low, c_cat = self._saved
        self._saved = None
        dcat = self.classifier.bwd(dy, sink)
        da = ops.bilinear_bwd(dcat[..., c_low:c_cat], ha, wa)
        dhi = self.aspp.bwd(da, sink)
        dlow = self.project.bwd(dcat[..., :c_low], sink)
        return {'low_level': dlow, 'out': dhi}

    def forward(self, feature):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _HeadBridge.apply(self, feature['low_level'], feature['out'],
                                     *[p for p in self.parameters() if p.requires_grad])
        y = self.fwd({k: ops.nchw_to_nhwc(v) for k, v in feature.items()}, False)
        return ops.nhwc_to_nchw(y, self.num_classes)


class _HeadBridge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, low, out, *params):
        ctx.head, ctx.shapes = head, (low.shape[1], out.shape[1])
        y = head.fwd({'low_level': ops.nchw_to_nhwc(low), 'out': ops.nchw_to_nhwc(out)}, True)
        return ops.nhwc_to_nchw(y, head.num_classes)

    @staticmethod
    def backward(ctx, dy):
        head = ctx.head
        cp = _hip.pad4(head.num_classes)
        d = head.bwd(ops.nchw_to_nhwc(dy.contiguous(), cp), _hip.GradSink())
        dlow = ops.nhwc_to_nchw(d['low_level'], ctx.shapes[0]) if 'low_level' in d else None
        dout = ops.nhwc_to_nchw(d['out'], ctx.shapes[1])
        return (None, dlow, dout) + (None,) * (len(ctx.needs_input_grad) - 3)


class DeepLabHead(_hip.HipModule):
    def __init__(self, in_channels, num_classes, aspp_dilate=[12, 24, 36]):
        super(DeepLabHead, self).__init__()
        self.classifier = _hip.HipSequential(
            ASPP(in_channels, aspp_dilate),
            _hip.Conv2d(256, 256, 3, padding=1, bias=False),
            _hip.BatchNorm2d(256),
            _hip.ReLU(inplace=True),
            _hip.Conv2d(256, num_classes, 1)
        )
        self.num_classes = num_classes
        self._init_weight()

    _init_weight = _init_weight

    def fwd(self, feature, save):
        return self.classifier.fwd(feature['out'], save)

    def bwd(self, dy, sink):
        return {'out': self.classifier.bwd(dy, sink)}

    def forward(self, feature):
        return run_v3_head(self, feature)


def run_v3_head(head, feature):
    x = feature['out']
    if torch.is_grad_enabled() and any(p.requires_grad for p in head.parameters()):
        return _V3Bridge.apply(head, x, *[p for p in head.parameters() if p.requires_grad])
    return ops.nhwc_to_nchw(head.fwd({'out': ops.nchw_to_nhwc(x)}, False), head.num_classes)


class _V3Bridge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, head, x, *params):
        ctx.head, ctx.cin = head, x.shape[1]
        return ops.nhwc_to_nchw(head.fwd({'out': ops.nchw_to_nhwc(x)}, True), head.num_classes)

    @staticmethod
    def backward(ctx, dy):
        head = ctx.head
        d = head.bwd(ops.nchw_to_nhwc(dy.contiguous(), _hip.pad4(head.num_classes)), _hip.GradSink())
        return (None, ops.nhwc_to_nchw(d['out'], ctx.cin)) + (None,) * (len(ctx.needs_input_grad) - 2)


class AtrousSeparableConvolution(_hip.SeparableBase):
    """Depthwise-separable conv (reference :95-119): body = [depthwise KxK conv (groups = in_channels), pointwise 1x1
    conv], both with `bias`; no normalisation in between.  Same constructor, children and state_dict keys
    (body.0.weight [Cin,1,K,K], body.1.weight [Cout,Cin,1,1], biases).  The depthwise half is an HBM-bound streaming
    kernel (csrc/dwconv.hip), the pointwise half the ordinary implicit-GEMM conv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super(AtrousSeparableConvolution, self).__init__()
        self.body = _hip.HipSequential(
            # Separable Conv
            _hip.DepthwiseConv2d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                 dilation=dilation, bias=bias, groups=in_channels),
            # PointWise Conv
            _hip.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0, bias=bias),
        )
        self._init_weight()

    _init_weight = _init_weight

    @property
    def in_channels(self):
        return self.body[1].in_channels

    @property
    def out_channels(self):
        return self.body[1].out_channels

    @property
    def cin_p(self):
        return self.body[1].cin_p

    def fwd(self, x, save, out=None):
        return self.body.fwd(x, save, out=out)

    def bwd(self, dy, sink, need_dx=True, dx=None, accumulate=False):
        return self.body.bwd(dy, sink, need_dx, dx, accumulate)

    def out_channels_of(self, cin):
        return self.body[1].out_channels

    def forward(self, x):
        return _hip.run_module(self, x)


def convert_to_separable_conv(module):
    """reference :176-188: every Conv2d with a kernel larger than 1 becomes an AtrousSeparableConvolution with the
    same geometry (fresh weights); children are carried over."""
    new_module = module
    if isinstance(module, nn.Conv2d) and not isinstance(module, _hip.DepthwiseConv2d) and module.kernel_size[0] > 1:
        new_module = AtrousSeparableConvolution(module.in_channels,
                                                module.out_channels,
                                                module.kernel_size,
                                                module.stride,
                                                module.padding,
                                                module.dilation,
                                                module.bias)
    for name, child in module.named_children():
        new_module.add_module(name, convert_to_separable_conv(child))
    return new_module
