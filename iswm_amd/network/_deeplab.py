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
        for i, conv in enumerate(self.convs):
            conv.fwd(x, save, out=cat[..., i * oc:(i + 1) * oc])
        self._saved = (tuple(x.shape), oc) if save else None
        return self.project.fwd(cat, save)

    def bwd(self, dy, sink):
        xshape, oc = self._saved
        self._saved = None
        dcat = self.project.bwd(dy, sink)
        dx = None
        for i, conv in enumerate(self.convs):
            d = conv.bwd(dcat[..., i * oc:(i + 1) * oc], sink, True, dx, dx is not None)
            dx = d if dx is None else dx
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
            cat[..., c_cat:].zero_()
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
