"""Model wrapper and backbone feature tap -- counterpart of the reference's
network/utils.py (_SimpleSegmentationModel :7-25, IntermediateLayerGetter :28-93)."""
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _hip
from .. import ops
from .._hip_compat import DeQuantStub, QuantStub


class _SegBridge(torch.autograd.Function):
    """The whole model as ONE autograd node: NCHW image in, NCHW logits out."""

    @staticmethod
    def forward(ctx, model, x, *params):
        ctx.model = model
        ctx.need_dx = x.requires_grad
        ctx.cin = x.shape[1]
        return model._fwd(x, True)

    @staticmethod
    def backward(ctx, dlogits):
        dx = ctx.model._bwd(dlogits.contiguous(), ctx.need_dx)
        return (None, dx) + (None,) * (len(ctx.needs_input_grad) - 2)


class _SimpleSegmentationModel(nn.Module):
    def __init__(self, backbone, classifier):
        super(_SimpleSegmentationModel, self).__init__()
        self.backbone = backbone
        self.classifier = classifier
        # reference :12-14 -- identities in float mode, own no parameters
        self.quant = QuantStub()
        self.dequant = DeQuantStub()
        self._saved = None
        self._iswm_on_ready = None   # set by iswm_amd.parallel for gradient all-reduce overlap

    # x NCHW [B,C,H,W] fp32 on the GPU -> logits NCHW [B,num_classes,H,W]
    def _counters(self):
        """one flat tensor behind every BatchNorm's num_batches_tracked (rebuilt if the module was
        moved / reloaded in a way that broke the views)"""
        flat = getattr(self, "_iswm_nbt", None)
        first = next((m for m in self.modules() if isinstance(m, nn.BatchNorm2d)), None)
        if first is None:
            return None
        nbt = first.num_batches_tracked
        if flat is None or flat.device != nbt.device or nbt.data_ptr() != flat.data_ptr():
            flat = _hip.fuse_batch_counters(self)
            object.__setattr__(self, "_iswm_nbt", flat)
        return flat

    def _packer(self):
        pk = getattr(self, "_iswm_packer", None)
        if pk is None:
            pk = _hip.WeightPacker(self)
            object.__setattr__(self, "_iswm_packer", pk)
        return pk

    def _fwd(self, x, save):
        n, c, h, w = x.shape
        if self.training:
            flat = self._counters()
            if flat is not None:
                bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
                if all(b.training for b in bns):
                    flat.add_(1)                               # one op for every num_batches_tracked
                else:                                          # utils.fix_bn etc.: only layers in training mode count
                    for b in bns:
                        if b.training and b.num_batches_tracked is not None:
                            b.num_batches_tracked.add_(1)
        pk = self._packer()
        pk.begin()                                             # all conv weights -> bf16x6 fragments, one launch
        try:
            xh = ops.nchw_to_nhwc(x)                           # pads 3 -> 4 channels
            feats = self.backbone.fwd(xh, save)
            yl = self.classifier.fwd(feats, save)              # [B, hl, wl, pad4(num_classes)]
        except BaseException:
            pk.end()
            raise
        if not save:
            pk.end()                                           # no backward will follow: retire the packed weights
        nc = self.classifier.num_classes
        self._saved = (tuple(yl.shape), c) if save else None
        # bilinear upsample to the input size fused with NHWC -> NCHW (reference :22)
        return ops.bilinear_to_nchw_fwd(yl, nc, h, w)

    def _bwd(self, dlogits, need_dx):
        (n, hl, wl, cp), cin = self._saved
        self._saved = None
        sink = _hip.GradSink(self._iswm_on_ready)
        try:
            dyl = ops.bilinear_to_nchw_bwd(dlogits, hl, wl, cp)
            dfeats = self.classifier.bwd(dyl, sink)
            if getattr(self, "_debug_keep_dfeats", False):
                self._debug_dfeats = {k: v.clone() for k, v in dfeats.items()}
            dxh = self.backbone.bwd(dfeats, sink, need_dx)
        finally:
            self._packer().end()
        return ops.nhwc_to_nchw(dxh, cin) if need_dx else None

    def forward(self, x):
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):
            raise ValueError("iswm_amd model takes a 4-D fp32 CUDA NCHW tensor (there is no CPU fallback)")
        if torch.is_grad_enabled():
            params = [p for p in self.parameters() if p.requires_grad]
            if params or x.requires_grad:
                return _SegBridge.apply(self, x, *params)
        return self._fwd(x, False)


class IntermediateLayerGetter(nn.ModuleDict):
    """Runs the backbone children in registration order and returns the tapped
    activations (reference :59-93).  HIP version: conv1+bn1+relu is one fused stage,
    features are NHWC tensors."""

    def __init__(self, model, return_layers, hrnet_flag=False):
        if not set(return_layers).issubset([name for name, _ in model.named_children()]):
            raise ValueError("return_layers are not present in model")
        if hrnet_flag:
            raise NotImplementedError("HRNet backbones are not part of this build")
        self.hrnet_flag = hrnet_flag
        orig_return_layers = return_layers
        return_layers = {k: v for k, v in return_layers.items()}
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            if name in return_layers:
                del return_layers[name]
            if not return_layers:
                break
        super(IntermediateLayerGetter, self).__init__(layers)
        self.return_layers = orig_return_layers
        self._saved = None

    def fwd(self, x, save):
        out = OrderedDict()
        stem_ctx = None
        for name, module in self.named_children():
            if name == "conv1":
                # the max-pool reads the stem's output as fp32
                fmt = "f32" if "maxpool" in self._modules else None
                x, stem_ctx = _hip.cba_fwd(self["conv1"], self["bn1"], True, x, save, out_fmt=fmt)
            elif name in ("bn1", "relu"):
                pass                                   # fused into the conv1 stage above
            else:
                x = module.fwd(x, save)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        self._saved = stem_ctx if save else None
        return out

    def bwd(self, dfeats, sink, need_dx=False):
        names = [n for n, _ in self.named_children()]
        inv = {v: k for k, v in self.return_layers.items()}
        tap = {inv[k]: g for k, g in dfeats.items() if g is not None}
        dy = None
        order = list(reversed(names))
        for k, name in enumerate(order):
            if name in tap:
                g = tap.pop(name)
                dy = g if dy is None else ops.add_inplace(dy, g)
            if name in ("relu", "bn1"):
                continue
            if name == "conv1":
                dy, _ = _hip.cba_bwd(self["conv1"], self["bn1"], self._saved, dy, sink, need_dx=need_dx)
            else:
                # a tapped feature right below this layer (low_level under layer2): the layer's first block accumulates its data
                # gradient into the tap's gradient instead of a separate 800-MB add pass
                below = order[k + 1] if k + 1 < len(order) else None
                if below in tap and getattr(self[name], "accepts_dx0", False) and not ops.is_planes(tap[below]):
                    dy = self[name].bwd(dy, sink, dx0=tap.pop(below))
                else:
                    dy = self[name].bwd(dy, sink)
        self._saved = None
        return dy

    def forward(self, x):
        feats = self.fwd(ops.nchw_to_nhwc(x), False)
        return OrderedDict((k, ops.nhwc_to_nchw(v)) for k, v in feats.items())     # (joins pre-split features)
