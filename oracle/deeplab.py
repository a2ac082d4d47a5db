"""CPU restatement of the reference DeepLabV3/V3+ forward (TEST INFRASTRUCTURE).

Functional style over a flat ``state_dict`` (same 374/629 keys as the reference)
so that every op on the hot path is visible in one place; backward comes from
torch autograd over the same stock ATen fp32 ops the reference runs.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .synth import ArchCfg, synth_state_dict

BN_EPS = 1e-5        # nn.BatchNorm2d default, e.g. network/_deeplab.py:38
BN_MOMENTUM = 0.1


class OracleDeepLab:
    """Holds the parameters/buffers and evaluates the reference graph.

    forward == _SimpleSegmentationModel.forward (network/utils.py:16-25).
    """

    def __init__(self, cfg: ArchCfg, state_dict=None, dropout_p=0.1):
        self.cfg = cfg
        sd = state_dict if state_dict is not None else synth_state_dict(cfg)
        self.sd = OrderedDict()
        for k, v in sd.items():
            t = v.detach().clone()
            if t.is_floating_point() and not (k.endswith("running_mean") or k.endswith("running_var")):
                t.requires_grad_(True)
            self.sd[k] = t
        self.training = False
        self.conv_math = "f32"      # "bf16": see _conv
        # act_bf16 (with conv_math "bf16"): the product's mixed-precision mode STORES the activations that feed convolutions as
        # one bf16 plane (round to nearest even); the restatement rounds at the same points -- every stage output that the
        # product writes as planes (_store) -- with a straight-through gradient, as the hand-written backward has
        self.act_bf16 = False
        self.dropout_p = dropout_p  # nn.Dropout(0.1), network/_deeplab.py:165
        # Optional {site: bool NCHW mask}: when given, every ReLU uses the SUPPLIED sign pattern
        # (z * mask) instead of its own (z > 0).  Tests pass the HIP path's masks here so that
        # gradients can be compared element-wise: two fp32 implementations disagree on the sign of
        # the ~1e-6 fraction of pre-activations that are within rounding of zero, and one flipped
        # ReLU changes a small-batch gradient by O(1/pixels) -- a discontinuity, not an error.
        self.relu_masks = None
        # Optional [N,C,Ho,Wo] tap index (kh * 3 + kw) per output element of the stem's MaxPool: when given, the pool takes THAT
        # element of each window instead of its own argmax (the HIP path's choice; same reason as relu_masks: a near-tie between two
        # window elements is decided differently by two fp32 evaluations, and the window's gradient then goes to another pixel)
        self.pool_index = None
        self.preact = None          # when a dict: records every ReLU input (for near-tie margins)

    # -- nn.Module-like helpers -------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return [v for v in self.sd.values() if v.requires_grad]

    def named_parameters(self):
        return [(k, v) for k, v in self.sd.items() if v.requires_grad]

    def state_dict(self):
        return OrderedDict((k, v.detach()) for k, v in self.sd.items())

    def zero_grad(self):
        for p in self.parameters():
            p.grad = None

    # -- building blocks ----------------------------------------------------------
    def _store(self, t, force=False):
        """an activation the product keeps pre-split for a consumer conv: 64-aligned channel counts (or a slice of a wider
        planes buffer: force) are rounded to bf16 under act_bf16; identity otherwise"""
        if not (self.act_bf16 and self.conv_math == "bf16") or not (force or t.shape[1] % 64 == 0):
            return t
        return t + (t.bfloat16().to(t.dtype) - t).detach()

    def _relu(self, z, site, store=True, force=False):
        """nn.ReLU at the site named by the BatchNorm that feeds it (store: the product writes this output as planes)."""
        if self.preact is not None:
            self.preact[site] = z.detach()
        if self.relu_masks is not None:
            a = z * self.relu_masks[site].to(z.dtype)
        else:
            a = F.relu(z)
        return self._store(a, force) if store else a

    def _bn(self, x, prefix):
        """nn.BatchNorm2d train/eval incl. running-stat update (unbiased var,
        momentum 0.1) -- e.g. network/backbone/resnet.py:89-93."""
        sd = self.sd
        if self.training:
            sd[prefix + ".num_batches_tracked"] += 1
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                            sd[prefix + ".weight"], sd[prefix + ".bias"],
                            self.training, BN_MOMENTUM, BN_EPS)

    def _conv(self, x, key, stride=1, padding=0, dilation=1, bias=None):
        w = self.sd[key]
        if self.conv_math == "bf16":
            # mixed-precision arithmetic of the product's conv math 2 (no counterpart in the reference -- BASELINE
            # configs[4]): matrix operands rounded to bf16, exact products, fp32 accumulation.  Which of the three GEMMs
            # of a conv take rounded operands follows the product: the forward when the (padded) input channel count is
            # 32-aligned (everything but the stem; the decoder's 304 is padded to 320), the data gradient when Cout is
            # 32-aligned (not the 48-wide projection or the classifier), the weight gradient always.
            cin = w.shape[1]
            cin_p = cin if (cin % 32 == 0 or cin < 128) else (cin + 31) // 32 * 32
            y = _Bf16Conv.apply(x, w, stride, padding, dilation, cin_p % 32 == 0, w.shape[0] % 32 == 0)
            return y if bias is None else y + bias.view(1, -1, 1, 1)
        return F.conv2d(x, w, bias, stride, padding, dilation)

    def _bottleneck(self, x, pre, stride, dilation, down):
        """Bottleneck.forward, network/backbone/resnet.py:99-120."""
        out = self._relu(self._bn(self._conv(x, pre + ".conv1.weight"), pre + ".bn1"), pre + ".bn1")
        out = self._conv(out, pre + ".conv2.weight", stride, dilation, dilation)  # conv3x3 :27-30
        out = self._relu(self._bn(out, pre + ".bn2"), pre + ".bn2")
        out = self._bn(self._conv(out, pre + ".conv3.weight"), pre + ".bn3")
        identity = x
        if down:
            identity = self._bn(self._conv(x, pre + ".downsample.0.weight", stride),
                                pre + ".downsample.1")
        return self._relu(out + identity, pre + ".bn3")

    def backbone(self, x):
        """IntermediateLayerGetter.forward over the ResNet children
        (network/utils.py:78-93, network/backbone/resnet.py:144-155)."""
        x = self._conv(x, "backbone.conv1.weight", 2, 3)
        x = self._relu(self._bn(x, "backbone.bn1"), "backbone.bn1", store=False)
        if self.pool_index is not None:
            self.pool_gap = float((F.max_pool2d(x, 3, 2, 1) - _pool_select(x, self.pool_index)).detach().abs().max() /
                                  x.detach().abs().max())          # how far the imposed choices are from this evaluation's maxima
        x = self._store(F.max_pool2d(x, 3, 2, 1) if self.pool_index is None else _pool_select(x, self.pool_index))
        feats = OrderedDict()
        for li, L in enumerate(self.cfg.layers()):
            for bi, d in enumerate(L["dils"]):
                pre = "backbone.layer%d.%d" % (li + 1, bi)
                x = self._bottleneck(x, pre, L["stride"] if bi == 0 else 1, d,
                                     L["down"] and bi == 0)
            if li == 0:
                feats["low_level"] = x
        feats["out"] = x
        return feats

    def aspp(self, x, ap):
        """ASPP.forward, network/_deeplab.py:143-172 (ASPPConv :121-128,
        ASPPPooling :130-141)."""
        res = [self._relu(self._bn(self._conv(x, ap + ".convs.0.0.weight"), ap + ".convs.0.1"), ap + ".convs.0.1")]
        for i, r in zip((1, 2, 3), self.cfg.aspp_dilate):
            y = self._conv(x, ap + ".convs.%d.0.weight" % i, 1, r, r)
            res.append(self._relu(self._bn(y, ap + ".convs.%d.1" % i), ap + ".convs.%d.1" % i))
        size = x.shape[-2:]
        p = F.adaptive_avg_pool2d(x, 1)
        p = self._relu(self._bn(self._conv(p, ap + ".convs.4.1.weight"), ap + ".convs.4.2"), ap + ".convs.4.2", store=False)
        res.append(self._store(F.interpolate(p, size=size, mode="bilinear", align_corners=False)))
        y = torch.cat(res, dim=1)
        y = self._relu(self._bn(self._conv(y, ap + ".project.0.weight"), ap + ".project.1"), ap + ".project.1", store=False)
        return F.dropout(y, self.dropout_p, self.training)

    def head(self, feats):
        c = "classifier"
        if self.cfg.name == "deeplabv3plus":
            # DeepLabHeadV3Plus.forward, network/_deeplab.py:55-61
            low = self._relu(self._bn(self._conv(feats["low_level"], c + ".project.0.weight"),
                                      c + ".project.1"), c + ".project.1", force=True)      # a slice of the 320-wide planes buffer
            y = self.aspp(feats["out"], c + ".aspp")
            y = self._store(F.interpolate(y, size=low.shape[2:], mode="bilinear", align_corners=False))
            y = torch.cat([low, y], dim=1)
            y = self._relu(self._bn(self._conv(y, c + ".classifier.0.weight", 1, 1), c + ".classifier.1"), c + ".classifier.1")
            y = self._relu(self._bn(self._conv(y, c + ".classifier.3.weight", 1, 1), c + ".classifier.4"), c + ".classifier.4")
            return self._conv(y, c + ".classifier.6.weight", bias=self.sd[c + ".classifier.6.bias"])
        # DeepLabHead.forward, network/_deeplab.py:71-93
        y = self.aspp(feats["out"], c + ".classifier.0")
        y = self._relu(self._bn(self._conv(y, c + ".classifier.1.weight", 1, 1), c + ".classifier.2"), c + ".classifier.2")
        return self._conv(y, c + ".classifier.4.weight", bias=self.sd[c + ".classifier.4.bias"])

    def forward(self, x):
        input_shape = x.shape[-2:]
        y = self.head(self.backbone(x))
        return F.interpolate(y, size=input_shape, mode="bilinear", align_corners=False)

    __call__ = forward


def _pool_select(x, index):
    """MaxPool2d(3, stride 2, padding 1) (network/backbone/resnet.py:148) with the window element named by `index`"""
    xp = F.pad(x, (1, 1, 1, 1), value=float("-inf"))
    win = xp.unfold(2, 3, 2).unfold(3, 3, 2)                     # [N, C, Ho, Wo, 3, 3]
    win = win.reshape(win.shape[:4] + (9,))
    return win.gather(4, index.long().unsqueeze(-1)).squeeze(-1)


class _Bf16Conv(torch.autograd.Function):
    """conv2d whose GEMMs (forward, data gradient, weight gradient) take bf16-rounded operands and accumulate in fp32 --
    what iswm_set_conv_math(2) computes; the flags keep a GEMM in fp32 where the product does."""

    @staticmethod
    def forward(ctx, x, w, stride, padding, dilation, fwd_bf16, dgrad_bf16):
        r = lambda t: t.bfloat16().float()
        ctx.save_for_backward(x, w)
        ctx.geom = (stride, padding, dilation, dgrad_bf16)
        if not fwd_bf16:
            return F.conv2d(x, w, None, stride, padding, dilation)
        return F.conv2d(r(x), r(w), None, stride, padding, dilation)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, dilation, dgrad_bf16 = ctx.geom
        r = lambda t: t.bfloat16().float()
        dyr = r(dy)
        if dgrad_bf16:
            dx = torch.nn.grad.conv2d_input(x.shape, r(w), dyr, stride, padding, dilation)
        else:
            dx = torch.nn.grad.conv2d_input(x.shape, w, dy, stride, padding, dilation)
        dw = torch.nn.grad.conv2d_weight(r(x), w.shape, dyr, stride, padding, dilation)
        return dx, dw, None, None, None, None, None


def argmax_mask(logits):
    """``logits.max(1)[1]`` -- train.py:644,659; ties resolve to the lowest index."""
    return logits.max(1)[1]


def sepconv_forward(x, sd, prefix, stride, padding, dilation):
    """AtrousSeparableConvolution.forward (network/_deeplab.py:95-111): depthwise KxK conv (groups = channels,
    weight sd[prefix + 'body.0.weight'] of shape [C,1,K,K]) followed by a pointwise 1x1 conv; both carry the
    optional biases body.{0,1}.bias."""
    import torch.nn.functional as F
    w0, w1 = sd[prefix + "body.0.weight"], sd[prefix + "body.1.weight"]
    y = F.conv2d(x, w0, sd.get(prefix + "body.0.bias"), stride, padding, dilation, groups=w0.shape[0])
    return F.conv2d(y, w1, sd.get(prefix + "body.1.bias"))
