"""Deterministic synthetic weights / inputs shared by the oracle, the golden
generator and the tests (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Real weights are 161-237 MB and cannot be committed, and the reference's
pretrained download is unavailable offline, so every parity case loads the same
rule-generated ``state_dict`` into the reference, the oracle and the HIP path.

Key layout follows the reference module tree (374 keys for resnet50):
network/modeling.py:47-55, network/_deeplab.py:33-53,143-165,
network/backbone/resnet.py:78-98,144-155,176-198.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch

RESNET_BLOCKS = {"resnet50": [3, 4, 6, 3], "resnet101": [3, 4, 23, 3]}


class ArchCfg:
    """Static description of one DeepLab variant (what _segm_resnet decides,
    network/modeling.py:12-56)."""

    def __init__(self, name="deeplabv3plus", backbone="resnet50", num_classes=2,
                 output_stride=16, width=64, aspp_out=256, low_out=48,
                 blocks=None, in_channels=3):
        self.name = name
        self.in_channels = in_channels
        self.backbone = backbone
        self.num_classes = num_classes
        self.output_stride = output_stride
        self.width = width            # stem channels (64 in the reference)
        self.aspp_out = aspp_out      # 256 in the reference
        self.low_out = low_out        # 48 in the reference
        self.blocks = list(blocks) if blocks is not None else RESNET_BLOCKS[backbone]
        # network/modeling.py:14-19
        if output_stride == 8:
            self.replace = [False, True, True]
            self.aspp_dilate = [12, 24, 36]
        else:
            self.replace = [False, False, True]
            self.aspp_dilate = [6, 12, 18]

    def layers(self):
        """Per-stage (planes, stride, [dilation per block], has_downsample).
        Restates ResNet._make_layer, network/backbone/resnet.py:176-198."""
        out = []
        inplanes = self.width
        dilation = 1
        for li, nblocks in enumerate(self.blocks):
            planes = self.width * (2 ** li)
            stride = 1 if li == 0 else 2
            dilate = False if li == 0 else self.replace[li - 1]
            prev_dil = dilation
            if dilate:
                dilation *= stride
                stride = 1
            down = (stride != 1) or (inplanes != planes * 4)
            dils = [prev_dil] + [dilation] * (nblocks - 1)
            out.append(dict(planes=planes, stride=stride, dils=dils, down=down,
                            inplanes=inplanes))
            inplanes = planes * 4
        return out


def _bn(sh, prefix, c):
    sh[prefix + ".weight"] = (c,)
    sh[prefix + ".bias"] = (c,)
    sh[prefix + ".running_mean"] = (c,)
    sh[prefix + ".running_var"] = (c,)
    sh[prefix + ".num_batches_tracked"] = ()


def bottleneck_shapes(pre, inpl, p, down, sh=None):
    """Bottleneck(inplanes, planes, ...) keys, network/backbone/resnet.py:81-97."""
    sh = OrderedDict() if sh is None else sh
    sh[pre + ".conv1.weight"] = (p, inpl, 1, 1)
    _bn(sh, pre + ".bn1", p)
    sh[pre + ".conv2.weight"] = (p, p, 3, 3)
    _bn(sh, pre + ".bn2", p)
    sh[pre + ".conv3.weight"] = (4 * p, p, 1, 1)
    _bn(sh, pre + ".bn3", 4 * p)
    if down:
        sh[pre + ".downsample.0.weight"] = (4 * p, inpl, 1, 1)
        _bn(sh, pre + ".downsample.1", 4 * p)
    return sh


def aspp_shapes(ap, cin, a=256, sh=None):
    """ASPP(in_channels, rates) keys, network/_deeplab.py:143-165."""
    sh = OrderedDict() if sh is None else sh
    sh[ap + ".convs.0.0.weight"] = (a, cin, 1, 1)
    _bn(sh, ap + ".convs.0.1", a)
    for i in (1, 2, 3):
        sh[ap + ".convs.%d.0.weight" % i] = (a, cin, 3, 3)
        _bn(sh, ap + ".convs.%d.1" % i, a)
    sh[ap + ".convs.4.1.weight"] = (a, cin, 1, 1)
    _bn(sh, ap + ".convs.4.2", a)
    sh[ap + ".project.0.weight"] = (a, 5 * a, 1, 1)
    _bn(sh, ap + ".project.1", a)
    return sh


def head_v3plus_shapes(c, cin, low, num_classes, a=256, low_out=48, sh=None):
    """DeepLabHeadV3Plus keys, network/_deeplab.py:33-53."""
    sh = OrderedDict() if sh is None else sh
    sh[c + ".project.0.weight"] = (low_out, low, 1, 1)
    _bn(sh, c + ".project.1", low_out)
    aspp_shapes(c + ".aspp", cin, a, sh)
    sh[c + ".classifier.0.weight"] = (a, a + low_out, 3, 3)
    _bn(sh, c + ".classifier.1", a)
    sh[c + ".classifier.3.weight"] = (a, a, 3, 3)
    _bn(sh, c + ".classifier.4", a)
    sh[c + ".classifier.6.weight"] = (num_classes, a, 1, 1)
    sh[c + ".classifier.6.bias"] = (num_classes,)
    return sh


def head_v3_shapes(c, cin, num_classes, a=256, sh=None):
    """DeepLabHead keys, network/_deeplab.py:71-82."""
    sh = OrderedDict() if sh is None else sh
    aspp_shapes(c + ".classifier.0", cin, a, sh)
    sh[c + ".classifier.1.weight"] = (a, a, 3, 3)
    _bn(sh, c + ".classifier.2", a)
    sh[c + ".classifier.4.weight"] = (num_classes, a, 1, 1)
    sh[c + ".classifier.4.bias"] = (num_classes,)
    return sh


def param_shapes(cfg):
    """OrderedDict key -> shape in registration order of the reference."""
    sh = OrderedDict()
    w = cfg.width
    sh["backbone.conv1.weight"] = (w, cfg.in_channels, 7, 7)
    _bn(sh, "backbone.bn1", w)
    for li, L in enumerate(cfg.layers()):
        inpl = L["inplanes"]
        p = L["planes"]
        for bi in range(len(L["dils"])):
            pre = "backbone.layer%d.%d" % (li + 1, bi)
            bottleneck_shapes(pre, inpl, p, bi == 0 and L["down"], sh)
            inpl = 4 * p
    cin = cfg.width * 8 * 4
    low = cfg.width * 4
    if cfg.name == "deeplabv3plus":
        head_v3plus_shapes("classifier", cin, low, cfg.num_classes, cfg.aspp_out, cfg.low_out, sh)
    else:
        head_v3_shapes("classifier", cin, cfg.num_classes, cfg.aspp_out, sh)
    return sh


def synth_from_shapes(shapes, salt=0):
    return OrderedDict((k, synth_tensor(k, s, salt)) for k, s in shapes.items())


def synth_tensor(key, shape, salt=0):
    """One tensor from a per-key seeded generator (independent of key order)."""
    seed = zlib.crc32(key.encode()) ^ (salt * 0x9E3779B1 & 0xFFFFFFFF)
    g = np.random.Generator(np.random.PCG64(seed))
    if key.endswith("num_batches_tracked"):
        return torch.zeros((), dtype=torch.int64)
    if key.endswith("running_var"):
        a = g.uniform(0.5, 1.5, size=shape)
    elif key.endswith("running_mean"):
        a = g.uniform(-0.1, 0.1, size=shape)
    elif len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        a = g.normal(0.0, np.sqrt(2.0 / fan_in), size=shape)
    elif key.endswith(".weight"):      # BN gamma
        a = g.uniform(0.5, 1.5, size=shape)
    else:                              # BN beta / conv bias
        a = g.uniform(-0.1, 0.1, size=shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def synth_state_dict(cfg, salt=0):
    return synth_from_shapes(param_shapes(cfg), salt)


def synth_images(b, h, w, seed=0, c=3):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(b, c, h, w, generator=g, dtype=torch.float32)


def synth_labels(b, h, w, seed=0, p_fg=0.10, p_ignore=0.0):
    """int64 labels in {0,1,(255)} -- Bernoulli foreground."""
    g = torch.Generator().manual_seed(seed + 1000)
    u = torch.rand(b, h, w, generator=g)
    lab = (u < p_fg).to(torch.int64)
    if p_ignore > 0:
        lab[u > 1.0 - p_ignore] = 255
    return lab
