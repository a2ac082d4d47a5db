"""Dilated ResNet-50/101/152 backbone on the HIP execution layer.

Counterpart of the reference's network/backbone/resnet.py: same constructor
arguments, module tree and state_dict keys (Bottleneck :78-120, ResNet :123-215,
_make_layer :176-198, resnet50/101 :251-272); forward/backward run as
libiswm_hip.so kernels through ``_hip.cba_fwd`` / ``cba_bwd``.
"""
import torch
import torch.nn as nn

from .. import _hip
from ..._hip_compat import FloatFunctional
from ... import ops

__all__ = ['ResNet', 'resnet50', 'resnet101', 'resnet152']


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    """3x3 convolution with padding (reference :27-30)"""
    return _hip.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation, groups=groups,
                       bias=False, dilation=dilation)


def conv1x1(in_planes, out_planes, stride=1):
    """1x1 convolution (reference :33-35)"""
    return _hip.Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class Bottleneck(_hip.HipModule):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super(Bottleneck, self).__init__()
        if norm_layer is None:
            norm_layer = _hip.BatchNorm2d
        width = int(planes * (base_width / 64.)) * groups
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = _hip.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.add = FloatFunctional()   # reference :97 (identity in float mode, owns no state)
        self._saved = None

    def fwd(self, x, save):
        o1, c1 = _hip.cba_fwd(self.conv1, self.bn1, True, x, save)
        o2, c2 = _hip.cba_fwd(self.conv2, self.bn2, True, o1, save)
        identity, cd = x, None
        if self.downsample is not None:
            identity, cd = _hip.cba_fwd(self.downsample[0], self.downsample[1], False, x, save)
        # conv3 -> bn3 -> (+identity) -> relu in one fused elementwise pass (reference :110-118)
        out, c3 = _hip.cba_fwd(self.conv3, self.bn3, True, o2, save, residual=identity)
        self._saved = (c1, c2, c3, cd) if save else None
        return out

    def bwd(self, dout, sink, need_dx=True, up=None, dx0=None):
        """dx0: a gradient that already stands at this block's input (a tapped feature's: the decoder's low-level branch) -- the
        block's data gradients accumulate into it instead of a separate add pass.
        up: the bn3 stage context of the PREVIOUS block when this block's input gradient is consumed by that stage alone
        (identity blocks inside a layer): conv1's accumulating data gradient then applies that stage's ReLU pattern and takes
        its BatchNorm-backward sums (ops.BnStats.mask)"""
        c1, c2, c3, cd = self._saved
        self._saved = None
        # conv3's / conv2's data gradients are consumed by bn2's / bn1's backward alone: they take its reduction pass along
        d2, dres = _hip.cba_bwd(self.conv3, self.bn3, c3, dout, sink, up=c2)
        d1, _ = _hip.cba_bwd(self.conv2, self.bn2, c2, d2, sink, up=c1)
        if cd is not None:
            # conv1 (1x1, stride 1) writes every input pixel; the strided downsample conv then only touches the
            # pixels it reaches (its data gradient skips the other parity classes when accumulating)
            dx, _ = _hip.cba_bwd(self.conv1, self.bn1, c1, d1, sink, dx=dx0, accumulate=dx0 is not None)
            dx, _ = _hip.cba_bwd(self.downsample[0], self.downsample[1], cd, dres, sink, dx=dx, accumulate=True)
        else:
            # identity branch: the residual gradient IS dx; conv1's dgrad accumulates into it
            dx, _ = _hip.cba_bwd(self.conv1, self.bn1, c1, d1, sink, dx=dres, accumulate=True, up=up)
            if dx0 is not None:
                dx = ops.add_inplace(dx, dx0)
        return dx

    def out_channels_of(self, cin):
        return self.conv3.out_channels


class _Layer(_hip.HipModule, nn.Sequential):
    """nn.Sequential of Bottlenecks (reference _make_layer returns nn.Sequential)."""
    accepts_dx0 = True          # bwd(dy, sink, dx0): IntermediateLayerGetter hands a tapped feature's gradient down

    def fwd(self, x, save):
        for m in self:
            x = m.fwd(x, save)
        return x

    def bwd(self, dy, sink, dx0=None):
        """dx0: see Bottleneck.bwd (handed to the first block)"""
        blocks = list(self)
        for i in range(len(blocks) - 1, -1, -1):
            up = None
            if i > 0 and blocks[i].downsample is None and blocks[i - 1]._saved is not None:
                up = blocks[i - 1]._saved[2]            # the previous block's conv3 / bn3 / + identity / ReLU stage
            dy = blocks[i].bwd(dy, sink, up=up, dx0=dx0 if i == 0 else None)
        return dy

    def out_channels_of(self, cin):
        return list(self)[-1].conv3.out_channels


class MaxPool2d(_hip.HipModule, nn.MaxPool2d):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (reference :148)."""

    def fwd(self, x, save):
        if (self.kernel_size, self.stride, self.padding) != (3, 2, 1):
            raise NotImplementedError("HIP max-pool is the stem's 3x3 / stride 2 / pad 1")
        # layer1 follows: its 1x1 convs take the pooled map pre-split
        y, idx = ops.maxpool_fwd(x, planes=ops.planes_on() and x.shape[3] % 64 == 0)
        if _hip.POOL_RECORDER is not None:
            _hip.POOL_RECORDER[self] = idx
        self._saved = (idx, tuple(x.shape)) if save else None
        return y

    def bwd(self, dy, sink):
        idx, shape = self._saved
        self._saved = None
        return ops.maxpool_bwd(dy, idx, shape)


class ResNet(_hip.HipModule):

    def __init__(self, block, layers, num_classes=1000, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None):
        super(ResNet, self).__init__()
        if norm_layer is None:
            norm_layer = _hip.BatchNorm2d
        self._norm_layer = norm_layer
        if groups != 1 or width_per_group != 64:
            raise NotImplementedError("HIP backbone covers the groups=1 / width 64 family (resnet50/101/152)")
        self.inplanes = 64
        self.dilation = 1
        if replace_stride_with_dilation is None:
            replace_stride_with_dilation = [False, False, False]
        if len(replace_stride_with_dilation) != 3:
            raise ValueError("replace_stride_with_dilation should be None "
                             "or a 3-element tuple, got {}".format(replace_stride_with_dilation))
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = _hip.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = _hip.ReLU(inplace=True)
        self.maxpool = MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2, dilate=replace_stride_with_dilation[0])
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2, dilate=replace_stride_with_dilation[1])
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2, dilate=replace_stride_with_dilation[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        norm_layer = self._norm_layer
        downsample = None
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                conv1x1(self.inplanes, planes * block.expansion, stride),
                norm_layer(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width,
                        previous_dilation, norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                                dilation=self.dilation, norm_layer=norm_layer))
        return _Layer(*layers)

    def forward(self, x):
        raise NotImplementedError("the ImageNet classification head (avgpool+fc) is not on the segmentation hot "
                                  "path; wrap the backbone in network.utils.IntermediateLayerGetter")


def _resnet(arch, block, layers, pretrained, progress, **kwargs):
    model = ResNet(block, layers, **kwargs)
    if pretrained:
        # reference :220-223 downloads torchvision weights; there is no network in this deployment
        raise RuntimeError("pretrained=True needs a download; load a checkpoint with load_state_dict instead")
    return model


def resnet50(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet50', Bottleneck, [3, 4, 6, 3], pretrained, progress, **kwargs)


def resnet101(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet101', Bottleneck, [3, 4, 23, 3], pretrained, progress, **kwargs)


def resnet152(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet152', Bottleneck, [3, 8, 36, 3], pretrained, progress, **kwargs)
