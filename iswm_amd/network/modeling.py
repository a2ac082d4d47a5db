"""Model construction API -- counterpart of the reference's network/modeling.py
(_segm_resnet :12-56, _load_model :58-73, deeplabv3plus_resnet50 :75-83), same names
and signatures, so ``network.modeling.deeplabv3plus_resnet50(num_classes=...,
output_stride=...)`` in train.py:414-417 / predict.py:71-75 keeps working.

Differences, all deliberate:
  * ``pretrained_backbone`` defaults to False: the reference default triggers an HTTP
    download (network/backbone/resnet.py:220-223) that cannot work offline;
  * ``deeplabv3plus_resnet101`` / ``deeplabv3_resnet50/101`` are exposed (the reference
    reaches ResNet-101 only through the private ``_segm_resnet``).
"""
import torch
from torch import nn

from . import _hip
from ._deeplab import DeepLabHead, DeepLabHeadV3Plus, DeepLabV3
from .backbone import resnet
from .utils import IntermediateLayerGetter


def _segm_resnet(name, backbone_name, num_classes, output_stride, pretrained_backbone, in_channels=3):
    if output_stride == 8:
        replace_stride_with_dilation = [False, True, True]
        aspp_dilate = [12, 24, 36]
    else:
        replace_stride_with_dilation = [False, False, True]
        aspp_dilate = [6, 12, 18]

    backbone = resnet.__dict__[backbone_name](
        pretrained=pretrained_backbone,
        replace_stride_with_dilation=replace_stride_with_dilation)

    if in_channels != 3:
        # stem surgery for multi-channel satellite tiles (reference :25-43)
        original_conv = backbone.conv1
        backbone.conv1 = _hip.Conv2d(
            in_channels,
            original_conv.out_channels,
            kernel_size=original_conv.kernel_size,
            stride=original_conv.stride,
            padding=original_conv.padding,
            bias=original_conv.bias is not None
        )
        if pretrained_backbone:
            with torch.no_grad():
                backbone.conv1.weight[:, :3, :, :].data.copy_(original_conv.weight.data)
                original_weight_mean = original_conv.weight.mean(dim=1, keepdim=True)
                for i in range(3, in_channels):
                    backbone.conv1.weight[:, i:i + 1, :, :].data.copy_(original_weight_mean)
    inplanes = 2048
    low_level_planes = 256

    if name == 'deeplabv3plus':
        return_layers = {'layer4': 'out', 'layer1': 'low_level'}
        classifier = DeepLabHeadV3Plus(inplanes, low_level_planes, num_classes, aspp_dilate)
    elif name == 'deeplabv3':
        return_layers = {'layer4': 'out'}
        classifier = DeepLabHead(inplanes, num_classes, aspp_dilate)
    else:
        raise NotImplementedError(name)
    backbone = IntermediateLayerGetter(backbone, return_layers=return_layers)
    model = DeepLabV3(backbone, classifier)
    return model


def _load_model(arch_type, backbone, num_classes, output_stride, pretrained_backbone, temporal=False,
                model_type='parallel', opts=None, in_channels=3):
    if backbone.startswith('resnet'):
        model = _segm_resnet(arch_type, backbone, num_classes, output_stride=output_stride,
                             pretrained_backbone=pretrained_backbone, in_channels=in_channels)
    else:
        raise NotImplementedError
    return model


def deeplabv3plus_resnet50(num_classes=21, output_stride=8, pretrained_backbone=False):
    """Constructs a DeepLabV3+ model with a ResNet-50 backbone (reference :75-83)."""
    return _load_model('deeplabv3plus', 'resnet50', num_classes, output_stride=output_stride,
                       pretrained_backbone=pretrained_backbone)


def deeplabv3plus_resnet101(num_classes=21, output_stride=8, pretrained_backbone=False):
    """DeepLabV3+ with a ResNet-101 backbone (BASELINE.json's headline model)."""
    return _load_model('deeplabv3plus', 'resnet101', num_classes, output_stride=output_stride,
                       pretrained_backbone=pretrained_backbone)


def deeplabv3_resnet50(num_classes=21, output_stride=8, pretrained_backbone=False):
    return _load_model('deeplabv3', 'resnet50', num_classes, output_stride=output_stride,
                       pretrained_backbone=pretrained_backbone)


def deeplabv3_resnet101(num_classes=21, output_stride=8, pretrained_backbone=False):
    return _load_model('deeplabv3', 'resnet101', num_classes, output_stride=output_stride,
                       pretrained_backbone=pretrained_backbone)
