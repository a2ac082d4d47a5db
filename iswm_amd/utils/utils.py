"""Small helpers of the reference's utils/utils.py:6-38 (denormalize / Denormalize, set_bn_momentum, fix_bn, mkdir),
without the torchvision dependency: `normalize(t, m, s)` there is (t - m) / s per channel."""
import os

import numpy as np
import torch
import torch.nn as nn


def _normalize(tensor, mean, std):
    mean = torch.as_tensor(mean, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
    std = torch.as_tensor(std, dtype=tensor.dtype, device=tensor.device).view(-1, 1, 1)
    return (tensor - mean) / std


def denormalize(tensor, mean, std):
    mean = np.array(mean)
    std = np.array(std)
    _mean = -mean / std
    _std = 1 / std
    return _normalize(tensor, _mean, _std)


class Denormalize(object):
    def __init__(self, mean, std):
        mean = np.array(mean)
        std = np.array(std)
        self._mean = -mean / std
        self._std = 1 / std

    def __call__(self, tensor):
        if isinstance(tensor, np.ndarray):
            return (tensor - self._mean.reshape(-1, 1, 1)) / self._std.reshape(-1, 1, 1)
        return _normalize(tensor, self._mean, self._std)


def set_bn_momentum(model, momentum=0.1):
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum = momentum


def fix_bn(model):
    """BatchNorm layers to eval mode (frozen statistics) while the rest of the model trains; the fused conv -> BN stage
    then normalises with the running statistics and its backward is the eval-mode one (dy = gamma * invstd * dz)."""
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()


def mkdir(path):
    if not os.path.exists(path):
        os.mkdir(path)
