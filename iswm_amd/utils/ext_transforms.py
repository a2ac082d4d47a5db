"""Device-side training-input pipeline -- counterpart of the transforms the reference composes in
train.py:355-362 (utils/ext_transforms.py: ExtRandomScale :94-115, ExtRandomCrop :327-396,
ExtRandomHorizontalFlip :212-233, ExtToTensor :273-296, ExtNormalize :298-324, ExtCompose :39-64).

The reference runs them per sample on PIL images inside DataLoader workers.  Here the classes keep their names
and constructor arguments but only CARRY parameters; `ExtCompose.batch(images, labels)` draws the random
parameters of every sample on the host -- with Python's `random`, in the same order the reference's chain
consumes it (uniform scale; randint i, randint j unless the sizes already match; random() < p for the flip), so a
single-process loader with the same seed produces the same augmentations -- and runs the whole batch in ONE HIP
kernel (csrc/augment.hip) from uint8 source tiles resident on the GPU.

Bit-exactness with Pillow's resize is obtained by computing Pillow's own per-column / per-row resampling tables
here in double precision (`_resample_tables` follows Resample.c precompute_coeffs + normalize_coeffs_8bpc,
`_nearest_table` follows Geometry.c ImagingScaleAffine) and letting the kernel do only the integer arithmetic.
"""
import ctypes
import math
import numbers
import random

import numpy as np
import torch

from .. import _lib
from .._lib import call

PRECISION_BITS = 32 - 8 - 2      # Pillow Resample.c


class ExtRandomScale(object):
    def __init__(self, scale_range, interpolation="bilinear"):
        self.scale_range = scale_range
        self.interpolation = interpolation


class ExtRandomCrop(object):
    def __init__(self, size, padding=0, pad_if_needed=False):
        self.size = (int(size), int(size)) if isinstance(size, numbers.Number) else tuple(size)
        if padding:
            raise NotImplementedError("ExtRandomCrop(padding>0) is not used by train.py and not built")
        self.padding = padding
        self.pad_if_needed = pad_if_needed


class ExtRandomHorizontalFlip(object):
    def __init__(self, p=0.5):
        self.p = p


class ExtToTensor(object):
    def __init__(self, normalize=True, target_type='uint8'):
        if not normalize or target_type != 'uint8':
            raise NotImplementedError("only ExtToTensor(normalize=True, target_type='uint8') (train.py:359) is built")
        self.normalize, self.target_type = normalize, target_type


class ExtNormalize(object):
    def __init__(self, mean, std):
        self.mean, self.std = list(mean), list(std)


# ---- Pillow's tables ---------------------------------------------------------------------------------------
def _resample_tables(in_size, out_size):
    """Pillow Resample.c precompute_coeffs (BILINEAR, box = whole axis) + normalize_coeffs_8bpc.
    Returns (bounds int32 [out,2] = (first source index, count), weights int32 [out, ksize], ksize)."""
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale                              # bilinear support = 1
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = (center - support + 0.5).astype(np.int64)          # C (int) cast: truncation
    xmin = np.maximum(xmin, 0)
    xmax = (center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    k = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):                                    # sequential accumulation, as the C loop
        arg = np.abs((float(x) + xmin - center + 0.5) * ss)
        w = np.where(arg < 1.0, 1.0 - arg, 0.0)
        w = np.where(x < xmax, w, 0.0)
        k[:, x] = w
        ww = ww + w
    nz = ww != 0.0
    k[nz] = k[nz] / ww[nz, None]
    kk = np.floor(0.5 + k * float(1 << PRECISION_BITS)).astype(np.int32)     # weights are never negative
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk, ksize


def _nearest_table(in_size, out_size):
    """Pillow Geometry.c ImagingScaleAffine: xo = a*0.5, then xo += a per output index; index = (int)xo."""
    a = float(in_size) / float(out_size)
    steps = np.full(out_size, a, dtype=np.float64)
    steps[0] = 0.0 + a * 0.5
    xo = np.cumsum(steps)                                     # sequential double additions
    idx = np.where(xo < 0.0, -1, xo.astype(np.int64))
    return np.clip(idx, 0, in_size - 1).astype(np.int32)       # (indices outside the source cannot occur for a resize)


class _AugSample(ctypes.Structure):
    _fields_ = [("img_off", ctypes.c_longlong), ("lbl_off", ctypes.c_longlong),
                ("src_h", ctypes.c_int), ("src_w", ctypes.c_int), ("rs_h", ctypes.c_int), ("rs_w", ctypes.c_int),
                ("pad", ctypes.c_int), ("crop_i", ctypes.c_int), ("crop_j", ctypes.c_int), ("flip", ctypes.c_int),
                ("tab_off", ctypes.c_int), ("ksize_h", ctypes.c_int), ("ksize_v", ctypes.c_int),
                ("reserved", ctypes.c_int)]


class ExtCompose(object):
    """Same constructor as the reference (a list of the Ext* transforms above).  `batch()` is the device path."""

    def __init__(self, transforms):
        self.transforms = transforms
        self.scale = self.crop = self.hflip = self.norm = None
        for t in transforms:
            if isinstance(t, ExtRandomScale):
                self.scale = t
            elif isinstance(t, ExtRandomCrop):
                self.crop = t
            elif isinstance(t, ExtRandomHorizontalFlip):
                self.hflip = t
            elif isinstance(t, ExtNormalize):
                self.norm = t
            elif not isinstance(t, ExtToTensor):
                raise NotImplementedError("transform %s has no device implementation" % type(t).__name__)
        if self.crop is None:
            raise NotImplementedError("the device pipeline needs an ExtRandomCrop (fixed output size per batch)")
        if self.norm is None:
            self.norm = ExtNormalize([0.0, 0.0, 0.0], [1.0, 1.0, 1.0])

    # ---- random parameters, consumed in the reference's order -------------------------------------------------
    def draw(self, src_h, src_w):
        """(rs_h, rs_w, pad, crop_i, crop_j, flip) of one sample"""
        rs_h, rs_w = src_h, src_w
        if self.scale is not None:
            s = random.uniform(self.scale.scale_range[0], self.scale.scale_range[1])      # :109
            rs_h, rs_w = int(src_h * s), int(src_w * s)                                   # :110
            if rs_h < 1 or rs_w < 1:
                raise ValueError("rescaled size %dx%d is empty" % (rs_h, rs_w))
        th, tw = self.crop.size
        h, w, pad = rs_h, rs_w, 0
        if self.crop.pad_if_needed and w < tw:                 # :380-382 pads ALL four sides
            p = int((1 + tw - w) / 2)
            pad, h, w = pad + p, h + 2 * p, w + 2 * p
        if self.crop.pad_if_needed and h < th:                 # :385-387
            p = int((1 + th - h) / 2)
            pad, h, w = pad + p, h + 2 * p, w + 2 * p
        if h < th or w < tw:
            raise ValueError("image %dx%d smaller than the crop %dx%d (pad_if_needed=False)" % (h, w, th, tw))
        if w == tw and h == th:                                # get_params :352-360
            i, j = 0, 0
        else:
            i = random.randint(0, h - th)
            j = random.randint(0, w - tw)
        flip = 0
        if self.hflip is not None:
            flip = 1 if random.random() < self.hflip.p else 0   # :228
        return rs_h, rs_w, pad, i, j, flip

    # ---- one launch for the whole batch -------------------------------------------------------------------
    def batch(self, images, labels, params=None):
        """images: list of uint8 [H,W,3] CUDA tensors; labels: list of uint8 [H,W] CUDA tensors (sizes may differ per
        sample).  Returns (float32 [B,3,th,tw], uint8 [B,th,tw]) on the same device.  `params` overrides the random
        draw with explicit (rs_h, rs_w, pad, crop_i, crop_j, flip) tuples (tests)."""
        if len(images) != len(labels) or not images:
            raise ValueError("need equally many images and labels (got %d, %d)" % (len(images), len(labels)))
        dev = images[0].device
        th, tw = self.crop.size
        B = len(images)
        samples = (_AugSample * B)()
        tabs, tab_off, img_off, lbl_off = [], 0, 0, 0
        for b, (im, lb) in enumerate(zip(images, labels)):
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_cuda:
                raise ValueError("image %d: expected a uint8 [H,W,3] CUDA tensor, got %s %s" % (b, tuple(im.shape), im.dtype))
            if lb.dtype != torch.uint8 or tuple(lb.shape) != tuple(im.shape[:2]) or not lb.is_cuda:
                raise ValueError("label %d: expected a uint8 %s CUDA tensor, got %s %s" % (b, tuple(im.shape[:2]), tuple(lb.shape), lb.dtype))
            sh, sw = int(im.shape[0]), int(im.shape[1])
            rs_h, rs_w, pad, ci, cj, flip = params[b] if params is not None else self.draw(sh, sw)
            hb, hk, ksh = _resample_tables(sw, rs_w)
            vb, vk, ksv = _resample_tables(sh, rs_h)
            t = np.concatenate([_nearest_table(sw, rs_w), _nearest_table(sh, rs_h), hb.reshape(-1), hk.reshape(-1),
                                vb.reshape(-1), vk.reshape(-1)])
            s = samples[b]
            s.img_off, s.lbl_off = img_off, lbl_off
            s.src_h, s.src_w, s.rs_h, s.rs_w = sh, sw, rs_h, rs_w
            s.pad, s.crop_i, s.crop_j, s.flip = pad, ci, cj, flip
            s.tab_off, s.ksize_h, s.ksize_v = tab_off, ksh, ksv
            tabs.append(t)
            tab_off += t.size
            img_off += sh * sw * 3
            lbl_off += sh * sw
        img_buf = torch.cat([im.reshape(-1) for im in images])
        lbl_buf = torch.cat([lb.reshape(-1) for lb in labels])
        tables = torch.from_numpy(np.concatenate(tabs).astype(np.int32)).to(dev)
        sbuf = torch.frombuffer(bytearray(bytes(samples)), dtype=torch.uint8).to(dev)
        out = torch.empty((B, 3, th, tw), dtype=torch.float32, device=dev)
        out_lbl = torch.empty((B, th, tw), dtype=torch.uint8, device=dev)
        mean = (ctypes.c_float * 3)(*[float(np.float32(m)) for m in self.norm.mean])
        std = (ctypes.c_float * 3)(*[float(np.float32(v)) for v in self.norm.std])
        _lib.load()
        call("iswm_augment_batch", img_buf.data_ptr(), lbl_buf.data_ptr(), sbuf.data_ptr(), tables.data_ptr(), B, th, tw,
             mean, std, out.data_ptr(), out_lbl.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return out, out_lbl

    def __call__(self, img, lbl):
        """single-sample form of the reference's ExtCompose.__call__: (float32 [3,th,tw], uint8 [th,tw])"""
        out, out_lbl = self.batch([img], [lbl])
        return out[0], out_lbl[0]
