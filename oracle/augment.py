"""TEST INFRASTRUCTURE ONLY -- the reference's training transform chain (train.py:355-362) restated on PIL images,
following utils/ext_transforms.py (ExtRandomScale :94-115, ExtRandomCrop :327-396, ExtRandomHorizontalFlip :212-233,
ExtToTensor :273-296, ExtNormalize :298-324).  The reference module itself needs torchvision (absent here, SURVEY.md
8c); the torchvision.transforms.functional calls it makes on PIL images are one-liners over Pillow, restated below
next to each step.  Pillow (the arithmetic those calls run on) is importable and is used directly, so the device
pipeline is pinned against the third-party resampler itself.  Never imported by the product."""
import random

import numpy as np
import torch
from PIL import Image, ImageOps


def draw_params(src_h, src_w, crop, scale_range=(0.5, 2.0), pad_if_needed=True, p_flip=0.5):
    """consumes Python's `random` exactly as ExtCompose([ExtRandomScale, ExtRandomCrop, ExtRandomHorizontalFlip]) does
    for one sample; returns (rs_h, rs_w, pad, crop_i, crop_j, flip)"""
    th, tw = crop
    scale = random.uniform(scale_range[0], scale_range[1])                 # :109
    rs_h, rs_w = int(src_h * scale), int(src_w * scale)                    # :110 (img.size = (w, h))
    w, h, pad = rs_w, rs_h, 0
    if pad_if_needed and w < tw:                                           # :380-382
        p = int((1 + tw - w) / 2)
        w, h, pad = w + 2 * p, h + 2 * p, pad + p
    if pad_if_needed and h < th:                                           # :385-387
        p = int((1 + th - h) / 2)
        w, h, pad = w + 2 * p, h + 2 * p, pad + p
    if w == tw and h == th:                                                # get_params :352-360
        i, j = 0, 0
    else:
        i = random.randint(0, h - th)
        j = random.randint(0, w - tw)
    flip = 1 if random.random() < p_flip else 0                            # :228
    return rs_h, rs_w, pad, i, j, flip


def augment_sample(img_u8, lbl_u8, params, crop, mean, std):
    """img_u8 [H,W,3] uint8, lbl_u8 [H,W] uint8 numpy -> (float32 [3,th,tw] tensor, uint8 [th,tw] tensor)"""
    rs_h, rs_w, pad, i, j, flip = params
    th, tw = crop
    img = Image.fromarray(img_u8, mode="RGB")
    lbl = Image.fromarray(lbl_u8, mode="L")
    # F.resize(img, (h, w), interpolation) on a PIL image == img.resize((w, h), interpolation)      (:111)
    img = img.resize((rs_w, rs_h), Image.BILINEAR)
    lbl = lbl.resize((rs_w, rs_h), Image.NEAREST)
    # F.pad(img, padding=int) on a PIL image == ImageOps.expand(img, border=padding, fill=0)        (:381-387)
    if pad:
        img = ImageOps.expand(img, border=pad, fill=0)
        lbl = ImageOps.expand(lbl, border=pad, fill=0)
    # F.crop(img, i, j, h, w) == img.crop((j, i, j + w, i + h))                                    (:391)
    img = img.crop((j, i, j + tw, i + th))
    lbl = lbl.crop((j, i, j + tw, i + th))
    if flip:                                                               # F.hflip == transpose(FLIP_LEFT_RIGHT)
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
        lbl = lbl.transpose(Image.FLIP_LEFT_RIGHT)
    # F.to_tensor: uint8 HWC -> CHW float32, .div(255); label: torch.from_numpy(np.array(lbl, dtype='uint8'))  (:291)
    t = torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    lab = torch.from_numpy(np.array(lbl, dtype="uint8"))
    # F.normalize: tensor.sub_(mean[:, None, None]).div_(std[:, None, None]) with fp32 mean / std             (:321)
    m = torch.as_tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.as_tensor(std, dtype=torch.float32)[:, None, None]
    t.sub_(m).div_(s)
    return t, lab
