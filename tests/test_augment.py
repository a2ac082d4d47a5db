"""Input pipeline: the device augmentation kernel vs the reference's PIL transform chain.
CPU tests pin the host-side Pillow tables (iswm_amd/utils/ext_transforms.py) against Pillow itself through a numpy
emulation of the kernel's integer arithmetic, and the order random numbers are consumed in; GPU tests compare the
kernel bit-for-bit with oracle/augment.py."""
import random

import numpy as np
import pytest
import torch

from oracle import augment as oaug

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]          # train.py:360-361


def _emulate_resize(img, rs_h, rs_w):
    """what csrc/augment.hip computes for the resized image, in numpy int64, from the product's tables"""
    from iswm_amd.utils.ext_transforms import PRECISION_BITS, _resample_tables
    h, w, _ = img.shape
    hb, hk, _ = _resample_tables(w, rs_w)
    vb, vk, _ = _resample_tables(h, rs_h)
    half = 1 << (PRECISION_BITS - 1)
    tmp = np.zeros((h, rs_w, 3), dtype=np.int64)
    for x in range(rs_w):
        x0, n = hb[x]
        acc = np.full((h, 3), half, dtype=np.int64)
        for c in range(n):
            acc += img[:, x0 + c, :].astype(np.int64) * int(hk[x, c])
        tmp[:, x, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.zeros((rs_h, rs_w, 3), dtype=np.int64)
    for y in range(rs_h):
        y0, n = vb[y]
        acc = np.full((rs_w, 3), half, dtype=np.int64)
        for r in range(n):
            acc += tmp[y0 + r] * int(vk[y, r])
        out[y] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out.astype(np.uint8)


@pytest.mark.parametrize("h,w,rs_h,rs_w", [(40, 56, 20, 28), (40, 56, 80, 112), (33, 47, 21, 60), (37, 29, 37, 29),
                                           (64, 64, 33, 127), (50, 70, 25, 141), (31, 45, 61, 22)])
def test_pillow_tables_reproduce_pillow_resize(h, w, rs_h, rs_w):
    from PIL import Image
    from iswm_amd.utils.ext_transforms import _nearest_table
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    lbl = rng.integers(0, 2, (h, w), dtype=np.uint8)
    want = np.array(Image.fromarray(img, mode="RGB").resize((rs_w, rs_h), Image.BILINEAR))
    assert np.array_equal(_emulate_resize(img, rs_h, rs_w), want)
    want_l = np.array(Image.fromarray(lbl, mode="L").resize((rs_w, rs_h), Image.NEAREST))
    xi, yi = _nearest_table(w, rs_w), _nearest_table(h, rs_h)
    assert np.array_equal(lbl[yi][:, xi], want_l)


def test_random_draw_order_matches_reference_chain():
    from iswm_amd.utils import ext_transforms as et
    comp = et.ExtCompose([et.ExtRandomScale((0.5, 2.0)), et.ExtRandomCrop(size=(65, 65), pad_if_needed=True),
                          et.ExtRandomHorizontalFlip(), et.ExtToTensor(), et.ExtNormalize(MEAN, STD)])
    random.seed(7)
    got = [comp.draw(60, 90) for _ in range(200)]
    random.seed(7)
    want = [oaug.draw_params(60, 90, (65, 65)) for _ in range(200)]
    assert got == want
    assert any(p[2] > 0 for p in got) and any(p[2] == 0 for p in got) and any(p[5] for p in got)


def test_device_pipeline_argument_errors():
    from iswm_amd.utils import ext_transforms as et
    with pytest.raises(NotImplementedError):
        et.ExtCompose([et.ExtRandomScale((0.5, 2.0)), et.ExtToTensor()])
    with pytest.raises(NotImplementedError):
        et.ExtRandomCrop(65, padding=3)
    comp = et.ExtCompose([et.ExtRandomCrop(size=(65, 65), pad_if_needed=False)])
    with pytest.raises(ValueError):
        comp.draw(40, 40)


@pytest.mark.gpu
def test_augment_batch_bit_exact_vs_pil_chain():
    from iswm_amd.utils import ext_transforms as et
    dev = torch.device("cuda:0")
    crop = (65, 65)
    comp = et.ExtCompose([et.ExtRandomScale((0.5, 2.0)), et.ExtRandomCrop(size=crop, pad_if_needed=True),
                          et.ExtRandomHorizontalFlip(), et.ExtToTensor(), et.ExtNormalize(MEAN, STD)])
    rng = np.random.default_rng(3)
    sizes = [(60, 90), (100, 70), (40, 40), (65, 65), (130, 131), (33, 200)]
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    lbls = [(rng.random((h, w)) < 0.3).astype(np.uint8) for h, w in sizes]
    random.seed(11)
    params = [oaug.draw_params(h, w, crop) for h, w in sizes]
    params[3] = (65, 65, 0, 0, 0, 1)                       # identity resize, exact fit, flipped
    out, out_lbl = comp.batch([torch.from_numpy(a).to(dev) for a in imgs], [torch.from_numpy(a).to(dev) for a in lbls],
                              params=params)
    assert out.shape == (6, 3, 65, 65) and out_lbl.shape == (6, 65, 65) and out_lbl.dtype == torch.uint8
    for b in range(len(sizes)):
        t, lab = oaug.augment_sample(imgs[b], lbls[b], params[b], crop, MEAN, STD)
        assert torch.equal(out_lbl[b].cpu(), lab), "label %d" % b
        assert torch.equal(out[b].cpu(), t), "image %d: max diff %g" % (b, float((out[b].cpu() - t).abs().max()))
    # the random path: same seed -> same augmentations as the reference chain
    random.seed(5)
    o2, l2 = comp.batch([torch.from_numpy(a).to(dev) for a in imgs], [torch.from_numpy(a).to(dev) for a in lbls])
    random.seed(5)
    for b, (h, w) in enumerate(sizes):
        t, lab = oaug.augment_sample(imgs[b], lbls[b], oaug.draw_params(h, w, crop), crop, MEAN, STD)
        assert torch.equal(o2[b].cpu(), t) and torch.equal(l2[b].cpu(), lab)


@pytest.mark.gpu
def test_augment_full_size_batch():
    """BASELINE tile size: 16 samples of 513 x 513 -> crop 513 (train.py --crop_size), checked on two of them"""
    from iswm_amd.utils import ext_transforms as et
    dev = torch.device("cuda:0")
    crop = (513, 513)
    comp = et.ExtCompose([et.ExtRandomScale((0.5, 2.0)), et.ExtRandomCrop(size=crop, pad_if_needed=True),
                          et.ExtRandomHorizontalFlip(), et.ExtToTensor(), et.ExtNormalize(MEAN, STD)])
    rng = np.random.default_rng(9)
    imgs = [rng.integers(0, 256, (513, 513, 3), dtype=np.uint8) for _ in range(16)]
    lbls = [(rng.random((513, 513)) < 0.1).astype(np.uint8) for _ in range(16)]
    random.seed(21)
    params = [oaug.draw_params(513, 513, crop) for _ in range(16)]
    out, out_lbl = comp.batch([torch.from_numpy(a).to(dev) for a in imgs], [torch.from_numpy(a).to(dev) for a in lbls],
                              params=params)
    assert out.shape == (16, 3, 513, 513) and bool(torch.isfinite(out).all())
    for b in (0, 9):
        t, lab = oaug.augment_sample(imgs[b], lbls[b], params[b], crop, MEAN, STD)
        assert torch.equal(out_lbl[b].cpu(), lab) and torch.equal(out[b].cpu(), t)
