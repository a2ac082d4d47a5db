#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE's own modules
(TEST INFRASTRUCTURE; runs only in the build container, where /root/reference is
mounted read-only -- the reference itself never travels to the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py [--ref /root/reference]

What is imported: network.modeling / network._deeplab / network.utils /
network.backbone.resnet (with empty stand-in *modules* for the dead imports
``src.*`` and ``cv2`` that no class in those files uses, network/_deeplab.py:8-13)
and utils/loss.py (loaded standalone; its package __init__ pulls torchvision).
train.py cannot be imported here (mlflow/seaborn/datasets absent), so the
optimizer fixtures call torch.optim with exactly the arguments of
train.py:426-442 and the criterion fixtures call nn.CrossEntropyLoss with the
arguments of train.py:454-459.

Only inputs that cannot be regenerated and expected OUTPUTS are stored; weights
and inputs come from the seeded rules in oracle/synth.py.  Big tensors are stored
channel-strided (see ``pack``).
"""
import argparse
import importlib.util
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.synth import (ArchCfg, synth_images, synth_labels, synth_state_dict,  # noqa: E402
                          synth_tensor)

MAX_ELEMS = 160_000


def pack(out, name, t):
    """Store tensor ``t``; tensors above MAX_ELEMS keep every ``step``-th channel
    (dim 1) and record the step as ``<name>__cstep``."""
    a = t.detach().cpu().numpy()
    step = 1
    if a.ndim >= 2:
        while a[:, ::step].size > MAX_ELEMS and step < a.shape[1]:
            step *= 2
    out[name] = np.ascontiguousarray(a[:, ::step]) if step > 1 else a
    out[name + "__cstep"] = np.int64(step)


def import_reference(ref):
    for name in ["src", "src.utils", "src.utils.ext_transforms", "src.datasets", "cv2"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["src.utils"].ext_transforms = sys.modules["src.utils.ext_transforms"]
    sys.modules["src.datasets"].FeatureVisDataset = object
    sys.path.insert(0, ref)
    import network.modeling as modeling          # noqa: F401
    import network._deeplab as deeplab
    import network.backbone.resnet as resnet
    spec = importlib.util.spec_from_file_location("ref_loss", os.path.join(ref, "utils", "loss.py"))
    loss = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(loss)
    return modeling, deeplab, resnet, loss


def load_synth(module, prefix, salt=0):
    """Fill a reference module from the per-key rule (key = prefix + local key)."""
    sd = OrderedDict()
    for k, v in module.state_dict().items():
        sd[k] = synth_tensor(prefix + k, tuple(v.shape), salt)
    module.load_state_dict(sd, strict=True)
    return module


def grads_of(module, out, names):
    for n, p in module.named_parameters():
        if p.grad is None:
            continue
        if p.dim() == 4 and p.numel() > 20000:
            out["grad." + n + "__first8"] = p.grad[:8].detach().numpy().copy()
        else:
            out["grad." + n] = p.grad.detach().numpy().copy()


def upstream(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def gen_aspp(deeplab, outdir):
    for tag, rates, hw in (("os16_17", [6, 12, 18], 17), ("os16_25", [6, 12, 18], 25),
                           ("os8_41", [12, 24, 36], 41)):
        out = {}
        m = load_synth(deeplab.ASPP(64, rates), "aspp.")
        m.project[3].p = 0.0
        x = synth_images(2, hw, hw, seed=11, c=64)
        m.eval()
        pack(out, "eval_out", m(x))
        m.train()
        xg = x.clone().requires_grad_(True)
        y = m(xg)
        pack(out, "train_out", y)
        (y * upstream(y.shape, 5)).sum().backward()
        pack(out, "grad_x", xg.grad)
        grads_of(m, out, None)
        for k, v in m.state_dict().items():
            if "running" in k:
                out["buf." + k] = v.numpy().copy()
        np.savez_compressed(os.path.join(outdir, "aspp_%s.npz" % tag), **out)


def gen_head(deeplab, outdir):
    out = {}
    m = load_synth(deeplab.DeepLabHeadV3Plus(64, 16, 2, [6, 12, 18]), "classifier.")
    m.aspp.project[3].p = 0.0
    low = synth_images(2, 65, 65, seed=21, c=16)
    hi = synth_images(2, 17, 17, seed=22, c=64)
    m.eval()
    pack(out, "eval_out", m({"low_level": low, "out": hi}))
    m.train()
    lg, hg = low.clone().requires_grad_(True), hi.clone().requires_grad_(True)
    y = m({"low_level": lg, "out": hg})
    pack(out, "train_out", y)
    (y * upstream(y.shape, 6)).sum().backward()
    pack(out, "grad_low", lg.grad)
    pack(out, "grad_out", hg.grad)
    grads_of(m, out, None)
    np.savez_compressed(os.path.join(outdir, "head_v3plus.npz"), **out)


BOTTLENECK_CASES = OrderedDict([
    # tag: (inplanes, planes, stride, dilation, downsample, H)
    ("s2_down", (32, 16, 2, 1, True, 33)),
    ("d2", (64, 16, 1, 2, False, 17)),
    ("d4", (64, 16, 1, 4, False, 19)),
    ("s1_down", (16, 16, 1, 1, True, 21)),
])


def gen_bottleneck(resnet, outdir):
    out = {}
    for tag, (inpl, pl, s, d, down, hw) in BOTTLENECK_CASES.items():
        ds = None
        if down:
            ds = nn.Sequential(resnet.conv1x1(inpl, pl * 4, s), nn.BatchNorm2d(pl * 4))
        m = load_synth(resnet.Bottleneck(inpl, pl, s, ds, 1, 64, d), "block.")
        x = synth_images(2, hw, hw, seed=31, c=inpl)
        m.eval()
        pack(out, tag + ".eval_out", m(x))
        m.train()
        xg = x.clone().requires_grad_(True)
        y = m(xg)
        pack(out, tag + ".train_out", y)
        (y * upstream(y.shape, 7)).sum().backward()
        pack(out, tag + ".grad_x", xg.grad)
        for n, p in m.named_parameters():
            out[tag + ".grad." + n] = p.grad.numpy().copy()
        for k, v in m.state_dict().items():
            if "running" in k:
                out[tag + ".buf." + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(outdir, "bottleneck.npz"), **out)


def gen_stem(resnet, outdir):
    out = {}
    r = resnet.resnet50(replace_stride_with_dilation=[False, False, True])
    for name in ("conv1", "bn1"):
        load_synth(getattr(r, name), "backbone.%s." % name)
    x = synth_images(2, 65, 65, seed=41).requires_grad_(True)
    r.train()
    y = r.maxpool(r.relu(r.bn1(r.conv1(x))))
    pack(out, "train_out", y)
    (y * upstream(y.shape, 8)).sum().backward()
    out["grad.conv1.weight"] = r.conv1.weight.grad.numpy().copy()
    out["grad.bn1.weight"] = r.bn1.weight.grad.numpy().copy()
    out["grad.bn1.bias"] = r.bn1.bias.grad.numpy().copy()
    pack(out, "grad_x", x.grad)
    np.savez_compressed(os.path.join(outdir, "stem.npz"), **out)


BILINEAR_CASES = [(17, 65, 8), (1, 17, 8), (33, 129, 4), (65, 257, 2), (129, 513, 2), (16, 64, 4),
                  (9, 65, 4)]


def gen_bilinear(outdir):
    """F.interpolate(mode='bilinear', align_corners=False) at the reference's call
    sites: network/_deeplab.py:58,141 and network/utils.py:22."""
    out = {}
    for hin, hout, c in BILINEAR_CASES:
        x = synth_images(2, hin, hin, seed=51, c=c).requires_grad_(True)
        y = F.interpolate(x, size=(hout, hout), mode="bilinear", align_corners=False)
        tag = "%d_%d" % (hin, hout)
        pack(out, tag + ".out", y)
        (y * upstream(y.shape, 9)).sum().backward()
        pack(out, tag + ".grad_x", x.grad)
    np.savez_compressed(os.path.join(outdir, "bilinear.npz"), **out)


LOSS_CASES = [("ce", None), ("wce", [1.0, 3.7])]
FOCAL_CASES = [(1.0, 0.0, True), (0.25, 2.0, True), (0.25, 2.0, False), (1.0, 0.5, True)]


def gen_loss(loss_mod, outdir):
    out = {}
    for c in (2, 5):
        logits0 = synth_images(2, 65, 65, seed=61, c=c) * 2.0
        labels = synth_labels(2, 65, 65, seed=61, p_fg=0.3, p_ignore=0.1)
        if c == 5:
            g = torch.Generator().manual_seed(62)
            labels = torch.where(labels == 255, labels, torch.randint(0, 5, labels.shape, generator=g))
        out["c%d.labels" % c] = labels.numpy().astype(np.uint8)
        for tag, w in LOSS_CASES:
            wt = None if w is None else torch.tensor((w * 3)[:c], dtype=torch.float32)
            lg = logits0.clone().requires_grad_(True)
            crit = nn.CrossEntropyLoss(weight=wt, ignore_index=255, reduction="mean")  # train.py:454-459
            val = crit(lg, labels)
            val.backward()
            out["c%d.%s.value" % (c, tag)] = val.detach().numpy()
            out["c%d.%s.grad" % (c, tag)] = lg.grad.numpy().copy()
        for alpha, gamma, avg in FOCAL_CASES:
            for tag, w in LOSS_CASES:
                wt = None if w is None else torch.tensor((w * 3)[:c], dtype=torch.float32)
                lg = logits0.clone().requires_grad_(True)
                crit = loss_mod.FocalLoss(alpha=alpha, gamma=gamma, size_average=avg,
                                          ignore_index=255, weight=wt)  # utils/loss.py:14-35
                val = crit(lg, labels)
                val.backward()
                key = "c%d.focal_a%g_g%g_%s_%s" % (c, alpha, gamma, "mean" if avg else "sum", tag)
                out[key + ".value"] = val.detach().numpy()
                out[key + ".grad"] = lg.grad.numpy().copy()
    np.savez_compressed(os.path.join(outdir, "loss.npz"), **out)


MODEL_CASES = [("r50_os16", "resnet50", 16), ("r101_os8", "resnet101", 8)]
# input [N, H, W] per case.  r101 / os8 runs on 4 x 97 x 97 (13 x 13 maps: 676 values per BatchNorm channel): on 2 x 65 x 65
# (9 x 9 maps, 162 values) train-mode BatchNorm through 100 layers is so ill-conditioned that the reference's own fp32
# result sits 8.6e-4 from the float64 evaluation of the same graph -- no fp32 path can be held to 1e-3 against it.
MODEL_INPUT = {"r50_os16": (2, 65, 65), "r101_os8": (4, 97, 97)}


# r101 / os8 additionally damps the residual branches (every bn3.weight x 0.3, as in a trained or zero-init-residual
# ResNet).  With the synthetic rule's gamma in [0.5, 1.5] on all 33 residual branches a rounding error grows ~1e4-fold
# through the 100 train-mode BatchNorm layers, WHATEVER the map size: the reference's own fp32 logits sit 9.5e-4 (4 x 97 x 97)
# and 8.6e-4 (2 x 65 x 65) from the float64 evaluation of the same graph; with the damping 3.9e-5.
MODEL_BN3_DAMP = {"r50_os16": 1.0, "r101_os8": 0.3}


def model_state(tag, cfg):
    """the state_dict of a whole-model case (oracle.synth's per-key rule, plus the case's residual damping)"""
    sd = synth_state_dict(cfg)
    damp = MODEL_BN3_DAMP[tag]
    if damp != 1.0:
        for k in sd:
            if k.endswith(".bn3.weight"):
                sd[k] = sd[k] * damp
    return sd


def model_input(tag):
    """(images, labels) of a whole-model case -- the tests rebuild them with the same seeded rule"""
    n, h, w = MODEL_INPUT[tag]
    return synth_images(n, h, w, seed=71), synth_labels(n, h, w, seed=71, p_fg=0.2, p_ignore=0.05)
WATCH = ["backbone.conv1.weight", "backbone.layer1.0.conv2.weight", "backbone.layer4.2.conv2.weight",
         "classifier.aspp.convs.2.0.weight", "classifier.classifier.0.weight",
         "classifier.classifier.6.weight", "classifier.classifier.6.bias",
         "backbone.bn1.weight", "backbone.layer3.1.bn2.weight", "classifier.aspp.convs.4.2.bias",
         "classifier.project.1.weight"]


def gen_model(modeling, outdir):
    for tag, backbone, os_ in MODEL_CASES:
        out = {}
        cfg = ArchCfg("deeplabv3plus", backbone, 2, os_)
        m = modeling._segm_resnet("deeplabv3plus", backbone, 2, os_, False)
        m.load_state_dict(model_state(tag, cfg), strict=True)
        m.classifier.aspp.project[3].p = 0.0
        x, labels = model_input(tag)
        out["labels"] = labels.numpy().astype(np.uint8)
        m.eval()
        with torch.no_grad():
            lg = m(x)
        out["eval_logits"] = lg.numpy().copy()
        out["eval_mask"] = lg.max(1)[1].numpy().astype(np.uint8)       # train.py:644,659
        m.train()
        lg = m(x)
        out["train_logits"] = lg.detach().numpy().copy()
        crit = nn.CrossEntropyLoss(weight=torch.tensor([1.0, 3.0]), ignore_index=255, reduction="mean")
        loss = crit(lg, labels)
        loss.backward()
        out["loss"] = loss.detach().numpy()
        params = dict(m.named_parameters())
        for k in WATCH:
            g = params[k].grad
            out["grad." + k] = (g[:8] if g.dim() == 4 and g.numel() > 20000 else g).numpy().copy()
        sd = m.state_dict()
        for k in ("backbone.bn1.running_mean", "backbone.bn1.running_var",
                  "backbone.layer4.2.bn3.running_var", "classifier.aspp.convs.4.2.running_mean",
                  "classifier.classifier.4.running_var"):
            out["buf." + k] = sd[k].numpy().copy()
        np.savez_compressed(os.path.join(outdir, "model_%s.npz" % tag), **out)


def gen_model_v3(modeling, outdir):
    """the other constructor path of _segm_resnet (network/modeling.py:25-56): DeepLabV3 head (name='deeplabv3',
    ASPP -> 3x3 -> 1x1, network/_deeplab.py:71-93) on a 5-channel stem (in_channels != 3, modeling.py:25-43)."""
    out = {}
    cfg = ArchCfg("deeplabv3", "resnet50", 2, 16, in_channels=5)
    m = modeling._segm_resnet("deeplabv3", "resnet50", 2, 16, False, in_channels=5)
    sd = synth_state_dict(cfg)
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd, strict=True)
    m.classifier.classifier[0].project[3].p = 0.0
    x = synth_images(4, 65, 65, seed=73, c=5)
    labels = synth_labels(4, 65, 65, seed=73, p_fg=0.2, p_ignore=0.05)
    out["labels"] = labels.numpy().astype(np.uint8)
    out["keys"] = np.array(list(sd.keys()))
    m.eval()
    with torch.no_grad():
        lg = m(x)
    out["eval_logits"] = lg.numpy().copy()
    out["eval_mask"] = lg.max(1)[1].numpy().astype(np.uint8)
    m.train()
    lg = m(x)
    out["train_logits"] = lg.detach().numpy().copy()
    loss = nn.CrossEntropyLoss(weight=torch.tensor([1.0, 3.0]), ignore_index=255, reduction="mean")(lg, labels)
    loss.backward()
    out["loss"] = loss.detach().numpy()
    params = dict(m.named_parameters())
    for k in ("backbone.conv1.weight", "classifier.classifier.1.weight", "classifier.classifier.4.weight",
              "classifier.classifier.4.bias", "classifier.classifier.0.project.0.weight"):
        g = params[k].grad
        out["grad." + k] = (g[:8] if g.dim() == 4 and g.numel() > 20000 else g).numpy().copy()
    np.savez_compressed(os.path.join(outdir, "model_v3_in5.npz"), **out)


OPTIM_SHAPES = [("conv1.weight", (64, 3, 7, 7)), ("bn.weight", (256,)), ("proj.weight", (48, 256, 1, 1)),
                ("odd", (1001,))]
OPTIM_STEPS = 3


def optim_inputs():
    params = OrderedDict((k, synth_tensor("optim." + k, s)) for k, s in OPTIM_SHAPES)
    grads = [OrderedDict((k, synth_tensor("optim.grad%d." % t + k, s, salt=t + 1) * 0.05)
                         for k, s in OPTIM_SHAPES) for t in range(OPTIM_STEPS)]
    return params, grads


def gen_optim(outdir):
    """torch.optim built with the arguments of setup_optimizer (train.py:421-444) and
    CosineAnnealingLR of setup_scheduler (train.py:446-452, --lr 0.01 default,
    T_max=10 here) stepped once per iteration as train.py:1103 does."""
    out = {}
    ctors = (("sgd", lambda p: torch.optim.SGD(p, momentum=0.9, weight_decay=1e-4, nesterov=True)),
             ("adam", lambda p: torch.optim.Adam(p, weight_decay=1e-4)),
             ("adamw", lambda p: torch.optim.AdamW(p, weight_decay=1e-4)))
    for oname, ctor in ctors:
        params, grads = optim_inputs()
        ps = [v.clone().requires_grad_(True) for v in params.values()]
        opt = ctor(ps)
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=0.01 * 0.01)
        for t in range(OPTIM_STEPS):
            for p, g in zip(ps, grads[t].values()):
                p.grad = g.clone()
            opt.step()
            sch.step()
        for k, p in zip(params, ps):
            out["%s.%s" % (oname, k)] = p.detach().numpy().copy()
        out["%s.lr" % oname] = np.float64(opt.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(outdir, "optim.npz"), **out)


SEPCONV_CASES = OrderedDict([
    # tag: (cin, cout, k, stride, padding, dilation, bias, H, W)
    ("k3_d2_bias", (16, 24, 3, 1, 2, 2, True, 13, 11)),
    ("k3_s2_nobias", (32, 16, 3, 2, 1, 1, False, 17, 17)),
    ("k5_d1_bias", (8, 12, 5, 1, 2, 1, True, 9, 14)),
])


def gen_sepconv(deeplab, outdir):
    """AtrousSeparableConvolution (network/_deeplab.py:95-119) and convert_to_separable_conv (:176-188) applied to
    an ASPPConv (conv -> BN -> ReLU with the conv replaced by depthwise + pointwise)."""
    out = {}
    for tag, (cin, cout, k, s, p, d, bias, h, w) in SEPCONV_CASES.items():
        m = load_synth(deeplab.AtrousSeparableConvolution(cin, cout, k, s, p, d, bias), "sep.%s." % tag)
        x = synth_images(2, h, w, seed=31, c=cin)
        xg = x.clone().requires_grad_(True)
        y = m(xg)
        pack(out, tag + ".out", y)
        (y * upstream(y.shape, 7)).sum().backward()
        pack(out, tag + ".grad_x", xg.grad)
        for n, prm in m.named_parameters():
            out["%s.grad.%s" % (tag, n)] = prm.grad.detach().numpy().copy()
    m = deeplab.convert_to_separable_conv(deeplab.ASPPConv(32, 16, 3))
    assert isinstance(m[0], deeplab.AtrousSeparableConvolution)
    m = load_synth(m, "sep.asppconv.")
    x = synth_images(4, 19, 19, seed=32, c=32)
    m.eval()
    pack(out, "asppconv.eval_out", m(x))
    m.train()
    xg = x.clone().requires_grad_(True)
    y = m(xg)
    pack(out, "asppconv.train_out", y)
    (y * upstream(y.shape, 8)).sum().backward()
    pack(out, "asppconv.grad_x", xg.grad)
    for n, prm in m.named_parameters():
        out["asppconv.grad.%s" % n] = prm.grad.detach().numpy().copy()
    for kk, v in m.state_dict().items():
        if "running" in kk:
            out["asppconv.buf." + kk] = v.numpy().copy()
    out["asppconv.keys"] = np.array(list(m.state_dict().keys()))
    np.savez_compressed(os.path.join(outdir, "sepconv.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--skip-model", action="store_true")
    ap.add_argument("--only", default=None, help="generate just this fixture group (e.g. sepconv)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    modeling, deeplab, resnet, loss = import_reference(args.ref)
    if args.only == "sepconv":
        gen_sepconv(deeplab, args.out)
        return
    if args.only == "model":
        gen_model(modeling, args.out)
        return
    if args.only == "model_v3":
        gen_model_v3(modeling, args.out)
        return
    gen_sepconv(deeplab, args.out)
    gen_aspp(deeplab, args.out)
    gen_head(deeplab, args.out)
    gen_bottleneck(resnet, args.out)
    gen_stem(resnet, args.out)
    gen_bilinear(args.out)
    gen_loss(loss, args.out)
    gen_optim(args.out)
    if not args.skip_model:
        gen_model(modeling, args.out)
        gen_model_v3(modeling, args.out)
    for f in sorted(os.listdir(args.out)):
        print("%-28s %8.1f KB" % (f, os.path.getsize(os.path.join(args.out, f)) / 1024))


if __name__ == "__main__":
    main()
