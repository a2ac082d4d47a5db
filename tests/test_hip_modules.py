"""Module- and model-level parity of the HIP path (iswm_amd.network, through the
reference's construction API) against the golden vectors produced by the reference's
own modules and against the CPU oracle on the same seeded inputs.

Tolerance: 1e-3 relative fp32 (BASELINE.json north_star), argmax masks bit-exact."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from tests.util import RTOL, check, check_grad, load, rel_err

pytestmark = pytest.mark.gpu


def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU path")
    return torch.device("cuda:0")


def load_sd(module, sd, prefix):
    own = module.state_dict()
    mapped = OrderedDict((k, sd[prefix + k]) for k in own)
    module.load_state_dict(mapped, strict=True)
    return module.to(dev())


def upstream(shape, seed):
    from oracle.make_golden import upstream as up
    return up(shape, seed).to(dev())


@pytest.mark.parametrize("tag,rates,hw", [("os16_17", (6, 12, 18), 17), ("os16_25", (6, 12, 18), 25),
                                          ("os8_41", (12, 24, 36), 41)])
def test_aspp_golden(tag, rates, hw):
    from iswm_amd.network._deeplab import ASPP
    from oracle.synth import aspp_shapes, synth_from_shapes, synth_images
    fx = load("aspp_%s.npz" % tag)
    m = load_sd(ASPP(64, list(rates)), synth_from_shapes(aspp_shapes("aspp", 64)), "aspp.")
    m.project[3].p = 0.0
    x = synth_images(2, hw, hw, seed=11, c=64).to(dev())
    m.eval()
    with torch.no_grad():
        check(m(x), fx, "eval_out")
    m.train()
    xg = x.clone().requires_grad_(True)
    y = m(xg)
    check(y, fx, "train_out")
    (y * upstream(y.shape, 5)).sum().backward()
    check(xg.grad, fx, "grad_x")
    for k, p in m.named_parameters():
        check_grad(p.grad, fx, "grad." + k)
    for k in fx.files:
        if k.startswith("buf."):
            assert rel_err(m.state_dict()[k[4:]], fx[k]) <= RTOL, k


def test_head_v3plus_golden():
    from iswm_amd.network._deeplab import DeepLabHeadV3Plus
    from oracle.synth import head_v3plus_shapes, synth_from_shapes, synth_images
    fx = load("head_v3plus.npz")
    m = load_sd(DeepLabHeadV3Plus(64, 16, 2, [6, 12, 18]),
                synth_from_shapes(head_v3plus_shapes("classifier", 64, 16, 2)), "classifier.")
    m.aspp.project[3].p = 0.0
    low = synth_images(2, 65, 65, seed=21, c=16).to(dev())
    hi = synth_images(2, 17, 17, seed=22, c=64).to(dev())
    m.eval()
    with torch.no_grad():
        check(m({"low_level": low, "out": hi}), fx, "eval_out")
    m.train()
    lg, hg = low.clone().requires_grad_(True), hi.clone().requires_grad_(True)
    y = m({"low_level": lg, "out": hg})
    check(y, fx, "train_out")
    (y * upstream(y.shape, 6)).sum().backward()
    check(lg.grad, fx, "grad_low")
    check(hg.grad, fx, "grad_out")
    for k, p in m.named_parameters():
        check_grad(p.grad, fx, "grad." + k)


def test_bottleneck_golden():
    from iswm_amd.network import _hip
    from iswm_amd.network.backbone import resnet
    from oracle.make_golden import BOTTLENECK_CASES
    from oracle.synth import bottleneck_shapes, synth_from_shapes, synth_images
    import torch.nn as nn
    fx = load("bottleneck.npz")
    for tag, (inpl, pl, s, d, down, hw) in BOTTLENECK_CASES.items():
        ds = nn.Sequential(resnet.conv1x1(inpl, pl * 4, s), _hip.BatchNorm2d(pl * 4)) if down else None
        m = load_sd(resnet.Bottleneck(inpl, pl, s, ds, 1, 64, d),
                    synth_from_shapes(bottleneck_shapes("block", inpl, pl, down)), "block.")
        x = synth_images(2, hw, hw, seed=31, c=inpl).to(dev())
        m.eval()
        with torch.no_grad():
            check(m(x), fx, tag + ".eval_out")
        m.train()
        xg = x.clone().requires_grad_(True)
        y = m(xg)
        check(y, fx, tag + ".train_out")
        (y * upstream(y.shape, 7)).sum().backward()
        check(xg.grad, fx, tag + ".grad_x")
        for k, p in m.named_parameters():
            check_grad(p.grad, fx, tag + ".grad." + k)
        for k in fx.files:
            if k.startswith(tag + ".buf."):
                assert rel_err(m.state_dict()[k[len(tag) + 5:]], fx[k]) <= RTOL, k


def _build(backbone, os_, num_classes=2):
    from iswm_amd.network import modeling
    from oracle.synth import ArchCfg, synth_state_dict
    cfg = ArchCfg("deeplabv3plus", backbone, num_classes, os_)
    m = modeling._segm_resnet("deeplabv3plus", backbone, num_classes, os_, False)
    sd = synth_state_dict(cfg)
    assert list(m.state_dict().keys()) == list(sd.keys())         # drop-in key layout (374 / 680 keys)
    m.load_state_dict(sd, strict=True)
    m.classifier.aspp.project[3].p = 0.0
    return m.to(dev()), cfg, sd


@pytest.mark.parametrize("tag,backbone,os_", [("r50_os16", "resnet50", 16), ("r101_os8", "resnet101", 8)])
def test_whole_model_golden(tag, backbone, os_):
    """logits, bit-exact argmax mask, loss, gradients and BN running stats of one training
    step on [2,3,65,65] vs the reference's _segm_resnet + nn.CrossEntropyLoss(weight)."""
    from iswm_amd import ops
    from iswm_amd.utils.loss import CrossEntropyLoss
    from oracle.make_golden import WATCH
    from oracle.synth import synth_images
    fx = load("model_%s.npz" % tag)
    m, cfg, sd = _build(backbone, os_)
    x = synth_images(2, 65, 65, seed=71).to(dev())
    labels = torch.from_numpy(fx["labels"].astype(np.int64)).to(dev())
    m.eval()
    with torch.no_grad():
        lg = m(x)
    assert rel_err(lg, fx["eval_logits"]) <= RTOL
    mask = ops.argmax_nchw(lg).cpu().numpy()
    margin = np.abs(fx["eval_logits"][:, 1] - fx["eval_logits"][:, 0])
    sure = margin > 2 * RTOL * np.abs(fx["eval_logits"]).max()       # pixels that are not near-ties
    assert sure.mean() > 0.95
    assert (mask[sure] == fx["eval_mask"][sure]).all()
    assert torch.equal(ops.argmax_nchw(lg).cpu(), lg.cpu().max(1)[1])   # the argmax kernel itself: bit-exact
    m.train()
    lg = m(x)
    assert rel_err(lg, fx["train_logits"]) <= RTOL
    loss = CrossEntropyLoss(weight=torch.tensor([1.0, 3.0]), ignore_index=255)(lg, labels)
    assert rel_err(loss, fx["loss"]) <= RTOL
    loss.backward()
    params = dict(m.named_parameters())
    for k in WATCH:
        check_grad(params[k].grad, fx, "grad." + k, 3 * RTOL)
    for k in fx.files:
        if k.startswith("buf."):
            assert rel_err(m.state_dict()[k[4:]], fx[k]) <= RTOL, k


def test_train_steps_match_oracle():
    """three full SGD-nesterov steps (train.py:1045-1049 sequence) on a reduced-depth net:
    HIP path vs the CPU oracle from the same initial state."""
    from iswm_amd.network import modeling
    from iswm_amd.optim import FusedSGD
    from iswm_amd.utils.loss import CrossEntropyLoss
    from oracle import loss as oloss
    from oracle.deeplab import OracleDeepLab
    from oracle.optim import OracleSGD
    from oracle.synth import ArchCfg, synth_images, synth_labels, synth_state_dict
    m, cfg, sd = _build("resnet50", 16)
    o = OracleDeepLab(cfg, sd, dropout_p=0.0).train()
    oopt = OracleSGD(o.parameters())
    opt = FusedSGD(m.parameters(), momentum=0.9, weight_decay=1e-4, nesterov=True)
    crit = CrossEntropyLoss(weight=torch.tensor([1.0, 3.0]))
    m.train()
    for it in range(3):
        x = synth_images(2, 49, 49, seed=100 + it)
        lab = synth_labels(2, 49, 49, seed=100 + it, p_fg=0.2, p_ignore=0.05)
        lo = oloss.weighted_ce(o(x), lab, torch.tensor([1.0, 3.0]))
        o.zero_grad()
        lo.backward()
        oopt.step()
        lg = m(x.to(dev()))
        l = crit(lg, lab.to(dev()))
        opt.zero_grad()
        l.backward()
        opt.step()
        assert rel_err(l, lo.detach()) <= RTOL, it
    osd = o.state_dict()
    msd = m.state_dict()
    worst = max(rel_err(msd[k], osd[k]) for k in osd if osd[k].is_floating_point())
    assert worst <= RTOL, worst
    assert int(msd["backbone.bn1.num_batches_tracked"]) == 3


def test_full_size_properties():
    """513x513 (BASELINE size), batch 2, resnet50: size-independent properties --
    finite logits of the right shape, mean-loss gradient sums to ~0 over classes per pixel,
    deterministic (bit-identical) repeat, and eval-mode argmax is stable under a repeat."""
    from iswm_amd import ops
    from iswm_amd.utils.loss import CrossEntropyLoss
    from oracle.synth import synth_images, synth_labels
    m, cfg, sd = _build("resnet50", 16)
    x = synth_images(2, 513, 513, seed=5).to(dev())
    lab = synth_labels(2, 513, 513, seed=5).to(dev())
    m.train()
    outs = []
    for rep in range(2):
        m.load_state_dict(sd, strict=True)
        lg = m(x)
        assert lg.shape == (2, 2, 513, 513) and bool(torch.isfinite(lg).all())
        loss = CrossEntropyLoss(weight=torch.tensor([1.0, 3.0]))(lg, lab)
        for p in m.parameters():
            p.grad = None
        loss.backward()
        outs.append((lg.detach().clone(), m.backbone.conv1.weight.grad.clone(),
                     m.classifier.classifier[6].weight.grad.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)                     # no atomics anywhere: bit-reproducible
    m.eval()
    with torch.no_grad():
        lg = m(x)
    assert torch.equal(ops.argmax_nchw(lg).cpu(), lg.cpu().max(1)[1])
