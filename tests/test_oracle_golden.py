"""Pin the CPU oracle (oracle/) against vectors produced by the reference's own
code (oracle/make_golden.py -> tests/golden/).  CPU only."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import loss as oloss
from oracle import optim as ooptim
from oracle.deeplab import OracleDeepLab, argmax_mask
from oracle.make_golden import (BILINEAR_CASES, BOTTLENECK_CASES, FOCAL_CASES, LOSS_CASES, MODEL_CASES,
                                WATCH, upstream)
from oracle.synth import (ArchCfg, aspp_shapes, bottleneck_shapes, head_v3plus_shapes, synth_from_shapes,
                          synth_images, synth_labels, synth_state_dict, synth_tensor)
from tests.util import check, check_grad, load, rel_err

TOL = 2e-5   # oracle and reference run the same ATen CPU ops; only op order differs


def _oracle(sd, rates=(6, 12, 18)):
    cfg = ArchCfg(output_stride=16 if tuple(rates) == (6, 12, 18) else 8)
    return OracleDeepLab(cfg, sd, dropout_p=0.0)


@pytest.mark.parametrize("tag,rates,hw", [("os16_17", (6, 12, 18), 17), ("os16_25", (6, 12, 18), 25),
                                          ("os8_41", (12, 24, 36), 41)])
def test_aspp(tag, rates, hw):
    fx = load("aspp_%s.npz" % tag)
    o = _oracle(synth_from_shapes(aspp_shapes("aspp", 64)), rates)
    x = synth_images(2, hw, hw, seed=11, c=64)
    check(o.eval().aspp(x, "aspp"), fx, "eval_out", TOL)
    o.train()
    xg = x.clone().requires_grad_(True)
    y = o.aspp(xg, "aspp")
    check(y, fx, "train_out", TOL)
    (y * upstream(y.shape, 5)).sum().backward()
    check(xg.grad, fx, "grad_x", TOL)
    for k, p in o.named_parameters():
        check_grad(p.grad, fx, "grad." + k[len("aspp."):], 1e-4)
    for k in fx.files:
        if k.startswith("buf."):
            assert rel_err(o.sd["aspp." + k[4:]], fx[k]) <= TOL, k


def test_head_v3plus():
    fx = load("head_v3plus.npz")
    sd = synth_from_shapes(head_v3plus_shapes("classifier", 64, 16, 2))
    o = _oracle(sd)
    low = synth_images(2, 65, 65, seed=21, c=16)
    hi = synth_images(2, 17, 17, seed=22, c=64)
    check(o.eval().head({"low_level": low, "out": hi}), fx, "eval_out", TOL)
    o.train()
    lg, hg = low.clone().requires_grad_(True), hi.clone().requires_grad_(True)
    y = o.head({"low_level": lg, "out": hg})
    check(y, fx, "train_out", TOL)
    (y * upstream(y.shape, 6)).sum().backward()
    check(lg.grad, fx, "grad_low", 1e-4)
    check(hg.grad, fx, "grad_out", 1e-4)
    for k, p in o.named_parameters():
        check_grad(p.grad, fx, "grad." + k[len("classifier."):], 1e-4)


@pytest.mark.parametrize("tag", list(BOTTLENECK_CASES))
def test_bottleneck(tag):
    fx = load("bottleneck.npz")
    inpl, pl, s, d, down, hw = BOTTLENECK_CASES[tag]
    o = _oracle(synth_from_shapes(bottleneck_shapes("block", inpl, pl, down)))
    x = synth_images(2, hw, hw, seed=31, c=inpl)
    check(o.eval()._bottleneck(x, "block", s, d, down), fx, tag + ".eval_out", TOL)
    o.train()
    xg = x.clone().requires_grad_(True)
    y = o._bottleneck(xg, "block", s, d, down)
    check(y, fx, tag + ".train_out", TOL)
    (y * upstream(y.shape, 7)).sum().backward()
    check(xg.grad, fx, tag + ".grad_x", 1e-4)
    for k, p in o.named_parameters():
        check_grad(p.grad, fx, tag + ".grad." + k[len("block."):], 1e-4)
    for k in fx.files:
        if k.startswith(tag + ".buf."):
            assert rel_err(o.sd["block." + k[len(tag) + 5:]], fx[k]) <= TOL, k


def test_stem():
    fx = load("stem.npz")
    sd = {k: synth_tensor(k, s) for k, s in [("backbone.conv1.weight", (64, 3, 7, 7)),
                                              ("backbone.bn1.weight", (64,)), ("backbone.bn1.bias", (64,)),
                                              ("backbone.bn1.running_mean", (64,)),
                                              ("backbone.bn1.running_var", (64,)),
                                              ("backbone.bn1.num_batches_tracked", ())]}
    o = _oracle(sd).train()
    x = synth_images(2, 65, 65, seed=41).requires_grad_(True)
    y = F.max_pool2d(F.relu(o._bn(o._conv(x, "backbone.conv1.weight", 2, 3), "backbone.bn1")), 3, 2, 1)
    check(y, fx, "train_out", TOL)
    (y * upstream(y.shape, 8)).sum().backward()
    check(x.grad, fx, "grad_x", 1e-4)
    check_grad(o.sd["backbone.conv1.weight"].grad, fx, "grad.conv1.weight", 1e-4)
    check_grad(o.sd["backbone.bn1.weight"].grad, fx, "grad.bn1.weight", 1e-4)


@pytest.mark.parametrize("c", [2, 5])
def test_losses(c):
    fx = load("loss.npz")
    logits0 = synth_images(2, 65, 65, seed=61, c=c) * 2.0
    labels = torch.from_numpy(fx["c%d.labels" % c].astype(np.int64))
    for tag, w in LOSS_CASES:
        wt = None if w is None else torch.tensor((w * 3)[:c], dtype=torch.float32)
        lg = logits0.clone().requires_grad_(True)
        val = oloss.weighted_ce(lg, labels, wt, 255)
        val.backward()
        assert rel_err(val, fx["c%d.%s.value" % (c, tag)]) <= TOL
        assert rel_err(lg.grad, fx["c%d.%s.grad" % (c, tag)]) <= TOL
        for alpha, gamma, avg in FOCAL_CASES:
            lg = logits0.clone().requires_grad_(True)
            val = oloss.focal_loss(lg, labels, alpha, gamma, avg, 255, wt)
            val.backward()
            key = "c%d.focal_a%g_g%g_%s_%s" % (c, alpha, gamma, "mean" if avg else "sum", tag)
            assert rel_err(val, fx[key + ".value"]) <= TOL, key
            assert rel_err(lg.grad, fx[key + ".grad"]) <= 1e-4, key


def test_class_weights():
    """train.py:388-410 cannot be imported (mlflow etc. absent); formula restated and
    checked on the bench's 10 % foreground case: sqrt(0.9/0.1) = 3."""
    lab = torch.zeros(10, 10, dtype=torch.int64)
    lab[0] = 1
    w = oloss.class_weights(lab)
    assert w.dtype == torch.float32 and w[0] == 1.0 and abs(float(w[1]) - 3.0) < 1e-6


@pytest.mark.parametrize("tag,backbone,os_", MODEL_CASES)
def test_whole_model(tag, backbone, os_):
    fx = load("model_%s.npz" % tag)
    cfg = ArchCfg("deeplabv3plus", backbone, 2, os_)
    from oracle.make_golden import model_input, model_state
    o = OracleDeepLab(cfg, model_state(tag, cfg), dropout_p=0.0)
    x = model_input(tag)[0]
    labels = torch.from_numpy(fx["labels"].astype(np.int64))
    assert tuple(labels.shape) == (x.shape[0],) + tuple(x.shape[2:])
    with torch.no_grad():
        lg = o.eval()(x)
    assert rel_err(lg, fx["eval_logits"]) <= 1e-4
    mask = argmax_mask(lg).numpy()
    margin = np.abs(fx["eval_logits"][:, 1] - fx["eval_logits"][:, 0])
    sure = margin > 1e-4 * np.abs(fx["eval_logits"]).max()
    assert (mask[sure] == fx["eval_mask"][sure]).all()
    o.train()
    lg = o(x)
    assert rel_err(lg, fx["train_logits"]) <= 1e-4
    loss = oloss.weighted_ce(lg, labels, torch.tensor([1.0, 3.0]), 255)
    assert rel_err(loss, fx["loss"]) <= 1e-5
    loss.backward()
    for k in WATCH:
        check_grad(o.sd[k].grad, fx, "grad." + k, 1e-3)
    for k in fx.files:
        if k.startswith("buf."):
            assert rel_err(o.sd[k[4:]], fx[k]) <= 1e-4, k


@pytest.mark.parametrize("oname", ["sgd", "adam", "adamw"])
def test_optimizers(oname):
    """setup_optimizer + setup_scheduler arithmetic, train.py:421-452."""
    from oracle.make_golden import OPTIM_STEPS, optim_inputs
    fx = load("optim.npz")
    params, grads = optim_inputs()
    ps = [v.clone().requires_grad_(True) for v in params.values()]
    opt = {"sgd": lambda: ooptim.OracleSGD(ps), "adam": lambda: ooptim.OracleAdam(ps, decoupled=False),
           "adamw": lambda: ooptim.OracleAdam(ps, decoupled=True)}[oname]()
    for t in range(OPTIM_STEPS):
        for p, g in zip(ps, grads[t].values()):
            p.grad = g.clone()
        opt.step()
        opt.lr = ooptim.cosine_lr(1e-3, t + 1, 10, 1e-4)
    for k, p in zip(params, ps):
        assert rel_err(p, fx["%s.%s" % (oname, k)]) <= 2e-6, k
    assert math.isclose(opt.lr, float(fx["%s.lr" % oname]), rel_tol=1e-6)


def test_cosine_lr_matches_torch():
    """CosineAnnealingLR(T_max, eta_min=lr*0.01), train.py:446-452."""
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], momentum=0.9)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=50, eta_min=0.01 * 0.01)
    for t in range(1, 40):
        opt.step()
        sch.step()
        assert math.isclose(opt.param_groups[0]["lr"], ooptim.cosine_lr(1e-3, t, 50, 1e-4), rel_tol=1e-6)


def test_confusion_matrix_metrics_hand_case():
    """StreamMetrics._fast_hist / _calculate_foreground_metrics (metrics/stream_metrics.py:24-63) on a case
    worked by hand: gt = [0,0,0,1,1,1,1,255], pred = [0,1,0,1,1,0,1,1] -> TN 2, FP 1, FN 1, TP 3 (255 ignored)."""
    import numpy as np
    from oracle import metrics as ometrics
    gt = np.array([0, 0, 0, 1, 1, 1, 1, 255], dtype=np.uint8)
    pr = np.array([0, 1, 0, 1, 1, 0, 1, 1], dtype=np.int64)
    h = ometrics.fast_hist(gt, pr, 2)
    assert h.tolist() == [[2, 1], [1, 3]]
    miou, fiou, prec, rec, f1 = ometrics.foreground_metrics(h)
    assert math.isclose(fiou, 3 / 5, rel_tol=1e-6) and math.isclose(prec, 0.75, rel_tol=1e-6)
    assert math.isclose(rec, 0.75, rel_tol=1e-6) and math.isclose(f1, 0.75, rel_tol=1e-6)
    assert math.isclose(miou, (2 / 4 + 3 / 5) / 2, rel_tol=1e-6)


def test_separable_conv_golden():
    """AtrousSeparableConvolution (network/_deeplab.py:95-119) restated in oracle/deeplab.py vs the reference"""
    from oracle.deeplab import sepconv_forward
    from oracle.make_golden import SEPCONV_CASES, upstream
    from oracle.synth import synth_images, synth_tensor
    fx = load("sepconv.npz")
    for tag, (cin, cout, k, s, p, d, bias, h, w) in SEPCONV_CASES.items():
        pre = "sep.%s." % tag
        sd = {pre + "body.0.weight": synth_tensor(pre + "body.0.weight", (cin, 1, k, k)).requires_grad_(True),
              pre + "body.1.weight": synth_tensor(pre + "body.1.weight", (cout, cin, 1, 1)).requires_grad_(True)}
        if bias:
            sd[pre + "body.0.bias"] = synth_tensor(pre + "body.0.bias", (cin,)).requires_grad_(True)
            sd[pre + "body.1.bias"] = synth_tensor(pre + "body.1.bias", (cout,)).requires_grad_(True)
        x = synth_images(2, h, w, seed=31, c=cin).requires_grad_(True)
        y = sepconv_forward(x, sd, pre, s, p, d)
        assert rel_err(y, fx[tag + ".out"]) <= 2e-6, tag
        (y * upstream(y.shape, 7)).sum().backward()
        assert rel_err(x.grad, fx[tag + ".grad_x"]) <= 2e-6, tag
        for key, t in sd.items():
            assert rel_err(t.grad, fx["%s.grad.%s" % (tag, key[len(pre):])]) <= 2e-6, (tag, key)


def test_whole_model_v3_head_5_channel_stem():
    """_segm_resnet(name='deeplabv3', in_channels=5): DeepLabHead (network/_deeplab.py:71-93) and the stem surgery of
    network/modeling.py:25-43, oracle vs vectors from the reference"""
    fx = load("model_v3_in5.npz")
    cfg = ArchCfg("deeplabv3", "resnet50", 2, 16, in_channels=5)
    sd = synth_state_dict(cfg)
    assert list(sd.keys()) == [str(k) for k in fx["keys"]]
    o = OracleDeepLab(cfg, sd, dropout_p=0.0)
    x = synth_images(4, 65, 65, seed=73, c=5)
    labels = torch.from_numpy(fx["labels"].astype(np.int64))
    with torch.no_grad():
        lg = o.eval()(x)
    assert rel_err(lg, fx["eval_logits"]) <= 1e-4
    o.train()
    lg = o(x)
    assert rel_err(lg, fx["train_logits"]) <= 1e-4
    loss = oloss.weighted_ce(lg, labels, torch.tensor([1.0, 3.0]), 255)
    assert rel_err(loss, fx["loss"]) <= 1e-5
    loss.backward()
    for k in fx.files:
        if k.startswith("grad."):
            check_grad(o.sd[k[5:]].grad, fx, k, 1e-3)


def test_bf16_conv_oracle_definition():
    """oracle/deeplab.py _Bf16Conv (the checker of the product's mixed-precision conv math 2; the reference has no such
    mode): forward = fp32 conv of bf16-rounded operands, backward = the two gradient GEMMs on bf16-rounded operands --
    checked against autograd of the rounded-operand conv fed a pre-rounded upstream gradient."""
    import torch.nn.functional as F
    from oracle.deeplab import _Bf16Conv
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 9, 11, generator=g)
    w = torch.randn(48, 32, 3, 3, generator=g) * 0.1
    dy = torch.randn(2, 48, 9, 11, generator=g)
    r = lambda t: t.bfloat16().float()
    xa, wa = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = _Bf16Conv.apply(xa, wa, 1, 2, 2, True, True)
    y.backward(dy)
    xr, wr = r(x).requires_grad_(True), r(w).requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 2, 2)
    yr.backward(r(dy))
    assert torch.equal(y.detach(), yr.detach())
    assert rel_err(xa.grad, xr.grad) <= 1e-6 and rel_err(wa.grad, wr.grad) <= 1e-6
    assert 1e-4 < rel_err(y.detach(), F.conv2d(x, w, None, 1, 2, 2)) < 3e-2          # it is NOT the fp32 conv
    # flags: a GEMM the product keeps in fp32 stays in fp32
    xb, wb = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y2 = _Bf16Conv.apply(xb, wb, 1, 2, 2, False, False)
    assert torch.equal(y2.detach(), F.conv2d(x, w, None, 1, 2, 2))
    y2.backward(dy)
    assert rel_err(xb.grad, torch.nn.grad.conv2d_input(x.shape, w, dy, 1, 2, 2)) <= 1e-6


def test_oracle_pool_select_is_maxpool_for_its_own_argmax():
    """OracleDeepLab.pool_index (the HIP path's max-pool window choices, imposed in smoke() and the same-mask tests): with the
    oracle's OWN argmax as index, _pool_select must reproduce nn.MaxPool2d(3, 2, 1) (network/backbone/resnet.py:148) bit for
    bit, and route the gradient to the same pixels (a pixel that wins several overlapping windows sums their gradients in
    another order: equal to rounding)."""
    import torch.nn.functional as F
    from oracle.deeplab import _pool_select
    x = torch.randn(2, 5, 9, 11, generator=torch.Generator().manual_seed(4), requires_grad=True)
    y, ind = F.max_pool2d(x, 3, 2, 1, return_indices=True)
    ho, wo = y.shape[2:]
    oh = torch.arange(ho).view(1, 1, ho, 1)
    ow = torch.arange(wo).view(1, 1, 1, wo)
    tap = ((ind // 11 - (2 * oh - 1)) * 3 + (ind % 11 - (2 * ow - 1))).to(torch.uint8)
    ys = _pool_select(x, tap)
    assert torch.equal(ys, y)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    (gx,) = torch.autograd.grad(ys, x, g, retain_graph=True)
    (gr,) = torch.autograd.grad(y, x, g)
    assert torch.equal(gx != 0, gr != 0) and rel_err(gx, gr) < 1e-6
