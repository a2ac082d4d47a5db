"""Per-kernel parity of libiswm_hip.so (through the C ABI / ctypes wrappers) against
stock fp32 ATen ops on the CPU and the reference-generated golden vectors.

Tolerance: BASELINE.json's 1e-3 relative fp32 (tests/util.RTOL) is the contract; single
kernels are held to a much tighter bound, written next to each check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import load, rel_err

pytestmark = pytest.mark.gpu


def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU path")
    return torch.device("cuda:0")


def nhwc(t, cp=None):
    """CPU NCHW -> CUDA NHWC (optionally channel-padded)."""
    n, c, h, w = t.shape
    cp = cp or c
    out = torch.zeros(n, h, w, cp)
    out[..., :c] = t.permute(0, 2, 3, 1)
    return out.to(dev())


def nchw(t, c=None):
    return t.detach().cpu().permute(0, 3, 1, 2)[:, :c].contiguous()


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


CONV_CASES = [
    # cin, cout, k, stride, pad, dil, h, w, n
    (64, 64, 1, 1, 0, 1, 17, 19, 2),
    (64, 256, 1, 1, 0, 1, 33, 33, 2),
    (256, 64, 1, 2, 0, 1, 33, 33, 2),       # downsample 1x1 stride 2
    (32, 32, 3, 1, 1, 1, 23, 21, 2),
    (64, 64, 3, 2, 1, 1, 33, 33, 2),        # strided 3x3
    (64, 128, 3, 1, 2, 2, 17, 17, 2),       # dilation 2
    (64, 64, 3, 1, 4, 4, 19, 19, 1),        # dilation 4
    (64, 256, 3, 1, 6, 6, 33, 33, 2),       # ASPP rates
    (64, 256, 3, 1, 12, 12, 33, 33, 2),
    (64, 256, 3, 1, 18, 18, 17, 17, 2),     # padding-dominated: only the centre tap is in bounds
    (64, 256, 3, 1, 36, 36, 41, 41, 1),
    (304, 256, 3, 1, 1, 1, 21, 21, 1),      # decoder: Cin not a multiple of 32
    (256, 48, 1, 1, 0, 1, 21, 21, 2),       # low-level projection: Cout = 48
    (4, 64, 7, 2, 3, 1, 65, 65, 2),         # stem on the 4-channel padded image (bf16x6: csrc/conv_stem.hip)
    (4, 64, 7, 2, 3, 1, 30, 37, 3),         # ... even height, ragged last wave-tile
    (256, 4, 1, 1, 0, 1, 33, 33, 2),        # classifier padded to 4 classes
    (2048, 256, 1, 1, 0, 1, 1, 1, 4),       # image-pooling branch: 1x1 spatial
    (128, 512, 1, 1, 0, 1, 40, 40, 3),
    (1280, 256, 1, 1, 0, 1, 17, 17, 2),     # ASPP projection: dgrad has 10 N-tiles x 5 M-tiles
    (1280, 256, 1, 1, 0, 1, 25, 25, 2),
    (512, 64, 1, 1, 0, 1, 17, 17, 2),
    # stride-1 KxK with 32-aligned channels: the halo-patch bf16x6 kernel (patch shapes 11x11, 12x10, edge patches)
    (64, 64, 3, 1, 1, 1, 33, 33, 3),
    (96, 160, 3, 1, 1, 1, 40, 29, 2),
    (64, 48, 3, 1, 0, 1, 21, 37, 2),        # pad 0: output smaller than input
    (32, 64, 3, 1, 2, 2, 50, 50, 1),        # dilation 2
    (64, 32, 5, 1, 2, 1, 23, 23, 2),        # 5x5
]


@pytest.fixture
def conv_math(request):
    """run a test under one conv arithmetic (0 = exact fp32 MFMA, 1 = bf16x6 with pre-packed weight fragments,
    2 = bf16x6 through the plain-weight entry points) and restore the default"""
    from iswm_amd import _lib, ops
    lib = _lib.load()
    old, old_pk = lib.iswm_get_conv_math(), ops._USE_PACKED
    lib.iswm_set_conv_math(min(request.param, 1))
    ops._USE_PACKED = request.param == 1
    yield request.param
    lib.iswm_set_conv_math(old)
    ops._USE_PACKED = old_pk


@pytest.mark.parametrize("conv_math", [0, 1, 2], indirect=True, ids=["f32mfma", "bf16x6", "bf16x6-plainw"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "cin%d_cout%d_k%d_s%d_d%d_h%d" % (c[0], c[1], c[2], c[3], c[5], c[6]))
def test_conv_fwd_dgrad_wgrad(case, conv_math):
    from iswm_amd import ops
    cin, cout, k, s, p, d, h, w, n = case
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, k, k, seed=2) * (2.0 / (cin * k * k)) ** 0.5
    xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, s, p, d)
    dy = rnd(*y_ref.shape, seed=3)
    y_ref.backward(dy)

    xh = nhwc(x)
    w_ohwi = wt.permute(0, 2, 3, 1).contiguous().to(dev())
    g = ops.ConvGeom(xh, cout, k, k, s, p, d)
    y, partials, tiles = ops.conv2d_fwd(xh, w_ohwi, g, want_stats=True)
    assert rel_err(nchw(y), y_ref) < 2e-5
    # fused BN statistics: per-tile {sum, centred M2} merge to the batch mean / variance
    coef = ops.bn_finalize(partials, tiles[0], n * g.ho * g.wo, tiles[1], None, None, None, None, 0.1)
    yr = y_ref.detach().double()
    assert rel_err(coef[2], yr.mean((0, 2, 3))) < 1e-5
    if n * g.ho * g.wo > 1:
        assert rel_err(1.0 / coef[3].double().cpu() ** 2 - 1e-5, yr.var((0, 2, 3), unbiased=False)) < 1e-5
    dyh = nhwc(dy)
    dx = ops.conv2d_dgrad(dyh, w_ohwi, g, tuple(xh.shape))
    assert rel_err(nchw(dx), xr.grad) < 2e-5
    dx2 = ops.conv2d_dgrad(dyh, w_ohwi, g, tuple(xh.shape), dx=dx.clone(), accumulate=True)
    assert rel_err(nchw(dx2), 2 * xr.grad) < 2e-5
    dw = ops.conv2d_wgrad(xh, dyh, g)
    assert rel_err(dw.cpu().permute(0, 3, 1, 2), wr.grad) < 5e-5


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[0] % 32 == 0],
                         ids=lambda c: "cin%d_cout%d_k%d_s%d_d%d_h%d" % (c[0], c[1], c[2], c[3], c[5], c[6]))
def test_conv_bf16_mixed_precision(case):
    """conv math 2 ("bf16", BASELINE configs[4]): operands rounded to nearest bf16, ONE MFMA per product, fp32
    accumulation.  The oracle is therefore torch's fp32 conv on bf16-ROUNDED operands (products of bf16 values are exact
    in fp32), for the forward, the data gradient and the weight gradient -- tight tolerances, not bf16-sized ones."""
    from iswm_amd import _lib, ops
    cin, cout, k, s, p, d, h, w, n = case
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, k, k, seed=2) * (2.0 / (cin * k * k)) ** 0.5
    r = lambda t: t.bfloat16().float()                       # round to nearest even, as the kernels do
    xr, wr = r(x).requires_grad_(True), r(wt).requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, s, p, d)
    dy = rnd(*y_ref.shape, seed=3)
    y_ref.backward(r(dy))
    lib = _lib.load()
    old = lib.iswm_get_conv_math()
    lib.iswm_set_conv_math(2)
    try:
        xh = nhwc(x)
        w_ohwi = wt.permute(0, 2, 3, 1).contiguous().to(dev())
        g = ops.ConvGeom(xh, cout, k, k, s, p, d)
        y, partials, tiles = ops.conv2d_fwd(xh, w_ohwi, g, want_stats=True)
        assert rel_err(nchw(y), y_ref) < 2e-5
        coef = ops.bn_finalize(partials, tiles[0], n * g.ho * g.wo, tiles[1], None, None, None, None, 0.1)
        assert rel_err(coef[2], y_ref.detach().double().mean((0, 2, 3))) < 1e-5
        dyh = nhwc(dy)
        dx = ops.conv2d_dgrad(dyh, w_ohwi, g, tuple(xh.shape))
        if cout % 32 == 0:                                   # otherwise the data gradient runs on the fp32 kernels
            assert rel_err(nchw(dx), xr.grad) < 2e-5
        dw = ops.conv2d_wgrad(xh, dyh, g)
        assert rel_err(dw.cpu().permute(0, 3, 1, 2), wr.grad) < 5e-5
        # and it IS reduced precision: measurably different from the fp32 product
        y32 = F.conv2d(x, wt, None, s, p, d)
        assert 1e-4 < rel_err(nchw(y), y32) < 3e-2
    finally:
        lib.iswm_set_conv_math(old)


@pytest.mark.parametrize("conv_math", [0, 1, 2], indirect=True, ids=["f32mfma", "bf16x6", "bf16x6-plainw"])
def test_conv_pitched_slices_and_bias(conv_math):
    """reads a channel slice of a wider buffer and writes into a slice (torch.cat elimination)"""
    from iswm_amd import ops
    n, h, w = 2, 19, 17
    big = rnd(n, 96, h, w, seed=4)
    wt = rnd(48, 32, 3, 3, seed=5) * 0.1
    bias = rnd(48, seed=6)
    y_ref = F.conv2d(big[:, 32:64], wt, bias, 1, 1, 1)
    bigh = nhwc(big)
    out = torch.full((n, h, w, 112), 7.0, device=dev())
    xs, os_ = bigh[..., 32:64], out[..., 64:112]
    g = ops.ConvGeom(xs, 48, 3, 3, 1, 1, 1)
    ops.conv2d_fwd(xs, wt.permute(0, 2, 3, 1).contiguous().to(dev()), g, bias=bias.to(dev()), out=os_)
    assert rel_err(nchw(out[..., 64:112]), y_ref) < 2e-5
    assert float(out[..., :64].min()) == 7.0 and float(out[..., :64].max()) == 7.0


@pytest.mark.parametrize("c,m_shape", [(64, (2, 17, 19)), (48, (2, 9, 9)), (256, (4, 1, 1)), (304, (1, 13, 11)),
                                       (2048, (2, 5, 5)), (4, (2, 33, 33))])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False), (6, False), (6, True)])
def test_batchnorm_train_fwd_bwd(c, m_shape, relu, res):
    from iswm_amd import ops
    n, h, w = m_shape
    y = rnd(n, c, h, w, seed=7) * 2 + 0.5
    gamma, beta = rnd(c, seed=8) * 0.3 + 1, rnd(c, seed=9) * 0.1
    rm, rv = rnd(c, seed=10) * 0.1, rnd(c, seed=11).abs() + 0.5
    resid = rnd(n, c, h, w, seed=12) if res else None
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = resid.clone().requires_grad_(True) if res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    o_ref = F.batch_norm(yr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    if res:
        o_ref = o_ref + rr
    if relu == 6 and relu is not True:
        o_ref = F.relu6(o_ref)                                # nn.ReLU6: clamp to [0, 6]
    elif relu:
        o_ref = F.relu(o_ref)
    dout = rnd(n, c, h, w, seed=13)
    o_ref.backward(dout)

    yh = nhwc(y)
    partials, tiles, tile_rows = ops.colstat(yh)
    d = dev()
    gd, bd, rmd, rvd = gamma.to(d), beta.to(d), rm.to(d), rv.to(d)
    coef = ops.bn_finalize(partials, tiles, n * h * w, tile_rows, gd, bd, rmd, rvd, 0.1, 1e-5)
    assert rel_err(rmd, rm_ref) < 1e-5 and rel_err(rvd, rv_ref) < 1e-5
    rh = nhwc(resid) if res else None
    o = ops.bn_apply(yh, coef, relu, rh)
    assert rel_err(nchw(o), o_ref) < 1e-5
    dg, db = torch.empty(c, device=d), torch.empty(c, device=d)
    dy, dres = ops.bn_backward(nhwc(dout), o if relu else None, yh, coef, gd, relu, True, dg, db, want_dres=res)
    assert rel_err(nchw(dy), yr.grad) < 2e-4
    assert rel_err(dg, gr.grad) < 1e-4 and rel_err(db, br.grad) < 1e-4
    if res:
        assert rel_err(nchw(dres), rr.grad) < 1e-6


def test_batchnorm_eval():
    from iswm_amd import ops
    c = 64
    y = rnd(2, c, 9, 9, seed=1)
    gamma, beta, rm, rv = rnd(c, seed=2) + 1, rnd(c, seed=3), rnd(c, seed=4) * 0.1, rnd(c, seed=5).abs() + 0.5
    o_ref = F.relu(F.batch_norm(y, rm, rv, gamma, beta, False, 0.1, 1e-5))
    d = dev()
    coef = ops.bn_eval_coeffs(gamma.to(d), beta.to(d), rm.to(d), rv.to(d), 1e-5)
    assert rel_err(nchw(ops.bn_apply(nhwc(y), coef, True)), o_ref) < 1e-5


@pytest.mark.parametrize("h,w,c", [(65, 65, 64), (33, 35, 8), (4, 5, 4), (2, 2, 4), (257, 257, 4)])
def test_maxpool(h, w, c):
    from iswm_amd import ops
    x = torch.relu(rnd(2, c, h, w, seed=1))          # post-ReLU input: many exact ties at 0
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 3, 2, 1)
    dy = rnd(*y_ref.shape, seed=2)
    y_ref.backward(dy)
    xh = nhwc(x)
    y, idx = ops.maxpool_fwd(xh)
    assert torch.equal(nchw(y), y_ref.detach())
    dx = ops.maxpool_bwd(nhwc(dy), idx, tuple(xh.shape))
    assert rel_err(nchw(dx), xr.grad) < 1e-6


def test_global_pool_and_broadcast():
    from iswm_amd import ops
    x = rnd(3, 2048, 5, 7, seed=1)
    xh = nhwc(x)
    p = ops.gap_fwd(xh)
    assert rel_err(nchw(p), F.adaptive_avg_pool2d(x, 1)) < 1e-5
    out = torch.zeros(3, 5, 7, 64 + 2048, device=dev())
    ops.bcast_fwd(p, out[..., 64:])
    assert rel_err(nchw(out[..., 64:]), F.adaptive_avg_pool2d(x, 1).expand(-1, -1, 5, 7)) < 1e-5
    dy = rnd(3, 2048, 5, 7, seed=2)
    dv = ops.bcast_bwd(nhwc(dy))
    assert rel_err(nchw(dv), dy.sum((2, 3), keepdim=True)) < 1e-5
    dx = torch.ones(3, 5, 7, 2048, device=dev())
    ops.gap_bwd(dv, dx, True)
    assert rel_err(nchw(dx), 1 + dy.sum((2, 3), keepdim=True).expand(-1, -1, 5, 7) / 35) < 1e-5


def test_bilinear_golden():
    """F.interpolate(bilinear, align_corners=False) -- vectors from the reference call sites"""
    from iswm_amd import ops
    from oracle.make_golden import BILINEAR_CASES, upstream
    from oracle.synth import synth_images
    fx = load("bilinear.npz")
    for hin, hout, c in BILINEAR_CASES:
        tag = "%d_%d" % (hin, hout)
        x = synth_images(2, hin, hin, seed=51, c=c)
        cp = (c + 3) // 4 * 4                      # NHWC kernels work on channel groups of 4
        y = ops.bilinear_fwd(nhwc(x, cp), hout, hout)
        step = int(fx[tag + ".out__cstep"])
        assert rel_err(nchw(y, c)[:, ::step], fx[tag + ".out"]) < 1e-5, tag
        dy = upstream((2, c, hout, hout), 9)
        dx = ops.bilinear_bwd(nhwc(dy, cp), hin, hin)
        step = int(fx[tag + ".grad_x__cstep"])
        assert rel_err(nchw(dx, c)[:, ::step], fx[tag + ".grad_x"]) < 1e-5, tag
        # fused NHWC -> NCHW variant on the first 2 channels (the logits path)
        y2 = ops.bilinear_to_nchw_fwd(nhwc(x, cp), 2, hout, hout)
        assert rel_err(y2, F.interpolate(x[:, :2], size=(hout, hout), mode="bilinear", align_corners=False)) < 1e-5
        dx2 = ops.bilinear_to_nchw_bwd(dy[:, :2].contiguous().to(dev()), hin, hin, 4)
        xr = x[:, :2].clone().requires_grad_(True)
        F.interpolate(xr, size=(hout, hout), mode="bilinear", align_corners=False).backward(dy[:, :2])
        assert rel_err(nchw(dx2, 2), xr.grad) < 1e-5
        assert float(dx2[..., 2:].abs().max()) == 0.0


@pytest.mark.parametrize("c", [2, 5])
def test_loss_golden(c):
    """weighted CE (train.py:454-459) and FocalLoss (utils/loss.py:14-35) value + gradient"""
    from iswm_amd.utils.loss import CrossEntropyLoss, FocalLoss
    from oracle.make_golden import FOCAL_CASES, LOSS_CASES
    from oracle.synth import synth_images
    fx = load("loss.npz")
    logits0 = (synth_images(2, 65, 65, seed=61, c=c) * 2.0).to(dev())
    for label_dtype in (torch.int64, torch.uint8):
        labels = torch.from_numpy(fx["c%d.labels" % c]).to(label_dtype).to(dev())
        for tag, w in LOSS_CASES:
            wt = None if w is None else torch.tensor((w * 3)[:c], dtype=torch.float32)
            lg = logits0.clone().requires_grad_(True)
            val = CrossEntropyLoss(weight=wt, ignore_index=255)(lg, labels)
            val.backward()
            assert rel_err(val, fx["c%d.%s.value" % (c, tag)]) < 1e-5
            assert rel_err(lg.grad, fx["c%d.%s.grad" % (c, tag)]) < 1e-5
            for alpha, gamma, avg in FOCAL_CASES:
                lg = logits0.clone().requires_grad_(True)
                val = FocalLoss(alpha, gamma, avg, 255, wt)(lg, labels)
                val.backward()
                key = "c%d.focal_a%g_g%g_%s_%s" % (c, alpha, gamma, "mean" if avg else "sum", tag)
                assert rel_err(val, fx[key + ".value"]) < 1e-5, key
                assert rel_err(lg.grad, fx[key + ".grad"]) < 2e-5, key


def test_loss_edge_cases():
    from iswm_amd.utils.loss import CrossEntropyLoss
    d = dev()
    lg = torch.zeros(1, 2, 3, 3, device=d, requires_grad=True)
    lab = torch.full((1, 3, 3), 255, dtype=torch.int64, device=d)
    lab[0, 1, 1] = 1
    v = CrossEntropyLoss()(lg, lab)
    v.backward()
    assert abs(float(v) - np.log(2)) < 1e-6
    assert float(lg.grad[0, :, 0, 0].abs().max()) == 0.0            # ignored pixel: exactly zero gradient
    assert abs(float(lg.grad[0, 1, 1, 1]) + 0.5) < 1e-6


def test_argmax_bit_exact():
    from iswm_amd import ops
    lg = rnd(2, 5, 33, 31, seed=3)
    lg[:, 3] = lg[:, 1]                      # exact ties -> lowest index wins (torch.max semantics)
    out = ops.argmax_nchw(lg.to(dev()))
    assert torch.equal(out.cpu(), lg.max(1)[1])


@pytest.mark.parametrize("ldt,pdt", [(torch.uint8, torch.int64), (torch.int64, torch.int64), (torch.uint8, torch.uint8),
                                     (torch.int64, torch.uint8)])
@pytest.mark.parametrize("nc", [2, 5])
def test_confusion_matrix_bit_exact(ldt, pdt, nc):
    """StreamMetrics._fast_hist (metrics/stream_metrics.py:24-31): exact counts, 255 and negative labels ignored,
    accumulation over calls; full BASELINE mask size (16 x 513 x 513) in one of the cases"""
    import numpy as np
    from iswm_amd import ops
    from oracle import metrics as ometrics
    shape = (16, 513, 513) if (nc == 2 and ldt == torch.uint8 and pdt == torch.int64) else (3, 65, 47)
    g = torch.Generator().manual_seed(11)
    gt = torch.randint(0, nc, shape, generator=g)
    gt[torch.rand(shape, generator=g) < 0.05] = 255
    if ldt == torch.int64:
        gt[0, 0, :7] = -1                                      # numpy masks negatives too
    pr = torch.randint(0, nc, shape, generator=g)
    ref = ometrics.fast_hist(gt.numpy(), pr.numpy(), nc)
    hist = ops.confusion_matrix(gt.to(ldt).to(dev()) if ldt == torch.int64 else gt.clamp(min=0).to(ldt).to(dev()),
                                pr.to(pdt).to(dev()), nc)
    if ldt == torch.uint8:                                     # -1 is not representable: compare against the same data
        ref = ometrics.fast_hist(gt.clamp(min=0).numpy(), pr.numpy(), nc)
    assert hist.dtype == torch.int64 and np.array_equal(hist.cpu().numpy(), ref)
    ops.confusion_matrix(gt.clamp(min=0).to(torch.uint8).to(dev()), pr.to(dev()), nc, hist=hist)
    assert np.array_equal(hist.cpu().numpy(), ref + ometrics.fast_hist(gt.clamp(min=0).numpy(), pr.numpy(), nc))


def test_stream_metrics_device_vs_oracle():
    """iswm_amd.metrics.StreamMetrics: sequence / batch update semantics (:100-122), fused argmax from logits,
    result keys and formulas (:33-63)"""
    import numpy as np
    from iswm_amd.metrics import StreamMetrics
    from oracle import metrics as ometrics
    g = torch.Generator().manual_seed(12)
    logits = torch.randn(4, 2, 65, 65, generator=g)
    logits[:, 1, :3] = logits[:, 0, :3]                        # ties -> class 0
    gt = (torch.rand(4, 65, 65, generator=g) < 0.3).to(torch.uint8)
    gt[:, 10:12] = 255
    pred = logits.max(1)[1]
    m = StreamMetrics(2)
    m.update(gt.numpy(), pred.numpy(), sequence_data=True)     # a sequence contributes its last frame only
    assert np.array_equal(m.confusion_matrix, ometrics.fast_hist(gt[-1].numpy(), pred[-1].numpy(), 2))
    m.reset()
    m.update(gt.to(dev()), pred.to(dev()), sequence_data=False)
    ref = ometrics.fast_hist(gt.numpy(), pred.numpy(), 2)
    assert np.array_equal(m.confusion_matrix, ref)
    m2 = StreamMetrics(2)
    m2.update_logits(gt.to(dev()), logits.to(dev()))
    assert np.array_equal(m2.confusion_matrix, ref)
    res, want = m.get_results(), ometrics.foreground_metrics(ref)
    for key, w in zip(["MIoU", "Foreground IoU", "Precision", "Recall", "Foreground F1"], want):
        assert abs(res[key] - w) <= 1e-12, key
    with pytest.raises(TypeError):
        m.update(gt.float(), pred, sequence_data=False)


@pytest.mark.parametrize("oname", ["sgd", "adam", "adamw"])
def test_optimizers_golden(oname):
    """torch.optim arithmetic with the arguments of train.py:421-452, 3 steps + cosine LR"""
    from iswm_amd.optim import FusedAdam, FusedSGD
    from oracle.make_golden import OPTIM_STEPS, optim_inputs
    fx = load("optim.npz")
    params, grads = optim_inputs()
    ps = [torch.nn.Parameter(v.clone().to(dev())) for v in params.values()]
    ps[0].data = ps[0].data.contiguous(memory_format=torch.channels_last)
    if oname == "sgd":
        opt = FusedSGD(ps, momentum=0.9, weight_decay=1e-4, nesterov=True)
    else:
        opt = FusedAdam(ps, weight_decay=1e-4, decoupled=(oname == "adamw"))
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=0.01 * 0.01)
    for t in range(OPTIM_STEPS):
        opt.zero_grad()
        for p, g in zip(ps, grads[t].values()):
            p.grad = g.clone().to(dev())
        opt.step()
        sch.step()
    for k, p in zip(params, ps):
        assert rel_err(p, fx["%s.%s" % (oname, k)]) < 2e-6, k
    assert abs(opt.param_groups[0]["lr"] - float(fx["%s.lr" % oname])) < 1e-12


def test_dropout_statistics():
    from iswm_amd import ops
    x = torch.ones(1 << 20, device=dev())
    y, mask = ops.dropout_fwd(x, 0.1, 1234, 1)
    keep = float(mask.float().mean())
    assert abs(keep - 0.9) < 2e-3
    assert abs(float(y.mean()) - 1.0) < 3e-3
    assert torch.equal(y, mask.float() / 0.9 * x) or rel_err(y, mask.float() / 0.9) < 1e-6
    y2, mask2 = ops.dropout_fwd(x, 0.1, 1234, 2)
    assert not torch.equal(mask, mask2)
    dx = ops.dropout_bwd(torch.full_like(x, 2.0), mask, 0.1)
    assert rel_err(dx, mask.float() * 2 / 0.9) < 1e-6


def test_layout_roundtrip_and_copy():
    from iswm_amd import ops
    x = rnd(2, 3, 17, 13, seed=1)
    xh = ops.nchw_to_nhwc(x.to(dev()))
    assert xh.shape == (2, 17, 13, 4) and float(xh[..., 3].abs().max()) == 0
    assert torch.equal(ops.nhwc_to_nchw(xh, 3).cpu(), x)
    a = nhwc(rnd(2, 48, 5, 5, seed=2))
    cat = torch.zeros(2, 5, 5, 304, device=dev())
    ops.copy_channels(a, cat[..., 256:])
    assert torch.equal(cat[..., 256:], a)


def _fuzz_cases(count, seed):
    import random as _r
    rng = _r.Random(seed)
    cases = []
    while len(cases) < count:
        k = rng.choice([1, 1, 3, 3, 3, 5])
        s = rng.choice([1, 1, 1, 2])
        d = 1 if k == 1 else rng.choice([1, 1, 2, 3, 6])
        p = rng.choice([0, d * (k // 2), d * (k // 2), 1]) if k > 1 else 0
        cin = rng.choice([4, 8, 32, 32, 64, 96, 128, 160, 256, 320])
        cout = rng.choice([4, 8, 24, 32, 48, 64, 96, 128, 192, 256])
        h, w = rng.randint(1, 47), rng.randint(1, 47)
        n = rng.randint(1, 5)
        if (h + 2 * p - d * (k - 1) - 1) // s + 1 < 1 or (w + 2 * p - d * (k - 1) - 1) // s + 1 < 1:
            continue
        if n * h * w * cin * k * k > 6_000_000:
            continue
        cases.append((cin, cout, k, s, p, d, h, w, n))
    return cases


@pytest.mark.parametrize("conv_math", [0, 1], indirect=True, ids=["f32mfma", "bf16x6"])
def test_conv_geometry_fuzz(conv_math):
    """60 seeded random geometries (odd / tiny maps down to 1x1, 5x5 filters, padding smaller or larger than 'same',
    channel counts on both sides of the 32 / 64 tile edges, batch 1..5) through forward, data gradient (plain and
    accumulating) and weight gradient: every tile picker / patch planner / parity / culling branch the fixed cases may
    miss, vs torch fp32 on the CPU"""
    from iswm_amd import ops
    for case in _fuzz_cases(60, 20261004):
        cin, cout, k, s, p, d, h, w, n = case
        x = rnd(n, cin, h, w, seed=1)
        wt = rnd(cout, cin, k, k, seed=2) * (2.0 / (cin * k * k)) ** 0.5
        xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        y_ref = F.conv2d(xr, wr, None, s, p, d)
        dy = rnd(*y_ref.shape, seed=3)
        y_ref.backward(dy)
        xh = nhwc(x)
        w_ohwi = wt.permute(0, 2, 3, 1).contiguous().to(dev())
        g = ops.ConvGeom(xh, cout, k, k, s, p, d)
        y, partials, tiles = ops.conv2d_fwd(xh, w_ohwi, g, want_stats=True)
        assert rel_err(nchw(y), y_ref) < 2e-5, case
        count = n * g.ho * g.wo
        coef = ops.bn_finalize(partials, tiles[0], count, tiles[1], None, None, None, None, 0.1)
        assert rel_err(coef[2], y_ref.detach().double().mean((0, 2, 3))) < 1e-5, case
        dyh = nhwc(dy)
        dx = ops.conv2d_dgrad(dyh, w_ohwi, g, tuple(xh.shape))
        scale = max(float(xr.grad.abs().max()), 1e-30)
        assert float((nchw(dx) - xr.grad).abs().max()) <= 2e-5 * scale + 1e-12, case
        dx2 = ops.conv2d_dgrad(dyh, w_ohwi, g, tuple(xh.shape), dx=dx.clone(), accumulate=True)
        assert float((nchw(dx2) - 2 * xr.grad).abs().max()) <= 4e-5 * scale + 1e-12, case
        dw = ops.conv2d_wgrad(xh, dyh, g)
        assert rel_err(dw.cpu().permute(0, 3, 1, 2), wr.grad) < 5e-5, case


def test_loss_all_pixels_ignored_matches_torch():
    """every label == ignore_index: nn.CrossEntropyLoss(reduction='mean') divides 0 by 0 -> nan (value and gradient
    of zeros * nan); FocalLoss (mean over ALL pixels, utils/loss.py:23-35) gives 0 with a zero gradient"""
    from iswm_amd.utils.loss import CrossEntropyLoss, FocalLoss
    lg = rnd(2, 2, 9, 7, seed=4)
    lab = torch.full((2, 9, 7), 255, dtype=torch.int64)
    ref = F.cross_entropy(lg, lab, weight=torch.tensor([1.0, 3.0]), ignore_index=255)
    got = CrossEntropyLoss(weight=torch.tensor([1.0, 3.0]), ignore_index=255)(lg.to(dev()), lab.to(dev()))
    assert bool(torch.isnan(ref)) and bool(torch.isnan(got.cpu()))
    x = lg.to(dev()).requires_grad_(True)
    f = FocalLoss(alpha=0.25, gamma=2.0, ignore_index=255)(x, lab.to(dev()))
    f.backward()
    assert float(f.detach()) == 0.0 and float(x.grad.abs().max()) == 0.0


@pytest.mark.parametrize("channels_last", [False, True])
def test_weight_padding_helpers_are_bit_exact(channels_last):
    """iswm_pad_weights / iswm_unpad_weights / iswm_zero_cols: the zero-padded OHWI copy of an OIHW parameter (either
    memory format), its inverse for the gradient, and the zero channels of a concatenation buffer -- against slicing."""
    from iswm_amd import ops
    d = dev()
    for cout, cin, k, cout_p, cin_p in ((64, 3, 7, 64, 4), (2, 256, 1, 4, 256), (256, 304, 3, 256, 320), (48, 256, 1, 48, 256)):
        w = rnd(cout, cin, k, k, seed=cout + cin).to(d)
        if channels_last:
            w = w.contiguous(memory_format=torch.channels_last)
        wp = ops.pad_weights(w, cout_p, cin_p)
        ref = torch.zeros(cout_p, k, k, cin_p, device=d)
        ref[:cout, :, :, :cin] = w.permute(0, 2, 3, 1)
        assert torch.equal(wp, ref)
        g = torch.full_like(w, 7.0)
        dw = rnd(cout_p, k, k, cin_p, seed=5).to(d)
        ops.unpad_weights(dw, g)
        assert torch.equal(g, dw[:cout, :, :, :cin].permute(0, 3, 1, 2))
        assert g.stride() == w.stride()
    b = rnd(2, seed=9).to(d)
    bp = ops.pad_weights(b.view(-1, 1, 1, 1), 4, 1).view(-1)
    assert torch.equal(bp, torch.cat([b, torch.zeros(2, device=d)]))
    x = rnd(2, 5, 7, 320, seed=3).to(d)
    y = x.clone()
    ops.zero_channels(y, 304)
    assert torch.equal(y[..., :304], x[..., :304]) and not y[..., 304:].any()
    p = ops.split_planes(x)
    keep = p.t.clone()
    ops.zero_channels(p, 304)
    assert torch.equal(p.t[..., :304], keep[..., :304]) and not p.t[..., 304:].any()
    v = ops.zero_channels(ops.split_planes(x[..., :8].contiguous()), 4)          # the few-channel gradient: 4 -> 8
    assert not v.t[..., 4:].any() and torch.equal(v.t[..., :4], keep[..., :4])


def test_stem_at_production_size_vs_float64():
    """k_stem_fwd at the production map (513 x 513, more wave-tiles than the persistent grid has waves) against float64;
    the BatchNorm partials against the float64 batch statistics (network/backbone/resnet.py:137,145-146)."""
    from iswm_amd import ops
    n, h = 4, 513
    x = rnd(n, 3, h, h, seed=11)
    wt = rnd(64, 3, 7, 7, seed=12) * (2.0 / 147) ** 0.5
    y_ref = F.conv2d(x.double(), wt.double(), None, 2, 3)
    xh = nhwc(x, 4)
    w_ohwi = torch.zeros(64, 7, 7, 4)
    w_ohwi[..., :3] = wt.permute(0, 2, 3, 1)
    w_ohwi = w_ohwi.to(dev())
    g = ops.ConvGeom(xh, 64, 7, 7, 2, 3, 1)
    y, partials, tiles = ops.conv2d_fwd(xh, w_ohwi, g, want_stats=True)
    assert ops._kernel_name(g.desc(4, 64), 0).startswith("k_stem_fwd")
    assert rel_err(nchw(y), y_ref) < 2e-6
    coef = ops.bn_finalize(partials, tiles[0], n * g.ho * g.wo, tiles[1], None, None, None, None, 0.1)
    assert rel_err(coef[2], y_ref.mean((0, 2, 3))) < 1e-5
    assert rel_err(1.0 / coef[3].double().cpu() ** 2 - 1e-5, y_ref.var((0, 2, 3), unbiased=False)) < 1e-5
    # k_stem_wgrad: every workgroup of the grid takes a pixel range; partials summed in order -> bit-identical repeats
    dy = rnd(n, 64, g.ho, g.wo, seed=13)
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), wt.shape, dy.double(), 2, 3)
    dyh = nhwc(dy)
    assert ops._kernel_name(g.desc(4, 64), 2) == "k_stem_wgrad"
    dw = ops.conv2d_wgrad(xh, dyh, g)
    assert rel_err(dw[..., :3].cpu().permute(0, 3, 1, 2), dw_ref) < 2e-6
    assert not dw[..., 3].any()
    assert torch.equal(dw, ops.conv2d_wgrad(xh, dyh, g))
