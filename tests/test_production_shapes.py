"""The planes conv kernels at the geometries the headline step actually runs (deeplabv3plus_resnet101, output_stride 16,
16 x 513 x 513 per GPU -- BASELINE.json configs[2]): every distinct conv shape of that step, forward, data gradient
(plain and accumulating) and weight gradient, through the op wrappers over the C ABI, against float64 ATen on the CPU.

Why a file of its own: the tile planners (conv_pl2_pick_rbw, the narrow / wide wave layouts, the parity-ordered strided
data gradient, plan_wgrad_pl's pixel splits, the class-ordered rows of the atrous branches) choose different kernel
instantiations by GEMM size, so the small cases of tests/test_planes.py do not reach the code the benchmark times.
The CPU side stays cheap by checking a strided SUBSET of channels over ALL pixels: a planner that drops or doubles any
pixel range, tile row or K split changes every channel.

Reference shapes: network/backbone/resnet.py:99-120,176-198 (Bottleneck stages), network/_deeplab.py:36-52,121-165
(decoder, ASPP)."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err

pytestmark = pytest.mark.gpu


def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU path")
    return torch.device("cuda:0")


# n, h, w, cin, cout, k, stride, pad, dil  (cin 320 = the decoder's 304-channel concat padded to a 64-aligned K axis)
PROD = [
    (16, 129, 129, 64, 64, 1, 1, 0, 1), (16, 129, 129, 64, 64, 3, 1, 1, 1), (16, 129, 129, 64, 256, 1, 1, 0, 1),
    (16, 129, 129, 256, 64, 1, 1, 0, 1),
    (16, 129, 129, 256, 128, 1, 1, 0, 1), (16, 129, 129, 128, 128, 3, 2, 1, 1), (16, 65, 65, 128, 512, 1, 1, 0, 1),
    (16, 129, 129, 256, 512, 1, 2, 0, 1), (16, 65, 65, 512, 128, 1, 1, 0, 1), (16, 65, 65, 128, 128, 3, 1, 1, 1),
    (16, 65, 65, 512, 256, 1, 1, 0, 1), (16, 65, 65, 256, 256, 3, 2, 1, 1), (16, 33, 33, 256, 1024, 1, 1, 0, 1),
    (16, 65, 65, 512, 1024, 1, 2, 0, 1), (16, 33, 33, 1024, 256, 1, 1, 0, 1), (16, 33, 33, 256, 256, 3, 1, 1, 1),
    (16, 33, 33, 1024, 512, 1, 1, 0, 1), (16, 33, 33, 512, 512, 3, 1, 1, 1), (16, 33, 33, 512, 2048, 1, 1, 0, 1),
    (16, 33, 33, 1024, 2048, 1, 1, 0, 1), (16, 33, 33, 2048, 512, 1, 1, 0, 1), (16, 33, 33, 512, 512, 3, 1, 2, 2),
    (16, 33, 33, 2048, 256, 1, 1, 0, 1), (16, 33, 33, 2048, 256, 3, 1, 6, 6), (16, 33, 33, 2048, 256, 3, 1, 12, 12),
    (16, 33, 33, 2048, 256, 3, 1, 18, 18), (16, 33, 33, 1280, 256, 1, 1, 0, 1),
    (16, 129, 129, 256, 48, 1, 1, 0, 1), (16, 129, 129, 320, 256, 3, 1, 1, 1), (16, 129, 129, 256, 256, 3, 1, 1, 1),
    (16, 129, 129, 256, 8, 1, 1, 0, 1),
    # the verdict's cheap production-M cases (small C, > 256 tiles per launch, multi-split weight gradient)
    (4, 129, 129, 256, 256, 1, 1, 0, 1), (8, 129, 129, 128, 128, 3, 2, 1, 1),
    # output_stride 8 at 769 x 769 (configs[3]): the 97 x 97 maps and rates 12 / 24 / 36, batch 2
    (2, 97, 97, 2048, 256, 3, 1, 12, 12), (2, 97, 97, 2048, 256, 3, 1, 36, 36), (2, 97, 97, 512, 512, 3, 1, 4, 4),
]


def _ids(c):
    return "n%d_%dx%d_c%d-%d_k%d_s%d_d%d" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[8])


def _sub(c, k):
    """k channel indices spread over [0, c)"""
    k = min(k, c)
    return torch.arange(k) * (c // k) + (c // k) // 2


@pytest.mark.parametrize("case", PROD, ids=_ids)
def test_production_geometry_vs_float64(case):
    from iswm_amd import ops
    if not ops.planes_on():
        pytest.skip("planes are a bf16x6 feature")
    n, h, w, cin, cout, k, stride, pad, dil = case
    g0 = torch.Generator().manual_seed(1234 + cin + 7 * cout + 13 * k + dil)
    x = torch.randn(n, h, w, cin, generator=g0)                       # NHWC on the host
    wt = torch.randn(cout, k, k, cin, generator=g0) * (2.0 / (cin * k * k)) ** 0.5       # OHWI
    xp = ops.split_planes(x.to(dev()))
    wo = wt.to(dev())
    g = ops.ConvGeom(xp, cout, k, k, stride, pad, dil)
    up = torch.randn(n, g.ho, g.wo, cout, generator=g0)
    co, ci = _sub(cout, 8), _sub(cin, 8)
    x64 = x.permute(0, 3, 1, 2).double()
    w64 = wt.permute(0, 3, 1, 2).double()                              # OIHW
    up64 = up.permute(0, 3, 1, 2).double()

    # ---- forward (+ the fused BatchNorm partials): 8 output channels over every pixel
    yh, partials, tiles = ops.conv2d_fwd(xp, wo, g, want_stats=True)
    y_ref = F.conv2d(x64, w64[co], None, stride, pad, dil)
    assert rel_err(yh.cpu().permute(0, 3, 1, 2)[:, co], y_ref) <= 2e-5
    cnt = n * g.ho * g.wo
    ones, zeros = torch.ones(cout, device=dev()), torch.zeros(cout, device=dev())
    coef = ops.bn_finalize(partials, tiles[0], cnt, tiles[1], ones, zeros, zeros.clone(), ones.clone(), 0.1)
    yc = y_ref.permute(1, 0, 2, 3).reshape(len(co), -1)
    # the batch mean against the spread of the values it averages (a mean of 17 424 ... 266 256 roughly centred values is
    # itself ~1e-2 of that spread: relative to its own size the fp32 tile sums would be asked for more than fp32 holds)
    assert float(((coef[2].cpu()[co].double() - yc.mean(1)).abs() / yc.std(1)).max()) <= 2e-6
    assert rel_err(1.0 / coef[3].cpu()[co] ** 2, yc.var(1, unbiased=False) + 1e-5) <= 1e-4
    del yh, y_ref, yc

    # ---- data gradient, plain and accumulating: 8 input channels over every pixel
    dyp = ops.split_planes(up.to(dev()))
    dx = ops.conv2d_dgrad(dyp, wo, g, (n, h, w, cin))
    opad = (h + 2 * pad - dil * (k - 1) - 1) % stride, (w + 2 * pad - dil * (k - 1) - 1) % stride
    dx_ref = F.conv_transpose2d(up64, w64[:, ci], None, stride, pad, opad, 1, dil)
    assert tuple(dx_ref.shape[2:]) == (h, w)
    assert rel_err(dx.cpu().permute(0, 3, 1, 2)[:, ci], dx_ref) <= 2e-5
    acc = torch.ones(n, h, w, cin, device=dev())
    ops.conv2d_dgrad(dyp, wo, g, (n, h, w, cin), dx=acc, accumulate=True)
    assert rel_err(acc.cpu().permute(0, 3, 1, 2)[:, ci] - 1.0, dx_ref) <= 5e-5
    del dx, acc, dx_ref

    # ---- weight gradient: 8 output channels x every tap x 16 input channels, summed over every pixel
    dw = ops.conv2d_wgrad(xp, dyp, g)
    ci2 = _sub(cin, 16)
    ws = torch.zeros(len(co), len(ci2), k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x64[:, ci2], ws, None, stride, pad, dil).backward(up64[:, co])
    got = dw.cpu().permute(0, 3, 1, 2)[co][:, ci2]
    assert rel_err(got, ws.grad) <= 5e-5


ASPP_CASES = [  # n, h, w, cin, cout, rates
    (16, 33, 33, 2048, 256, (6, 12, 18)),      # configs[2]: the headline step's ASPP
    (2, 97, 97, 2048, 256, (12, 24, 36)),      # configs[3]: output_stride 8 at 769 x 769
    (3, 17, 19, 64, 256, (6, 12, 18)),         # ragged, padding-dominated (rate 18 > the map: only the centre tap is in bounds)
    (2, 16, 16, 128, 128, (6, 12, 18)),        # 256 x 256 tiles at output_stride 16 (configs[0] geometry), one column tile
    (6, 6, 6, 2048, 256, (6, 12, 18)),         # 81 x 81 tiles: every rate reaches past the map
]


@pytest.mark.parametrize("case", ASPP_CASES, ids=lambda c: "n%d_%dx%d_c%d-%d_d%d" % (c[0], c[1], c[2], c[3], c[4], c[5][0]))
def test_fused_aspp_branches_vs_float64(case):
    """iswm_aspp_fwd / iswm_aspp_bwd (the 1x1 + three atrous 3x3 branches of network/_deeplab.py:143-172 from one tile table
    each way) through the C ABI: every branch output and its BatchNorm partials, and the summed data gradient, against float64
    ATen on a channel subset over every pixel."""
    import ctypes
    from iswm_amd import ops
    if not ops.planes_on():
        pytest.skip("planes are a bf16x6 feature")
    n, h, w, cin, cout, rates = case
    ksize, dil = [1, 3, 3, 3], [1] + list(rates)
    g0 = torch.Generator().manual_seed(77 + cin + h)
    x = torch.randn(n, h, w, cin, generator=g0)
    ws = [torch.randn(cout, k, k, cin, generator=g0) * (2.0 / (cin * k * k)) ** 0.5 for k in ksize]
    xp = ops.split_planes(x.to(dev()))
    ldp = ops.pgeom(xp)[4]

    def pack(wt, k, dl, kind, ldx, ldy):
        d = ops.ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, dl * (k - 1) // 2, dl, ldx, ldy)
        buf = torch.empty((ops._pl2_bytes(d, kind) // 4,), dtype=torch.float32, device=dev())
        ops.call("iswm_conv2d_pl2_pack_weights", ctypes.byref(d), kind, ops._p(wt.to(dev())), ops._p(buf), ops._stream())
        return buf

    wpk_f = [pack(wt, k, dl, 0, ldp, cout) for wt, k, dl in zip(ws, ksize, dil)]
    res = ops.aspp_fwd(xp, ksize, dil, cout, wpk_f, True)
    assert res is not None, "the fused ASPP kernel must cover this geometry"
    ys, parts, tiles = res
    x64 = x.permute(0, 3, 1, 2).double()
    co, ci = _sub(cout, 8), _sub(cin, 8)
    cnt = n * h * w
    ones, zeros = torch.ones(cout, device=dev()), torch.zeros(cout, device=dev())
    for b, (wt, k, dl) in enumerate(zip(ws, ksize, dil)):
        w64 = wt.permute(0, 3, 1, 2).double()
        ref = F.conv2d(x64, w64[co], None, 1, dl * (k - 1) // 2, dl)
        assert rel_err(ys[b].cpu().permute(0, 3, 1, 2)[:, co], ref) <= 2e-5, b
        coef = ops.bn_finalize(parts[b], tiles, cnt, ops.ASPP_TILE_ROWS, ones, zeros, zeros.clone(), ones.clone(), 0.1)
        yc = ref.permute(1, 0, 2, 3).reshape(len(co), -1)
        assert float(((coef[2].cpu()[co].double() - yc.mean(1)).abs() / yc.std(1)).max()) <= 2e-6, b
        assert rel_err(1.0 / coef[3].cpu()[co] ** 2, yc.var(1, unbiased=False) + 1e-5) <= 1e-4, b
    del ys, parts

    # ---- the data gradient of the four branches as one GEMM: dx = sum_b conv^T(dy_b, w_b)
    up = torch.randn(n, h, w, 4 * cout, generator=g0)
    dyc = ops.split_planes(up.to(dev()))
    wpk_d = [pack(wt, k, dl, 1, cin, 4 * cout) for wt, k, dl in zip(ws, ksize, dil)]
    dx = ops.aspp_dgrad(dyc, ksize, dil, cin, cout, wpk_d)
    assert dx is not None
    up64 = up.permute(0, 3, 1, 2).double()
    dx_ref = 0
    for b, (wt, k, dl) in enumerate(zip(ws, ksize, dil)):
        w64 = wt.permute(0, 3, 1, 2).double()
        dx_ref = dx_ref + F.conv_transpose2d(up64[:, b * cout:(b + 1) * cout], w64[:, ci], None, 1, dl * (k - 1) // 2, 0, 1, dl)
    assert rel_err(dx.cpu().permute(0, 3, 1, 2)[:, ci], dx_ref) <= 2e-5
    # the accumulating form, with the four weight gradients in the same call (dy is the pitched concat buffer there)
    acc = torch.ones(n, h, w, cin, device=dev())
    dws = [torch.empty(cout, k, k, cin, device=dev()) for k in ksize]
    ops.aspp_dgrad(dyc, ksize, dil, cin, cout, wpk_d, dx=acc, accumulate=True, x=xp, dws=dws)
    assert rel_err(acc.cpu().permute(0, 3, 1, 2)[:, ci] - 1.0, dx_ref) <= 5e-5
    ci2 = _sub(cin, 16)
    for b, (k, dl) in enumerate(zip(ksize, dil)):
        wsub = torch.zeros(len(co), len(ci2), k, k, dtype=torch.float64, requires_grad=True)
        F.conv2d(x64[:, ci2], wsub, None, 1, dl * (k - 1) // 2, dl).backward(up64[:, b * cout:(b + 1) * cout][:, co])
        got = dws[b].cpu().permute(0, 3, 1, 2)[co][:, ci2]
        assert rel_err(got, wsub.grad) <= 5e-5, b
