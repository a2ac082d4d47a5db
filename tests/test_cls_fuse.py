"""The 1x1 classifier folded into the BatchNorm passes of the stage in front of it (csrc/bn_classify.hip; network/_deeplab.py:44-52
`classifier`): the folded kernels against float64 ATen, and the folded head against the unfolded one."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_err

pytestmark = pytest.mark.gpu


def dev():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU; the product has no CPU path")
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape,k,training", [((2, 9, 11, 256), 2, True), ((3, 33, 33, 256), 3, True), ((1, 7, 5, 256), 4, False)])
def test_folded_classifier_kernels_vs_float64(shape, k, training):
    from iswm_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(7)
    n, h, w, c = shape
    y = torch.randn(shape, generator=g)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    wc = torch.randn(k, c, generator=g) * 0.1
    bias = torch.randn(k, generator=g)
    dl = torch.randn(n, h, w, k, generator=g)
    # float64 reference: BatchNorm (batch or running statistics) -> ReLU -> 1x1 conv
    y64 = y.double().requires_grad_(True)
    g64, b64, w64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True), wc.double().requires_grad_(True)
    if training:
        mean, var = y64.mean((0, 1, 2)), y64.var((0, 1, 2), unbiased=False)
    else:
        mean, var = torch.randn(c, generator=g).double() * 0.1, (torch.rand(c, generator=g) + 0.5).double()
    inv = 1.0 / torch.sqrt(var + 1e-5)
    act = F.relu((y64 - mean) * inv * g64 + b64)
    lg = act @ w64.t() + bias.double()
    lg.backward(dl.double())
    # HIP: coefficients as bn_finalize would give them (scale, shift, mean, invstd)
    coef = torch.stack([(g64 * inv).detach().float(), beta, mean.detach().float(), inv.detach().float()]).to(d).contiguous()
    wc4 = torch.zeros(4, c)
    wc4[:k] = wc
    b4 = torch.zeros(4)
    b4[:k] = bias
    yd = y.to(d)
    logits = ops.bn_apply_classify(yd, coef, wc4.to(d), b4.to(d))
    assert rel_err(logits[..., :k].cpu(), lg.detach()) < 2e-6
    assert torch.equal(logits[..., k:].cpu(), b4[k:].expand(n, h, w, 4 - k))
    dl4 = torch.zeros(n, h, w, 4)
    dl4[..., :k] = dl
    dgamma, dbeta = torch.empty(c, device=d), torch.empty(c, device=d)
    for planes in (False, True):
        dy, dwc4 = ops.bn_backward_classify(dl4.to(d), wc4.to(d), yd, coef, gamma.to(d), training, dgamma, dbeta, planes)
        dyf = ops.as_f32(dy)
        assert rel_err(dyf.cpu(), y64.grad) < 5e-6, planes
        assert rel_err(dwc4[:k].cpu(), w64.grad) < 2e-6
        assert not dwc4[k:].any()
        assert rel_err(dgamma.cpu(), g64.grad) < 2e-6 and rel_err(dbeta.cpu(), b64.grad) < 2e-6


def test_folded_head_equals_unfolded_head():
    """DeepLabHeadV3Plus forward + backward with the classifier folded vs as a conv of its own: logits and every gradient"""
    from iswm_amd.network import _deeplab, _hip
    d = dev()
    torch.manual_seed(3)
    head = _deeplab.DeepLabHeadV3Plus(2048, 256, 2, [6, 12, 18]).to(d).train()
    head.aspp.project[3].p = 0.0                     # nn.Dropout(0.1): the two passes must see the same activations
    feats = {"low_level": torch.randn(2, 256, 33, 33, device=d), "out": torch.randn(2, 2048, 9, 9, device=d)}
    dy = torch.randn(2, 2, 33, 33, device=d)
    res = []
    for fold in (True, False):
        _hip._CLS_FUSE = fold
        for p in head.parameters():
            p.grad = None
        f = {k: v.clone().requires_grad_(True) for k, v in feats.items()}
        sd = {k: v.clone() for k, v in head.state_dict().items()}
        out = head(f)
        out.backward(dy)
        res.append((out.detach().clone(), {k: p.grad.detach().clone() for k, p in head.named_parameters()},
                    {k: v.grad.detach().clone() for k, v in f.items()}))
        head.load_state_dict(sd)                      # same running statistics for the second pass
    _hip._CLS_FUSE = True
    (o1, g1, x1), (o0, g0, x0) = res
    assert rel_err(o1, o0) < 5e-6
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 2e-5, k
    for k in x0:
        assert rel_err(x1[k], x0[k]) < 2e-5, k
