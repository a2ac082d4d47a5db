import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import OrderedDict
from oracle.synth import aspp_shapes, synth_from_shapes, synth_images
from oracle.make_golden import upstream
from tests.util import rel_err
from iswm_amd.network._deeplab import ASPP
from iswm_amd.network import _hip
from iswm_amd import ops

dev = torch.device("cuda:0")
sd = synth_from_shapes(aspp_shapes("aspp", 64))
m = ASPP(64, [6, 12, 18])
m.load_state_dict(OrderedDict((k, sd["aspp." + k]) for k in m.state_dict()), strict=True)
m = m.to(dev).train()
m.project[3].p = 0.0
x = synth_images(2, 17, 17, seed=11, c=64).to(dev)
xh = ops.nchw_to_nhwc(x)
y = m.fwd(xh, True)
up = ops.nchw_to_nhwc(upstream((2, 256, 17, 17), 5).to(dev))
sink = _hip.GradSink()
dcat = m.project.bwd(up, sink)
torch.cuda.synchronize()
dcat0 = dcat.clone()
saved = {}
for i in range(4):
    c = m.convs[i]._saved[1][0]
    saved[i] = (c["out"].clone(), c["y"].clone(), c["x"].clone(), c["coef"].clone())
# reference per-branch dbeta before anything runs
ref = {}
for i in range(4):
    c = m.convs[i]._saved[1][0]
    ref[i] = (dcat[..., i*256:(i+1)*256] * (c["out"] > 0)).double().sum((0, 1, 2))
dx = None
for i in range(4):
    conv = m.convs[i]
    c = conv._saved[1][0]
    d = conv.bwd(dcat[..., i*256:(i+1)*256], sink, True, dx, dx is not None)
    dx = d if dx is None else dx
    torch.cuda.synchronize()
    print("after branch", i, "dcat same", torch.equal(dcat, dcat0),
          "dbeta err %.2e" % rel_err(conv[1].bias.grad, ref[i]))
    for j in range(4):
        cj = saved[j]
        if m.convs[j]._saved is not None:
            cc = m.convs[j]._saved[1][0]
            print("    branch", j, "out same", torch.equal(cc["out"], cj[0]), "y same", torch.equal(cc["y"], cj[1]),
                  "x same", torch.equal(cc["x"], cj[2]), "coef same", torch.equal(cc["coef"], cj[3]))
    for j in range(i + 1):
        print("    grads of", j, "dbeta err now %.2e" % rel_err(m.convs[j][1].bias.grad, ref[j]))
