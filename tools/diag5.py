import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from collections import OrderedDict
from oracle.deeplab import OracleDeepLab
from oracle.synth import ArchCfg, aspp_shapes, synth_from_shapes, synth_images
from oracle.make_golden import upstream
from tests.util import rel_err
from iswm_amd.network._deeplab import ASPP
from iswm_amd.network import _hip
from iswm_amd import ops
dev = torch.device("cuda:0")
sd = synth_from_shapes(aspp_shapes("aspp", 64))
hw = 17
o = OracleDeepLab(ArchCfg(), sd, dropout_p=0.0).train()
x = synth_images(2, hw, hw, seed=11, c=64)
xo = x.clone().requires_grad_(True)
ap = "aspp"
res, raw = [], []
for i, r in zip((0, 1, 2, 3), (0, 6, 12, 18)):
    if i == 0:
        yc = o._conv(xo, ap + ".convs.0.0.weight")
    else:
        yc = o._conv(xo, ap + ".convs.%d.0.weight" % i, 1, r, r)
    yc.retain_grad(); raw.append(yc)
    b = F.relu(o._bn(yc, ap + ".convs.%d.1" % i)); b.retain_grad(); res.append(b)
p = F.adaptive_avg_pool2d(xo, 1)
p = F.relu(o._bn(o._conv(p, ap + ".convs.4.1.weight"), ap + ".convs.4.2"))
b4 = F.interpolate(p, size=(hw, hw), mode="bilinear", align_corners=False); b4.retain_grad(); res.append(b4)
cat = torch.cat(res, 1)
yo = F.relu(o._bn(o._conv(cat, ap + ".project.0.weight"), ap + ".project.1"))
up = upstream(yo.shape, 5)
(yo * up).sum().backward()

m = ASPP(64, [6, 12, 18])
m.load_state_dict(OrderedDict((k, sd["aspp." + k]) for k in m.state_dict()), strict=True)
m = m.to(dev).train(); m.project[3].p = 0.0
xh = ops.nchw_to_nhwc(x.to(dev))
y = m.fwd(xh, True)
sink = _hip.GradSink()
dcat = m.project.bwd(ops.nchw_to_nhwc(up.to(dev)), sink)
nchw = lambda t: t.detach().cpu().permute(0, 3, 1, 2)
for i in range(5):
    c = m.convs[i]._saved[1][0] if i < 4 else None
    print("branch", i, "dcat err %.2e" % rel_err(nchw(dcat[..., i*256:(i+1)*256]), res[i].grad), end=" ")
    if c is not None:
        print("out err %.2e  y err %.2e  mask mismatches %d" % (rel_err(nchw(c["out"]), res[i].detach()), rel_err(nchw(c["y"]), raw[i].detach()),
              int(((nchw(c["out"]) > 0) != (res[i] > 0)).sum())), end=" ")
        dg = torch.empty(256, device=dev); db = torch.empty(256, device=dev)
        dy, _ = ops.bn_backward(dcat[..., i*256:(i+1)*256], c["out"], c["y"], c["coef"], m.convs[i][1].weight, True, True, dg, db)
        print("dbeta err %.2e dy err %.2e" % (rel_err(db, o.sd[ap + ".convs.%d.1.bias" % i].grad), rel_err(nchw(dy), raw[i].grad)), end="")
        bnm = o.sd  # oracle running stats not needed
    print()
