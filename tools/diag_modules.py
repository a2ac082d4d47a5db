"""GPU diagnostic: per-tensor relative error of the HIP ASPP / head against the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import OrderedDict
from oracle.deeplab import OracleDeepLab
from oracle.synth import ArchCfg, aspp_shapes, head_v3plus_shapes, synth_from_shapes, synth_images
from oracle.make_golden import upstream
from tests.util import rel_err
from iswm_amd.network._deeplab import ASPP, DeepLabHeadV3Plus

dev = torch.device("cuda:0")

def load_sd(module, sd, prefix):
    own = module.state_dict()
    module.load_state_dict(OrderedDict((k, sd[prefix + k]) for k in own), strict=True)
    return module.to(dev)

for hw in (17, 25):
    sd = synth_from_shapes(aspp_shapes("aspp", 64))
    m = load_sd(ASPP(64, [6, 12, 18]), sd, "aspp.")
    m.project[3].p = 0.0
    o = OracleDeepLab(ArchCfg(), sd, dropout_p=0.0).train()
    x = synth_images(2, hw, hw, seed=11, c=64)
    xo = x.clone().requires_grad_(True)
    yo = o.aspp(xo, "aspp")
    up = upstream(yo.shape, 5)
    (yo * up).sum().backward()
    m.train()
    xg = x.to(dev).requires_grad_(True)
    y = m(xg)
    (y * up.to(dev)).sum().backward()
    print("ASPP hw=%d out %.2e grad_x %.2e" % (hw, rel_err(y, yo.detach()), rel_err(xg.grad, xo.grad)))
    for k, p in m.named_parameters():
        print("   %-28s %.2e  |ref| %.2e" % (k, rel_err(p.grad, o.sd["aspp." + k].grad), float(o.sd["aspp." + k].grad.abs().max())))

sd = synth_from_shapes(head_v3plus_shapes("classifier", 64, 16, 2))
m = load_sd(DeepLabHeadV3Plus(64, 16, 2, [6, 12, 18]), sd, "classifier.")
m.aspp.project[3].p = 0.0
o = OracleDeepLab(ArchCfg(), sd, dropout_p=0.0).train()
low = synth_images(2, 65, 65, seed=21, c=16)
hi = synth_images(2, 17, 17, seed=22, c=64)
lo_, ho_ = low.clone().requires_grad_(True), hi.clone().requires_grad_(True)
yo = o.head({"low_level": lo_, "out": ho_})
up = upstream(yo.shape, 6)
(yo * up).sum().backward()
m.train()
lg, hg = low.to(dev).requires_grad_(True), hi.to(dev).requires_grad_(True)
y = m({"low_level": lg, "out": hg})
(y * up.to(dev)).sum().backward()
print("HEAD out %.2e grad_low %.2e grad_out %.2e" % (rel_err(y, yo.detach()), rel_err(lg.grad, lo_.grad), rel_err(hg.grad, ho_.grad)))
for k, p in m.named_parameters():
    print("   %-32s %.2e  |ref| %.2e" % (k, rel_err(p.grad, o.sd["classifier." + k].grad), float(o.sd["classifier." + k].grad.abs().max())))
