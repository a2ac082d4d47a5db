import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from collections import OrderedDict
from oracle.deeplab import OracleDeepLab
from oracle.synth import ArchCfg, aspp_shapes, synth_from_shapes, synth_images
from oracle.make_golden import upstream
from tests.util import rel_err
from iswm_amd.network._deeplab import ASPP
dev = torch.device("cuda:0")
sd = synth_from_shapes(aspp_shapes("aspp", 64))
hw = 17
o = OracleDeepLab(ArchCfg(), sd, dropout_p=0.0).train()
x = synth_images(2, hw, hw, seed=11, c=64)
xo = x.clone().requires_grad_(True)
yo = o.aspp(xo, "aspp")
up = upstream(yo.shape, 5)
(yo * up).sum().backward()
if len(sys.argv) > 1 and sys.argv[1] == "poison":
    junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(64)]
    del junk
for rep in range(3):
    m = ASPP(64, [6, 12, 18])
    m.load_state_dict(OrderedDict((k, sd["aspp." + k]) for k in m.state_dict()), strict=True)
    m = m.to(dev).train()
    m.project[3].p = 0.0
    xg = x.to(dev).requires_grad_(True)
    y = m(xg)
    (y * up.to(dev)).sum().backward()
    errs = {k: rel_err(p.grad, o.sd["aspp." + k].grad) for k, p in m.named_parameters()}
    bad = {k: "%.1e" % v for k, v in errs.items() if not v < 1e-4}
    print("rep", rep, "grad_x %.2e" % rel_err(xg.grad, xo.grad), "bad:", bad)
