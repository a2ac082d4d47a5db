import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tests.util import rel_err
from tests.test_hip_modules import _build, record_masks
from oracle.deeplab import OracleDeepLab
from oracle.synth import synth_images
from oracle.make_golden import upstream
dev = torch.device("cuda:0")
for bb, os_, hw in (("resnet50", 16, 65), ("resnet101", 8, 65), ("resnet50", 16, 129)):
    m, cfg, sd = _build(bb, os_)
    x = synth_images(2, hw, hw, seed=71)
    m.train()
    with record_masks(m, "") as rec:
        lg = m(x.to(dev))
    up = upstream(lg.shape, 12)
    lg.backward(up.to(dev))
    o = OracleDeepLab(cfg, sd, dropout_p=0.0).train()
    o.relu_masks, o.preact = rec.masks(), {}
    lgo = o(x)
    lgo.backward(up)
    params = dict(m.named_parameters())
    errs = sorted(((rel_err(params[k].grad, v.grad), k) for k, v in o.named_parameters()), reverse=True)
    print(bb, os_, hw, "logits err %.2e" % rel_err(lg, lgo.detach()))
    for e, k in errs[:6]:
        print("   %.2e %s  |ref| %.2e" % (e, k, float(o.sd[k].grad.abs().max())))
    order = [k for k, _ in o.named_parameters()]
    print("   bn2.bias trend:", " ".join("%.0e" % rel_err(params[k].grad, o.sd[k].grad) for k in order if k.endswith("bn2.bias")))
    print("   head:", " ".join("%s=%.0e" % (k.split("classifier.")[-1], rel_err(params[k].grad, o.sd[k].grad)) for k in order if k.startswith("classifier") and k.endswith(".bias")))
