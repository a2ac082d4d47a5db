import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tests.util import rel_err
from tests.test_hip_modules import _build, record_masks
from oracle.deeplab import OracleDeepLab
from oracle.synth import synth_images
from oracle.make_golden import upstream
import torch.nn.functional as F
dev = torch.device("cuda:0")
nchw = lambda t: t.detach().cpu().permute(0, 3, 1, 2)
for bb, os_, hw in (("resnet50", 16, 65), ("resnet101", 8, 65)):
    m, cfg, sd = _build(bb, os_)
    m._debug_keep_dfeats = True
    x = synth_images(2, hw, hw, seed=71)
    m.train()
    with record_masks(m, "") as rec:
        lg = m(x.to(dev))
    up = upstream(lg.shape, 12)
    lg.backward(up.to(dev))
    o = OracleDeepLab(cfg, sd, dropout_p=0.0).train()
    o.relu_masks, o.preact = rec.masks(), {}
    feats = o.backbone(x)
    for v in feats.values(): v.retain_grad()
    y = o.head(feats)
    lgo = F.interpolate(y, size=x.shape[-2:], mode="bilinear", align_corners=False)
    lgo.backward(up)
    print(bb, os_, "d_out err %.2e  d_low err %.2e" % (rel_err(nchw(m._debug_dfeats["out"]), feats["out"].grad), rel_err(nchw(m._debug_dfeats["low_level"]), feats["low_level"].grad)),
          "|d_out| %.2e" % float(feats["out"].grad.abs().max()))
    # is the d_out error a per-channel constant (pooling branch)?
    d = nchw(m._debug_dfeats["out"]) - feats["out"].grad
    print("   err: total max %.2e ; after removing per-(n,c) mean %.2e" % (float(d.abs().max()), float((d - d.mean((2, 3), keepdim=True)).abs().max())))
