import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tests.util import rel_err
from tests.test_hip_modules import _build, record_masks, check_sign_patterns
from iswm_amd.utils.loss import CrossEntropyLoss
from oracle import loss as oloss
from oracle.deeplab import OracleDeepLab
from oracle.synth import synth_images, synth_labels
dev = torch.device("cuda:0")
for hw in (65, 129):
    m, cfg, sd = _build("resnet50", 16)
    x = synth_images(2, hw, hw, seed=71)
    labels = synth_labels(2, hw, hw, seed=71, p_fg=0.2, p_ignore=0.05)
    m.train()
    with record_masks(m, "") as rec:
        lg = m(x.to(dev))
    w = torch.tensor([1.0, 3.0])
    loss = CrossEntropyLoss(weight=w, ignore_index=255)(lg, labels.to(dev))
    loss.backward()
    o = OracleDeepLab(cfg, sd, dropout_p=0.0).train()
    o.relu_masks, o.preact = rec.masks(), {}
    lgo = o(x)
    lo = oloss.weighted_ce(lgo, labels, w, 255)
    lo.backward()
    bad = tot = 0
    worstz = 0
    for site, mk in o.relu_masks.items():
        z = o.preact[site]; mism = (z > 0) != mk
        bad += int(mism.sum()); tot += mk.numel()
        if mism.any(): worstz = max(worstz, float(z[mism].abs().max() / z.abs().max()))
    print("hw", hw, "logits err %.2e loss err %.2e flips %d/%d worst |z|/scale %.1e" % (rel_err(lg, lgo.detach()), rel_err(loss, lo.detach()), bad, tot, worstz))
    params = dict(m.named_parameters())
    errs = sorted(((rel_err(params[k].grad, v.grad), k) for k, v in o.named_parameters()), reverse=True)
    for e, k in errs[:8]:
        print("   %.2e %s" % (e, k))
    order = [k for k, _ in o.named_parameters()]
    convs = [k for k in order if k.endswith("conv2.weight")]
    print("   conv2 trend:", " ".join("%.0e" % rel_err(params[k].grad, o.sd[k].grad) for k in convs))
