"""__graft_entry__.smoke() with the per-parameter gradient errors listed (worst first)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import _lib
from iswm_amd.network import _hip, modeling
from iswm_amd.optim import FusedSGD
from iswm_amd.utils.loss import CrossEntropyLoss
from oracle import loss as oloss
from oracle.deeplab import OracleDeepLab
from oracle.synth import ArchCfg, synth_images, synth_labels, synth_state_dict
_lib.load()
dev = torch.device("cuda:0")
cfg = ArchCfg("deeplabv3plus", "resnet50", 2, 16)
sd = synth_state_dict(cfg)
m = modeling.deeplabv3plus_resnet50(num_classes=2, output_stride=16)
m.load_state_dict(sd, strict=True)
m.classifier.aspp.project[3].p = 0.0
m = m.to(dev).train()
x = synth_images(4, 65, 65, seed=3)
lab = synth_labels(4, 65, 65, seed=3, p_fg=0.2, p_ignore=0.05)
w = torch.tensor([1.0, 3.0])
opt = FusedSGD(m.parameters(), momentum=0.9, weight_decay=1e-4, nesterov=True)
rec, pool = {}, {}
_hip.MASK_RECORDER, _hip.POOL_RECORDER = rec, pool
logits = m(x.to(dev))
_hip.MASK_RECORDER = _hip.POOL_RECORDER = None
loss = CrossEntropyLoss(weight=w)(logits, lab.to(dev))
opt.zero_grad()
loss.backward()
grads = {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()}
names = {mod: n for n, mod in m.named_modules()}
o = OracleDeepLab(cfg, sd, dropout_p=0.0).train()
o.relu_masks = {names[bn]: v.permute(0, 3, 1, 2).cpu() for bn, v in rec.items()}
if os.environ.get("SMOKE_OWN_POOL", "0") != "1":          # 1: let the oracle's max-pool pick its own maxima (shows the near-tie effect)
    o.pool_index = next(iter(pool.values())).permute(0, 3, 1, 2).cpu()
lgo = o(x)
lo = oloss.weighted_ce(lgo, lab, w)
lo.backward()
errs = sorted(((float((grads[k] - v.grad).abs().max() / v.grad.abs().max()), k) for k, v in o.named_parameters()), reverse=True)
for e, k in errs[:8]:
    print("%.3e  %s" % (e, k))
