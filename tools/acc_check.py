"""whole-model train-mode logits of this path vs (a) the reference's golden vector (torch CPU fp32) and (b) the oracle
evaluated in float64 -- separates this path's own rounding error from the reference's (r101 / os8 on 65x65, batch 2: the
ill-conditioned case of tests/test_hip_modules.py::test_whole_model)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import load, rel_err
from iswm_amd.network import modeling
from oracle.deeplab import OracleDeepLab
from oracle.synth import ArchCfg, synth_state_dict, synth_images

for tag, backbone, os_ in (("r50_os16", "resnet50", 16), ("r101_os8", "resnet101", 8)):
    fx = load("model_%s.npz" % tag)
    cfg = ArchCfg("deeplabv3plus", backbone, 2, os_)
    sd = synth_state_dict(cfg)
    x = synth_images(2, 65, 65, seed=71)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    o64 = OracleDeepLab(cfg, sd64, dropout_p=0.0).train()
    with torch.no_grad():
        ref64 = o64(x.double())
    m = modeling._segm_resnet("deeplabv3plus", backbone, 2, os_, False)
    m.load_state_dict(sd, strict=True)
    m.classifier.aspp.project[3].p = 0.0
    m = m.cuda().train()
    with torch.no_grad():
        lg = m(x.cuda())
    print("%s planes=%s: vs golden(fp32 cpu) %.3e   vs oracle fp64 %.3e   golden vs fp64 %.3e" %
          (tag, os.environ.get("ISWM_PLANES", "1"), rel_err(lg, fx["train_logits"]), rel_err(lg, ref64), rel_err(fx["train_logits"], ref64)))
