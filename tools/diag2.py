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
import torch.nn.functional as F

dev = torch.device("cuda:0")
sd = synth_from_shapes(aspp_shapes("aspp", 64))
m = ASPP(64, [6, 12, 18])
m.load_state_dict(OrderedDict((k, sd["aspp." + k]) for k in m.state_dict()), strict=True)
m = m.to(dev).train()
m.project[3].p = 0.0
x = synth_images(2, 17, 17, seed=11, c=64).to(dev)
xh = ops.nchw_to_nhwc(x)
y = m.fwd(xh, True)
# grab saved state of branch 1
st, ctxs = m.convs[1]._saved
c = ctxs[0]
out_f = c["out"].clone()
yraw = c["y"].clone()
up = ops.nchw_to_nhwc(upstream((2, 256, 17, 17), 5).to(dev))
# run project bwd manually
sink = _hip.GradSink()
dcat = m.project.bwd(up, sink)
print("out slice unchanged:", torch.equal(out_f, c["out"]))
for i in range(4):
    conv = m.convs[i]
    st, ctxs = conv._saved
    c = ctxs[0]
    dout = dcat[..., i * 256:(i + 1) * 256]
    dz_ref = (dout * (c["out"] > 0)).double()
    dbeta_ref = dz_ref.sum((0, 1, 2))
    xhat = (c["y"].double() - c["coef"][2].double()) * c["coef"][3].double()
    dgamma_ref = (dz_ref * xhat).sum((0, 1, 2))
    dg = torch.empty(256, device=dev); db = torch.empty(256, device=dev)
    dy, _ = ops.bn_backward(dout, c["out"], c["y"], c["coef"], conv[1].weight, True, True, dg, db)
    torch.cuda.synchronize()
    print(i, "dbeta err %.2e dgamma err %.2e" % (rel_err(db, dbeta_ref), rel_err(dg, dgamma_ref)),
          "strides", dout.stride(), c["out"].stride(), c["y"].stride(), "ptr%16", dout.data_ptr() % 16, c["out"].data_ptr() % 16)
    # contiguous copy variant
    dg2 = torch.empty(256, device=dev); db2 = torch.empty(256, device=dev)
    ops.bn_backward(dout.contiguous(), c["out"].contiguous(), c["y"], c["coef"], conv[1].weight, True, True, dg2, db2)
    print("   contiguous: dbeta err %.2e" % rel_err(db2, dbeta_ref))
    bad = (db - dbeta_ref.float()).abs()
    print("   worst channels", torch.topk(bad, 5).indices.tolist(), torch.topk(bad, 5).values.tolist())
