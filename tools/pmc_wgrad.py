"""Run ONE geometry of the dominant kernel a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import ops
dev = torch.device("cuda:0")
# layer3 3x3 256->256 @33x33, batch 16: the geometry that dominates k_conv_wgrad<128,128,1,true> (22 of its 33 launches/step)
xh = torch.randn(16, 33, 33, 256, device=dev); dyh = torch.randn(16, 33, 33, 256, device=dev)
g = ops.ConvGeom(xh, 256, 3, 3, 1, 1, 1)
for _ in range(10):
    ops.conv2d_wgrad(xh, dyh, g)
torch.cuda.synchronize()
print("flops per launch %.3f GF" % (g.flops() / 1e9))
