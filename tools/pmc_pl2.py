"""Run ONE geometry of the planes forward kernel a few times (for rocprofv3 --pmc passes).
usage: pmc_pl2.py [cin cout k pad dil]   (default: layer3 1x1 1024->256 at 33x33, batch 16)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import ops
dev = torch.device("cuda:0")
a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [1024, 256, 1, 0, 1]
cin, cout, k, pad, dil = a
xh = ops.split_planes(torch.randn(16, 33, 33, cin, device=dev))
w = torch.randn(cout, k, k, cin, device=dev) * 0.05
g = ops.ConvGeom(xh, cout, k, k, 1, pad, dil)
for _ in range(10):
    ops.conv2d_fwd(xh, w, g, want_stats=True)
torch.cuda.synchronize()
print("flops per launch %.3f GF" % (g.flops() / 1e9))
