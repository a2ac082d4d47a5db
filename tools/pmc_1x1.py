"""Run the layer3 1x1 conv forward (1024 -> 256 at 33x33, batch 16, bf16x6 packed) a few times for rocprofv3 --pmc."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import ops
dev = torch.device("cuda:0")
xh = torch.randn(16, 33, 33, 1024, device=dev); wo = torch.randn(256, 1, 1, 1024, device=dev) * 0.05
g = ops.ConvGeom(xh, 256, 1, 1, 1, 0, 1)
for _ in range(6):
    ops.conv2d_fwd(xh, wo, g, want_stats=True)
torch.cuda.synchronize()
