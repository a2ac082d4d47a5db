"""Run the decoder 3x3 conv forward (bf16x6) a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import ops
dev = torch.device("cuda:0")
xh = torch.randn(16, 129, 129, 256, device=dev); wo = torch.randn(256, 3, 3, 256, device=dev) * 0.05
g = ops.ConvGeom(xh, 256, 3, 3, 1, 1, 1)
for _ in range(6):
    ops.conv2d_fwd(xh, wo, g, want_stats=True)
torch.cuda.synchronize()
