"""kernel-only time (HIP events around the launch, ops.KernelProfile) of the packed bf16x6 forward kernel on three
layer3/4 geometries -- used with scratch ablation builds of conv_mfma_x6.hip"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import ops
dev = torch.device("cuda:0")
for (cin, cout, k, p, d) in ((1024, 256, 1, 0, 1), (256, 1024, 1, 0, 1), (512, 512, 3, 2, 2)):
    xh = torch.randn(16, 33, 33, cin, device=dev); wo = torch.randn(cout, k, k, cin, device=dev) * 0.05
    g = ops.ConvGeom(xh, cout, k, k, 1, p, d)
    for _ in range(3):
        ops.conv2d_fwd(xh, wo, g, want_stats=True)
    torch.cuda.synchronize()
    ops.KPROF = ops.KernelProfile()
    for _ in range(10):
        ops.conv2d_fwd(xh, wo, g, want_stats=True)
        torch.cuda.synchronize()
    s = ops.KPROF.summary(); ops.KPROF = None
    for name, v in s.items():
        print("c%d->%d k%d d%d: %s  %.1f us" % (cin, cout, k, d, name, v["ms"] * 1e3 / v["launches"]))
