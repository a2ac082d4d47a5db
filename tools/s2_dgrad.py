"""stride-2 data-gradient micro-benchmark (the four strided convs of ResNet-101 at 513x513, batch 16)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import ops
dev = torch.device("cuda:0")
for (cin, cout, k, h, p) in ((256, 512, 1, 129, 0), (512, 1024, 1, 65, 0), (128, 128, 3, 129, 1), (256, 256, 3, 65, 1)):
    x = torch.randn(16, h, h, cin, device=dev)
    g = ops.ConvGeom(x, cout, k, k, 2, p, 1)
    dy = torch.randn(16, g.ho, g.wo, cout, device=dev)
    w = torch.randn(cout, k, k, cin, device=dev) * 0.05
    for acc in (False, True):
        dx = torch.zeros_like(x)
        for _ in range(3):
            ops.conv2d_dgrad(dy, w, g, tuple(x.shape), dx=dx, accumulate=acc)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv2d_dgrad(dy, w, g, tuple(x.shape), dx=dx, accumulate=acc)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) * 100
        print("c%d->%d k%d %dx%d acc=%d: %.1f us (incl. weight pack)  %.1f TF" % (cin, cout, k, h, h, acc, us, g.flops() / us / 1e6))
