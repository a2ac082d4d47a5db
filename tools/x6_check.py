"""GPU: accuracy and speed of the bf16x6 conv path vs the exact-fp32 MFMA path (and vs fp64 on the CPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from iswm_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
def rnd(*s, seed=0):
    g = torch.Generator().manual_seed(seed); return torch.randn(*s, generator=g)
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().to(dev)
def nchw(t): return t.detach().cpu().permute(0, 3, 1, 2)
kinds = sys.argv[1].split(",") if len(sys.argv) > 1 else ["fwd"]
small = [(64, 256, 3, 1, 6, 6, 33, 2), (256, 64, 1, 1, 0, 1, 33, 2), (64, 64, 3, 2, 1, 1, 33, 2), (2048, 256, 3, 1, 18, 18, 17, 2), (128, 512, 1, 1, 0, 1, 40, 3)]
print("== accuracy (max abs err / max |ref|, ref = fp64 CPU) ==")
for cin, cout, k, s, p, d, h, n in small:
    x = rnd(n, cin, h, h, seed=1); w = rnd(cout, cin, k, k, seed=2) * (2.0 / (cin * k * k)) ** 0.5
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, s, p, d)
    dy = rnd(*ref.shape, seed=3)
    ref.backward(dy.double())
    xh, wo, dyh = nhwc(x), w.permute(0, 2, 3, 1).contiguous().to(dev), nhwc(dy)
    g = ops.ConvGeom(xh, cout, k, k, s, p, d)
    errs = {}
    for mode in (0, 1):
        lib.iswm_set_conv_math(mode)
        y, _, _ = ops.conv2d_fwd(xh, wo, g)
        e = [float((nchw(y).double() - ref.detach()).abs().max() / ref.detach().abs().max())]
        if "dgrad" in kinds:
            dx = ops.conv2d_dgrad(dyh, wo, g, tuple(xh.shape))
            e.append(float((nchw(dx).double() - xr.grad).abs().max() / xr.grad.abs().max()))
        if "wgrad" in kinds:
            dw = ops.conv2d_wgrad(xh, dyh, g)
            e.append(float((dw.cpu().permute(0, 3, 1, 2).double() - wr.grad).abs().max() / wr.grad.abs().max()))
        errs[mode] = e
    print("c%d->%d k%d s%d d%d h%d: f32 %s   bf16x6 %s" % (cin, cout, k, s, d, h, ["%.1e" % v for v in errs[0]], ["%.1e" % v for v in errs[1]]))
print("== speed (n16) ==")
big = [(256, 256, 3, 1, 1, 1, 129), (256, 256, 3, 1, 1, 1, 33), (1024, 256, 1, 1, 0, 1, 33), (256, 1024, 1, 1, 0, 1, 33), (2048, 256, 3, 1, 12, 12, 33), (512, 512, 3, 1, 2, 2, 33), (64, 256, 1, 1, 0, 1, 129)]
for cin, cout, k, s, p, d, h in big:
    xh = torch.randn(16, h, h, cin, device=dev); wo = torch.randn(cout, k, k, cin, device=dev) * 0.05
    g = ops.ConvGeom(xh, cout, k, k, s, p, d)
    dyh = torch.randn(16, g.ho, g.wo, cout, device=dev)
    line = "c%d->%d k%d d%d %dx%d:" % (cin, cout, k, d, h, h)
    for kind in kinds:
        for mode in (0, 1):
            lib.iswm_set_conv_math(mode)
            def run():
                if kind == "fwd": ops.conv2d_fwd(xh, wo, g, want_stats=True)
                elif kind == "dgrad": ops.conv2d_dgrad(dyh, wo, g, tuple(xh.shape))
                else: ops.conv2d_wgrad(xh, dyh, g)
            for _ in range(3): run()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): run()
            b.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b) * 100
            line += "  %s[%s] %7.1f us %6.1f TF" % (kind, "x6" if mode else "f32", us, g.flops() / us / 1e6)
    print(line)
lib.iswm_set_conv_math(0)
