"""kernel-only time / bandwidth of the memory-bound BatchNorm passes on the layer3 block-output shape (16 x 33 x 33 x 1024)
and the layer1 shape (16 x 129 x 129 x 256), fp32 and planes forms"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import ops
dev = torch.device("cuda:0")

def t_us(fn, n=20):
    for _ in range(3): fn()
    a = [torch.cuda.Event(enable_timing=True) for _ in range(n)]; b = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for i in range(n):
        a[i].record(); fn(); b[i].record()
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) * 1e3 for x, y in zip(a, b))
    return ts[len(ts) // 2]

for (n, h, w, c) in ((16, 33, 33, 1024), (16, 33, 33, 256), (16, 129, 129, 256)):
    y = torch.randn(n, h, w, c, device=dev); r = torch.randn(n, h, w, c, device=dev); rp = ops.split_planes(r)
    coef = torch.rand(4, c, device=dev) + 0.5
    dout = torch.randn(n, h, w, c, device=dev); gamma = torch.rand(c, device=dev) + 0.5
    dg, db = torch.empty(c, device=dev), torch.empty(c, device=dev)
    e = n * h * w * c
    op = ops.new_planes(n, h, w, c, dev); of = torch.empty_like(y); dyp = ops.new_planes(n, h, w, c, dev); dyf = torch.empty_like(y)
    rows = [
        ("bn_apply relu        f32->f32 ", lambda: ops.bn_apply(y, coef, True, None, of), 8),
        ("bn_apply relu        f32->pl  ", lambda: ops.bn_apply(y, coef, True, None, op), 10),
        ("bn_apply relu+res    f32->f32 ", lambda: ops.bn_apply(y, coef, True, r, of), 12),
        ("bn_apply relu+res    pl ->pl  ", lambda: ops.bn_apply(y, coef, True, rp, op), 16),
        ("bn_backward masky    ->f32    ", lambda: ops.bn_backward(dout, of, y, coef, gamma, True, True, dg, db, dy=dyf), 8 + 12),
        ("bn_backward masky    ->pl     ", lambda: ops.bn_backward(dout, op, y, coef, gamma, True, True, dg, db, dy=dyp), 8 + 14),
        ("bn_backward res      f32->f32 ", lambda: ops.bn_backward(dout, of, y, coef, gamma, True, True, dg, db, want_dres=True, dy=dyf), 12 + 20),
        ("bn_backward res      pl ->pl  ", lambda: ops.bn_backward(dout, op, y, coef, gamma, True, True, dg, db, want_dres=True, dy=dyp), 10 + 20),
    ]
    for name, fn, bpe in rows:
        us = t_us(fn)
        print("%dx%dx%dx%-5d %s %7.1f us  %5.2f TB/s (%d B/elem)" % (n, h, w, c, name, us, e * bpe / us * 1e-6, bpe), flush=True)
