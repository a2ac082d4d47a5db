"""Isolated kernel time of the planes forward / data-gradient kernels on the production shapes (median of N launches, HIP
events on the launch stream).  One process = one setting of the env switches (ISWM_PL2_WIDE, ISWM_PL2W_PRIO, ...): run it
once per setting in the same gpurun call for a same-box A/B.   usage: pl2_shapes.py [rounds] [case filter substring]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from iswm_amd import ops

dev = torch.device("cuda:0")
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
FILT = sys.argv[2] if len(sys.argv) > 2 else ""
CASES = [  # n, h, w, cin, cout, k, stride, pad, dil
    (16, 33, 33, 1024, 256, 1, 1, 0, 1), (16, 33, 33, 256, 1024, 1, 1, 0, 1), (16, 33, 33, 256, 256, 3, 1, 1, 1),
    (16, 33, 33, 2048, 512, 1, 1, 0, 1), (16, 33, 33, 512, 2048, 1, 1, 0, 1), (16, 33, 33, 512, 512, 3, 1, 2, 2),
    (16, 33, 33, 1024, 2048, 1, 1, 0, 1), (16, 33, 33, 1024, 512, 1, 1, 0, 1),
    (16, 129, 129, 320, 256, 3, 1, 1, 1), (16, 129, 129, 256, 256, 3, 1, 1, 1),
    (16, 33, 33, 2048, 256, 3, 1, 6, 6), (16, 33, 33, 2048, 256, 3, 1, 12, 12), (16, 33, 33, 2048, 256, 3, 1, 18, 18),
    (16, 129, 129, 64, 256, 1, 1, 0, 1), (16, 129, 129, 256, 64, 1, 1, 0, 1), (16, 129, 129, 64, 64, 3, 1, 1, 1),
    (16, 65, 65, 128, 512, 1, 1, 0, 1), (16, 65, 65, 512, 128, 1, 1, 0, 1), (16, 65, 65, 128, 128, 3, 1, 1, 1),
]


def med(fn):
    a = [torch.cuda.Event(enable_timing=True) for _ in range(ROUNDS)]
    b = [torch.cuda.Event(enable_timing=True) for _ in range(ROUNDS)]
    for i in range(ROUNDS):
        a[i].record()
        fn()
        b[i].record()
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) * 1e3 for x, y in zip(a, b))
    return ts[len(ts) // 2]


print("env:", {k: v for k, v in os.environ.items() if k.startswith("ISWM_")})
tot = totw = 0.0
for c in CASES:
    n, h, w, cin, cout, k, s, p, d = c
    tag = "n%d %dx%d c%d->%d k%d s%d d%d" % (n, h, w, cin, cout, k, s, d)
    if FILT and FILT not in tag:
        continue
    x = ops.split_planes(torch.randn(n, h, w, cin, device=dev))
    wt = torch.randn(cout, k, k, cin, device=dev) * 0.05
    g = ops.ConvGeom(x, cout, k, k, s, p, d)
    dy = ops.split_planes(torch.randn(n, g.ho, g.wo, cout, device=dev))
    dx = torch.zeros(n, h, w, cin, device=dev)
    from iswm_amd.ops import _kernel_name
    import ctypes
    for _ in range(3):
        ops.conv2d_fwd(x, wt, g, want_stats=True)
        ops.conv2d_dgrad(dy, wt, g, (n, h, w, cin), dx=dx, accumulate=False)
    # pre-pack so that only the conv kernel is timed
    from iswm_amd.ops import _pl2_bytes, call, _p, _stream, pgeom, geom
    ldp, ps = pgeom(x)[4], pgeom(x)[5]
    y = torch.empty(n, g.ho, g.wo, cout, device=dev)
    d0 = g.desc(ldp, cout)
    wpk = torch.empty((_pl2_bytes(d0, 0) // 4,), device=dev)
    call("iswm_conv2d_pl2_pack_weights", ctypes.byref(d0), 0, _p(wt), _p(wpk), _stream())
    tf = med(lambda: ops.conv2d_fwd(x, wt, g, out=y, want_stats=True, wpk2=wpk))
    d1 = g.desc(cin, pgeom(dy)[4])
    nb = _pl2_bytes(d1, 1)
    td = ta = tb = float("nan")
    if nb:
        wpk1 = torch.empty((nb // 4,), device=dev)
        call("iswm_conv2d_pl2_pack_weights", ctypes.byref(d1), 1, _p(wt), _p(wpk1), _stream())
        td = med(lambda: ops.conv2d_dgrad(dy, wt, g, (n, h, w, cin), dx=dx, wpk2=wpk1))
        ta = med(lambda: ops.conv2d_dgrad(dy, wt, g, (n, h, w, cin), dx=dx, accumulate=True, wpk2=wpk1))
        # the residual-stage form (relu code 3): accumulate + mask by the producer's saved output + its two BatchNorm sums
        yprod = torch.randn(n, h, w, cin, device=dev)
        coef = torch.stack([torch.ones(cin, device=dev), torch.zeros(cin, device=dev), torch.zeros(cin, device=dev), torch.ones(cin, device=dev)])
        try:
            tb = med(lambda: ops.conv2d_dgrad(dy, wt, g, (n, h, w, cin), dx=dx, accumulate=True, wpk2=wpk1,
                                              bn_stats=ops.BnStats(yprod, coef, True, mask=x)))
        except Exception as e:
            print("   bn3 form:", str(e)[:100])
    fl = g.flops()
    dw = torch.empty(cout, k, k, cin, device=dev)
    for _ in range(2):
        ops.conv2d_wgrad(x, dy, g, dw)
    tw = med(lambda: ops.conv2d_wgrad(x, dy, g, dw))
    d7 = ops.ConvDesc(n, h, w, cin, g.ho, g.wo, cout, k, k, s, p, d, pgeom(x)[4], pgeom(dy)[4])
    print("%-34s fwd %7.1f us %6.1f TF %-28s| dgrad %7.1f us %6.1f TF  acc %7.1f us  bn3 %7.1f us  %s | wgrad %7.1f us %6.1f TF %s" %
          (tag, tf, fl / tf / 1e6, _kernel_name(d0, 5), td, fl / td / 1e6, ta, tb, _kernel_name(d1, 6) if nb else "-",
           tw, fl / tw / 1e6, _kernel_name(d7, 7)))
    tot += tf + (td if nb else 0)
    totw += tw
print("sum fwd + dgrad: %.1f us   sum wgrad: %.1f us" % (tot, totw))
