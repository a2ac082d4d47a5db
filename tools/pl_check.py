"""Pre-split ("planes") conv kernels vs the fp32-input bf16x6 kernels: outputs and kernel-only time.
(conv_mfma_pl2.hip / conv_wgrad_pl.hip; relative error bound -- the summation order differs).

    python tools/pl_check.py [rounds]

For each geometry: split the activation once (iswm_split_planes), run iswm_conv2d_{fwd,dgrad}_pl2 and the packed
fp32-input kernel on the same operands, bound the difference, then time both interleaved in one process."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from iswm_amd import _lib, ops
from iswm_amd.ops import _p, _stream, call

dev = torch.device("cuda:0")
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def split(x):
    n, h, w, c = x.shape
    m = n * h * w
    planes = torch.empty((3, m, c), dtype=torch.bfloat16, device=dev)
    call("iswm_split_planes", _p(x), m, c, c, _p(planes), c, m * c, _stream())
    return planes


def time_us(fn, rounds):
    a = [torch.cuda.Event(enable_timing=True) for _ in range(rounds)]
    b = [torch.cuda.Event(enable_timing=True) for _ in range(rounds)]
    for i in range(rounds):
        a[i].record()
        fn()
        b[i].record()
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) * 1e3 for x, y in zip(a, b))
    return ts[len(ts) // 2], ts[0]


if os.environ.get('PL_CASES'):
    NC_ = int(os.environ['PL_CASES'])
else:
    NC_ = None
CASES = [  # n, h, w, cin, cout, k, stride, pad, dil
    (16, 33, 33, 1024, 256, 1, 1, 0, 1),
    (16, 33, 33, 256, 1024, 1, 1, 0, 1),
    (16, 33, 33, 2048, 512, 1, 1, 0, 1),
    (16, 33, 33, 512, 2048, 1, 1, 0, 1),
    (16, 33, 33, 2048, 256, 1, 1, 0, 1),
    (16, 33, 33, 1280, 256, 1, 1, 0, 1),
    (16, 65, 65, 512, 128, 1, 1, 0, 1),
    (16, 65, 65, 128, 512, 1, 1, 0, 1),
    (16, 129, 129, 256, 64, 1, 1, 0, 1),
    (16, 129, 129, 64, 256, 1, 1, 0, 1),
    (16, 129, 129, 256, 48, 1, 1, 0, 1),
    (16, 65, 65, 512, 1024, 1, 2, 0, 1),
    (16, 65, 65, 256, 256, 3, 2, 1, 1),
    (16, 33, 33, 256, 256, 3, 1, 1, 1),
    (16, 33, 33, 512, 512, 3, 1, 2, 2),
    (16, 129, 129, 64, 64, 3, 1, 1, 1),
    (16, 129, 129, 320, 256, 3, 1, 1, 1),
    (16, 33, 33, 2048, 256, 3, 1, 6, 6),
    (16, 33, 33, 2048, 256, 3, 1, 12, 12),
    (16, 33, 33, 2048, 256, 3, 1, 18, 18),
    (2, 17, 19, 64, 96, 3, 1, 2, 2),
    (5, 23, 19, 128, 200, 3, 2, 1, 1),
    (3, 9, 11, 64, 40, 1, 1, 0, 1),
]

os.environ["ISWM_X6_PATCH"] = "0"
lib = _lib.load()
print("conv math", lib.iswm_get_conv_math())
if os.environ.get('PL_ONLY'):          # comma-separated case indices
    CASES = [CASES[int(i)] for i in os.environ['PL_ONLY'].split(',')]
NOCHECK = bool(os.environ.get("ISWM_WG_ABL") or os.environ.get("ISWM_PL_ABL") or os.environ.get("ISWM_PL2_ABL"))     # ablations are wrong by design
for (n, h, w, cin, cout, k, s, p, d) in (CASES[:NC_] if NC_ else CASES):
    torch.manual_seed(0)
    x = torch.randn(n, h, w, cin, device=dev)
    wo = torch.randn(cout, k, k, cin, device=dev) * 0.05
    g = ops.ConvGeom(x, cout, k, k, s, p, d)
    desc = g.desc(cin, cout)
    tag = g.tag()
    if os.environ.get("PL_WGRAD", "1") != "0" and cin % 8 == 0 and cout % 8 == 0:
        dy = torch.randn(n, g.ho, g.wo, cout, device=dev)
        xp, dyp = split(x), split(dy)
        dw_ref = torch.zeros(cout, k, k, cin, device=dev)
        dw_new = torch.zeros_like(dw_ref)
        need = lib.iswm_conv2d_wgrad_workspace(ctypes.byref(desc))
        ws = torch.empty(max(need // 4, 4), device=dev)
        need2 = lib.iswm_conv2d_wgrad_planes_workspace(ctypes.byref(desc))
        ws2 = torch.empty(max(need2 // 4, 4), device=dev)
        f_ref = lambda: call("iswm_conv2d_wgrad", ctypes.byref(desc), _p(x), _p(dy), _p(dw_ref), _p(ws), need, _stream())
        f_new = lambda: call("iswm_conv2d_wgrad_planes", ctypes.byref(desc), _p(xp), xp.shape[1] * cin, _p(dyp), dyp.shape[1] * cout,
                             _p(dw_new), _p(ws2), need2, _stream())
        f_ref(); f_new()
        torch.cuda.synchronize()
        err = float((dw_ref - dw_new).abs().max() / dw_ref.abs().max())
        for _ in range(3):
            f_ref(); f_new()
        r_med, _ = time_us(f_ref, ROUNDS)
        n_med, _ = time_us(f_new, ROUNDS)
        fl = g.flops()
        print("%-34s wgrad err=%.1e  fp32-in %.1f us (%.0f TF)  planes %.1f us (%.0f TF)  x%.2f" %
              (tag, err, r_med, fl / r_med * 1e-6, n_med, fl / n_med * 1e-6, r_med / n_med), flush=True)
        assert NOCHECK or err < 5e-6, "planes weight gradient differs"
    if os.environ.get("PL_WGRAD_ONLY"):
        continue
    for kind in (0, 1):
        nb = lib.iswm_conv2d_packed_weight_bytes(ctypes.byref(desc), kind)
        if nb == 0:
            continue
        wpk = torch.empty((nb // 4,), dtype=torch.float32, device=dev)
        call("iswm_conv2d_pack_weights", ctypes.byref(desc), kind, _p(wo), _p(wpk), _stream())
        if kind == 0:
            xp = split(x)
            y_ref = torch.empty(n, g.ho, g.wo, cout, device=dev)
            y_new = torch.empty_like(y_ref)
            nt, tr = ctypes.c_int(0), ctypes.c_int(0)
            os.environ["ISWM_X6_PATCH"] = "0"
            tiles = lib.iswm_conv2d_stat_tiles(ctypes.byref(desc))
            st_ref = torch.zeros(2, tiles, cout, device=dev)
            st_new = torch.zeros(2, tiles, cout, device=dev)
            f_ref = lambda: call("iswm_conv2d_fwd_packed", ctypes.byref(desc), _p(x), _p(wpk), None, _p(y_ref), _p(st_ref), _stream())
            nb2 = lib.iswm_conv2d_pl2_weight_bytes(ctypes.byref(desc), 0)
            if nb2 == 0:
                continue
            wpk2 = torch.empty((nb2 // 4,), dtype=torch.float32, device=dev)
            call("iswm_conv2d_pl2_pack_weights", ctypes.byref(desc), 0, _p(wo), _p(wpk2), _stream())
            tr2 = lib.iswm_conv2d_pl2_tile_rows(ctypes.byref(desc), 0)
            st_new = torch.zeros(2, (n * g.ho * g.wo + tr2 - 1) // tr2, cout, device=dev)
            f_new = lambda: call("iswm_conv2d_fwd_pl2", ctypes.byref(desc), _p(xp), xp.shape[1] * cin, _p(wpk2), None,
                                 _p(y_new), _p(st_new), _stream())
            f_ref(); f_new()
            torch.cuda.synchronize()
            err = float((y_ref - y_new).abs().max() / y_ref.abs().max())
            serr = float((st_ref[0].sum(0) - st_new[0].sum(0)).abs().max() / st_ref[0].sum(0).abs().max())
            same = err < 5e-6 and serr < 1e-4
            if not same:
                print("   fwd err", err, "stat-sum err", serr)
            name = "fwd  "
        else:
            dy = torch.randn(n, g.ho, g.wo, cout, device=dev)
            dyp = split(dy)
            dx_ref = torch.zeros(n, h, w, cin, device=dev)
            dx_new = torch.zeros_like(dx_ref)
            f_ref = lambda: call("iswm_conv2d_dgrad_packed", ctypes.byref(desc), _p(dy), _p(wpk), _p(dx_ref), 0, _stream())
            nb2 = lib.iswm_conv2d_pl2_weight_bytes(ctypes.byref(desc), 1)
            if nb2 == 0:
                continue
            wpk2 = torch.empty((nb2 // 4,), dtype=torch.float32, device=dev)
            call("iswm_conv2d_pl2_pack_weights", ctypes.byref(desc), 1, _p(wo), _p(wpk2), _stream())
            f_new = lambda: call("iswm_conv2d_dgrad_pl2", ctypes.byref(desc), _p(dyp), dyp.shape[1] * cout, _p(wpk2),
                                 _p(dx_new), 0, _stream())
            f_ref(); f_new()
            torch.cuda.synchronize()
            err = float((dx_ref - dx_new).abs().max() / dx_ref.abs().max())
            same = err < 5e-6
            if not same:
                print("   dgrad err", err)
            name = "dgrad"
        for _ in range(3):
            f_ref(); f_new()
        r_med, r_min = time_us(f_ref, ROUNDS)
        n_med, n_min = time_us(f_new, ROUNDS)
        r2_med, _ = time_us(f_ref, ROUNDS)
        n2_med, _ = time_us(f_new, ROUNDS)
        fl = g.flops()
        print("%-34s %s equal=%s  fp32-in %.1f us (%.0f TF)  planes %.1f us (%.0f TF)  x%.2f   [2nd: %.1f / %.1f]" %
              (tag, name, same, r_med, fl / r_med * 1e-6, n_med, fl / n_med * 1e-6, r_med / n_med, r2_med, n2_med), flush=True)
        assert same or NOCHECK, "planes kernel differs from the fp32-input kernel"
