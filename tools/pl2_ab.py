"""Same-process A/B of the planes conv kernel with and without its diagnostic stamps (the stamped instantiation is the
production loop plus six untaken branches per stage): alternates the two for a few shapes and prints median microseconds.
usage: pl2_ab.py [rounds]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import _lib, ops
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib = _lib.load()
buf = torch.zeros(512, dtype=torch.int64, device=dev)
for (cin, cout, k, pad, dil) in ((1024, 256, 1, 0, 1), (256, 1024, 1, 0, 1), (2048, 512, 1, 0, 1), (256, 256, 3, 1, 1), (512, 512, 3, 2, 2)):
    xh = ops.split_planes(torch.randn(16, 33, 33, cin, device=dev))
    w = torch.randn(cout, k, k, cin, device=dev) * 0.05
    g = ops.ConvGeom(xh, cout, k, k, 1, pad, dil)
    if lib.iswm_conv2d_pl2_tile_rows(ctypes_desc := __import__("ctypes").byref(g.desc(cin, cout)), 0) != 144:
        continue
    for _ in range(5):
        ops.conv2d_fwd(xh, w, g, want_stats=True)
    t = {0: [], 1: []}
    for r in range(rounds):
        for v in (0, 1):
            lib.iswm_set_debug_buffer(buf.data_ptr() if v else None)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.conv2d_fwd(xh, w, g, want_stats=True)
            b.record()
            torch.cuda.synchronize()
            t[v].append(a.elapsed_time(b) * 1e3)
    lib.iswm_set_debug_buffer(None)
    m0, m1 = sorted(t[0])[rounds // 2], sorted(t[1])[rounds // 2]
    print("c%d->%d k%d d%d fwd: production %.1f us, stamped %.1f us (x%.3f)" % (cin, cout, k, dil, m0, m1, m1 / m0), flush=True)
