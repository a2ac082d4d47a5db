"""Where a 32-pixel step of the wide planes weight-gradient kernel spends its cycles: shader-clock stamps of workgroup 0
(waves 0 and 4, which share a SIMD).  usage: wgrad_timeline.py [cin cout k pad dil]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import _lib, ops
dev = torch.device("cuda:0")
a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [1024, 256, 1, 0, 1]
cin, cout, k, pad, dil = a
xh = ops.split_planes(torch.randn(16, 33, 33, cin, device=dev))
dyh = ops.split_planes(torch.randn(16, 33, 33, cout, device=dev))
g = ops.ConvGeom(xh, cout, k, k, 1, pad, dil)
for _ in range(int(os.environ.get("WG_WARM_LAUNCHES", "20000"))):      # >= 2 s of load: the clock the chip settles on is part of the answer
    ops.conv2d_wgrad(xh, dyh, g)
buf = torch.zeros(512, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.iswm_set_debug_buffer(buf.data_ptr())
ops.conv2d_wgrad(xh, dyh, g)
torch.cuda.synchronize()
lib.iswm_set_debug_buffer(None)
b = buf.cpu().tolist()
if b[503] > b[501]:
    cyc, ref = b[502] - b[500], b[503] - b[501]
    print("workgroup 0: %d shader cycles in %.2f us -> %.2f GHz in-kernel clock" % (cyc, ref / 100.0, cyc / (ref / 100.0) / 1e3))
names = ["wait vmcnt", "barrier", "next+issue", "multiply"]
for wv, base in ((0, 0), (4, 256)):
    print("wave %d: step | %s | to next step" % (wv, " | ".join(names)))
    for s in range(40):
        t = b[base + s * 6: base + s * 6 + 5]
        if t[0] == 0 or t[4] == 0:
            break
        d = [t[i + 1] - t[i] for i in range(4)]
        nxt = b[base + (s + 1) * 6]
        print("   %3d | %s | %6d" % (s, " | ".join("%10d" % v for v in d), nxt - t[0] if nxt else -1))
