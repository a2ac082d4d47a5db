"""Does running the weight-gradient and data-gradient kernels of one layer on two streams beat running them
back to back?  (layer3 geometries, batch 16 at 33x33)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import ops


def main():
    dev = torch.device("cuda:0")
    side = torch.cuda.Stream()
    for (cin, cout, k, p) in ((256, 1024, 1, 0), (1024, 256, 1, 0), (256, 256, 3, 1)):
        x = torch.randn(16, 33, 33, cin, device=dev)
        g = ops.ConvGeom(x, cout, k, k, 1, p, 1)
        dy = torch.randn(16, 33, 33, cout, device=dev)
        w = torch.randn(cout, k, k, cin, device=dev) * 0.05
        dw = torch.empty_like(w)
        dx = torch.empty_like(x)
        def seq():
            ops.conv2d_wgrad(x, dy, g, dw)
            ops.conv2d_dgrad(dy, w, g, tuple(x.shape), dx=dx)
        def par():
            ev = torch.cuda.Event(); ev.record()
            side.wait_event(ev)
            with torch.cuda.stream(side):
                ops.conv2d_wgrad(x, dy, g, dw)
            ops.conv2d_dgrad(dy, w, g, tuple(x.shape), dx=dx)
            ev2 = torch.cuda.Event(); ev2.record(side)
            torch.cuda.current_stream().wait_event(ev2)
        for name, fn in (("sequential", seq), ("two streams", par)):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): fn()
            b.record(); torch.cuda.synchronize()
            print("c%d->%d k%d: %-12s %.1f us per (wgrad + dgrad)" % (cin, cout, k, name, a.elapsed_time(b) * 50))


if __name__ == "__main__":
    main()
