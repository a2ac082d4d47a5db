"""BASELINE configs[3]/[4] geometry smoke: r101 os8 @769 and r101 os16 @1024, one training step each under both
conv arithmetics; prints loss, step time and peak memory."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import _lib
from iswm_amd.network import modeling
from iswm_amd.optim import FusedSGD
from iswm_amd.utils.loss import CrossEntropyLoss
dev = torch.device("cuda:0")
lib = _lib.load()
for os_, size, b in ((8, 769, 4), (16, 1024, 4)):
    losses = []
    for math in (1, 0):
        lib.iswm_set_conv_math(math)
        torch.manual_seed(1)
        m = modeling.deeplabv3plus_resnet101(num_classes=2, output_stride=os_).to(dev).train()
        opt = FusedSGD(m.parameters(), momentum=0.9, weight_decay=1e-4, nesterov=True)
        crit = CrossEntropyLoss(weight=torch.tensor([1.0, 3.0])).to(dev)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(b, 3, size, size, generator=g).to(dev)
        lab = (torch.rand(b, size, size, generator=g) < 0.1).long().to(dev)
        def step():
            loss = crit(m(x), lab); opt.zero_grad(); loss.backward(); opt.step(); return loss
        l0 = float(step()); torch.cuda.synchronize()
        t0 = time.perf_counter(); l1 = float(step()); l2 = float(step()); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        losses.append((l0, l1, l2))
        print("os%d %dx%d B=%d math=%s: losses %.6f %.6f %.6f  %.1f ms/step  %.1f img/s  peak %.1f GB" % (
            os_, size, size, b, "bf16x6" if math else "f32", l0, l1, l2, dt * 1e3, b / dt,
            torch.cuda.max_memory_allocated() / 2**30), flush=True)
        del m, opt; torch.cuda.empty_cache()
    print("  rel diff of first loss between arithmetics: %.2e" % (abs(losses[0][0] - losses[1][0]) / abs(losses[1][0])))
