"""Host-side cost of one training step: wall time to enqueue a step from an idle GPU, thread CPU time, GPU time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd.network import modeling
from iswm_amd.optim import FusedSGD
from iswm_amd.utils.loss import CrossEntropyLoss
dev = torch.device("cuda:0")
m = modeling.deeplabv3plus_resnet101(num_classes=2, output_stride=16).to(dev).train()
opt = FusedSGD(m.parameters(), momentum=0.9, weight_decay=1e-4, nesterov=True)
crit = CrossEntropyLoss(weight=torch.tensor([1.0, 3.0])).to(dev)
x = torch.randn(16, 3, 513, 513, device=dev); lab = (torch.rand(16, 513, 513, device=dev) < 0.1).long()
def step():
    loss = crit(m(x), lab); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for i in range(5):
    torch.cuda.synchronize()
    c0 = time.thread_time(); t0 = time.perf_counter()
    step()
    t1 = time.perf_counter(); c1 = time.thread_time()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("step %d: enqueue wall %.1f ms, main-thread cpu %.1f ms, until GPU idle %.1f ms" % (i, (t1 - t0) * 1e3, (c1 - c0) * 1e3, (t2 - t0) * 1e3), flush=True)
# back-to-back (queue stays full)
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("5 steps back-to-back: enqueue %.1f ms/step, total %.1f ms/step" % ((t1 - t0) * 200, (t2 - t0) * 200))
