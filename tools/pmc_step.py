"""Two training steps of the bench workload (r101 os16, 16 x 513 x 513) for rocprofv3 --pmc passes.
The second step is bracketed by marker kernels (a tiny torch fill of a tagged size) so pmc_aggregate.py can
cut it out of the trace."""
import sys, os
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
step()
torch.cuda.synchronize()
print("warm step done", flush=True)
step()
torch.cuda.synchronize()
print("profiled steps done", flush=True)
