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
# the conv launches of the second step in order (kernel name, geometry) -> PMC_SEQ file, for pmc_by_geometry.py
from iswm_amd import ops
seq_path = os.environ.get("PMC_SEQ")
if seq_path:
    ops.KPROF = ops.KernelProfile()
step()
torch.cuda.synchronize()
if seq_path:
    import json
    json.dump([[r[0], r[4], r[3]] for r in ops.KPROF.records], open(seq_path, "w"))
    ops.KPROF = None
print("profiled steps done", flush=True)
