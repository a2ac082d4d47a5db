"""GPU tuning harness: per-geometry time and TFLOP/s of the conv kernels inside a real training step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iswm_amd import ops
from iswm_amd.network import modeling
from iswm_amd.utils.loss import CrossEntropyLoss
model_name = sys.argv[1] if len(sys.argv) > 1 else "resnet101"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
torch.manual_seed(1)
m = getattr(modeling, "deeplabv3plus_" + model_name)(num_classes=2, output_stride=16).to(dev).train()
x = torch.randn(B, 3, 513, 513, device=dev)
lab = (torch.rand(B, 513, 513, device=dev) < 0.1).long()
crit = CrossEntropyLoss(weight=torch.tensor([1.0, 3.0])).to(dev)
def step():
    for p in m.parameters(): p.grad = None
    crit(m(x), lab).backward()
step(); step()
torch.cuda.synchronize()
ops.KPROF = ops.KernelProfile()
reps = 3
for _ in range(reps): step()
torch.cuda.synchronize()
g = ops.KPROF.by_geometry(); ops.KPROF = None
rows = sorted(g.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in g.values()) / reps
print("total conv ms/step %.2f" % tot)
print("%-32s %-34s %6s %9s %8s %7s" % ("kernel", "geometry", "calls", "ms/step", "us/call", "TF"))
for (name, tag), v in rows:
    print("%-32s %-34s %6d %9.3f %8.1f %7.1f" % (name, tag, v["launches"] // reps, v["ms"] / reps, v["ms"] * 1e3 / v["launches"], v["flops"] / v["ms"] / 1e9))
