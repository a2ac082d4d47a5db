import sys, os, cProfile, pstats
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
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
