"""Worker of tests/test_ddp_gpu.py: launched twice by torch.distributed.run (gloo rendezvous, BOTH ranks on cuda:0 --
the only GPU a test box has; the driver's real multi-GPU runs use RCCL, one GPU per rank).  Each rank trains one step on
its half of a global batch through DistributedDataParallelHIP + the globally normalised criterion; rank 0 then replays
both halves in one process and checks that the all-reduced gradients equal the sum of the per-half gradients of the
GLOBAL weighted-mean loss (local BatchNorm per rank, SUM of gradients, global normaliser) and that both
ranks ended with identical parameters after the optimizer step."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from iswm_amd.network import modeling
    from iswm_amd.optim import FusedSGD
    from iswm_amd.parallel import DistributedDataParallelHIP
    from iswm_amd.utils.loss import CrossEntropyLoss

    torch.manual_seed(100 + rank)                      # different init per rank: the broadcast must fix it
    model = modeling.deeplabv3plus_resnet50(num_classes=2, output_stride=16).to(dev).train()
    model.classifier.aspp.project[3].p = 0.0           # dropout off: its counter-based masks depend on the call index
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 3, 65, 65, generator=g)
    lab = (torch.rand(8, 65, 65, generator=g) < 0.2).to(torch.int64)
    lab[torch.rand(8, 65, 65, generator=g) < 0.05] = 255
    lab[:4, :40] = 0                                   # unequal class mix per half -> different local normalisers
    w = torch.tensor([1.0, 3.0])

    net = DistributedDataParallelHIP(model, process_group=dist.group.WORLD, bucket_mb=8.0)
    init = {k: v.detach().clone() for k, v in model.state_dict().items()}          # after the rank-0 broadcast
    opt = FusedSGD(model.parameters(), momentum=0.9, weight_decay=1e-4, nesterov=True)
    net.attach(opt)
    crit = CrossEntropyLoss(weight=w, ignore_index=255, group=dist.group.WORLD).to(dev)
    sl = slice(4 * rank, 4 * rank + 4)
    loss = crit(net(x[sl].to(dev)), lab[sl].to(dev))
    opt.zero_grad()
    loss.backward()
    net.finish_grad_sync()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    torch.cuda.synchronize()
    after = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    gathered = [torch.empty_like(after) for _ in range(world)]
    dist.all_gather(gathered, after)
    ok = bool(torch.equal(gathered[0], gathered[1]))                                # replicas stay identical
    losses = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(losses, loss.detach().cpu().reshape(1))
    ok = ok and float(losses[0]) == float(losses[1])                                # the global loss, on every rank

    if rank == 0:
        ref = modeling.deeplabv3plus_resnet50(num_classes=2, output_stride=16).to(dev).train()
        ref.classifier.aspp.project[3].p = 0.0
        ref.load_state_dict(init, strict=True)
        local = CrossEntropyLoss(weight=w, ignore_index=255).to(dev)
        valid = lab != 255
        den = [float(w[lab[h][valid[h]]].sum()) for h in (slice(0, 4), slice(4, 8))]
        total, acc = 0.0, None
        for h, dh in zip((slice(0, 4), slice(4, 8)), den):
            for p in ref.parameters():
                p.grad = None
            lh = local(ref(x[h].to(dev)), lab[h].to(dev))            # weighted mean over THIS half
            (lh * (dh / sum(den))).backward()                        # its share of the global weighted mean
            total += float(lh.detach()) * dh / sum(den)
            gs = {k: p.grad.detach().clone() for k, p in ref.named_parameters()}
            acc = gs if acc is None else {k: acc[k] + gs[k] for k in acc}
        worst = max((float((grads[k] - acc[k]).abs().max()) / max(float(acc[k].abs().max()), 1e-30), k) for k in acc)
        print("DDP_PARITY loss %.8f ref %.8f worst_grad_rel %.3e (%s) replicas_equal %s" %
              (float(loss.detach()), total, worst[0], worst[1], ok), flush=True)
        ok = ok and abs(float(loss.detach()) - total) <= 1e-6 * abs(total) and worst[0] <= 1e-5
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if float(flag) == 1.0 else 1)


if __name__ == "__main__":
    main()
