"""CPU-side checks (no GPU): the C-ABI library builds, loads and exports every symbol the
header declares; the drop-in construction API yields the reference's state_dict layout; the
product refuses CPU tensors (no fallback); data-parallel bucketing all-reduces correctly
over gloo with world_size 2."""
import os
import re
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from iswm_amd import _lib
    from iswm_amd.build import build
    build(verbose=False)
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "iswm_hip.h")).read()
    declared = set(re.findall(r"\b(iswm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.iswm_version() >= 100


def test_stat_tile_queries_are_pure_host_functions():
    from iswm_amd import _lib
    import ctypes
    lib = _lib.load()
    d = _lib.ConvDesc(16, 33, 33, 2048, 33, 33, 256, 3, 3, 1, 6, 6, 2048, 256)
    rows = lib.iswm_conv2d_stat_tile_rows(ctypes.byref(d))
    assert rows in (64, 128) and lib.iswm_conv2d_stat_tiles(ctypes.byref(d)) == (16 * 33 * 33 + rows - 1) // rows
    assert lib.iswm_conv2d_wgrad_workspace(ctypes.byref(d)) % (256 * 9 * 2048 * 4) == 0
    # planes kernels: tile height picked so that tiles ~ k x 256 CUs (33x33x16 -> 144 rows x 128 columns: 242 tiles)
    d1 = _lib.ConvDesc(16, 33, 33, 1024, 33, 33, 256, 1, 1, 1, 0, 1, 1024, 256)
    assert lib.iswm_conv2d_pl2_tile_rows(ctypes.byref(d1), 0) == 144
    assert lib.iswm_conv2d_pl2_weight_bytes(ctypes.byref(d1), 0) == 16 * (1024 // 32) * 3 * 64 * 16      # [256/16][K/32][3][64] x 16 B
    d48 = _lib.ConvDesc(16, 129, 129, 256, 129, 129, 48, 1, 1, 1, 0, 1, 256, 48)
    assert lib.iswm_conv2d_pl2_weight_bytes(ctypes.byref(d48), 1) == 0                                     # dgrad gathers 48 channels: not a multiple of 64
    assert lib.iswm_conv2d_wgrad_planes_ok(ctypes.byref(d48)) == 1 and lib.iswm_conv2d_wgrad_planes_workspace(ctypes.byref(d48)) % (48 * 256 * 4) == 0
    assert lib.iswm_colstat_tiles(1) == 1 and lib.iswm_colstat_tiles(10 ** 7) == 1024
    assert lib.iswm_colstat_tile_rows(578) == 31 and lib.iswm_colstat_tile_rows(1) == 1
    assert lib.iswm_loss_blocks(513 * 513 * 16) == (513 * 513 * 16 + 1023) // 1024      # 4 pixels per thread, 256 threads
    assert lib.iswm_loss_blocks(10 ** 9) == 8192 and lib.iswm_loss_blocks(1) == 1


@pytest.mark.parametrize("ctor,backbone,nkeys", [("deeplabv3plus_resnet50", "resnet50", 374),
                                                 ("deeplabv3plus_resnet101", "resnet101", 680)])
def test_construction_api_matches_reference_layout(ctor, backbone, nkeys):
    from iswm_amd.network import modeling
    from oracle.synth import ArchCfg, param_shapes
    m = getattr(modeling, ctor)(num_classes=2, output_stride=16)
    sd = m.state_dict()
    want = param_shapes(ArchCfg("deeplabv3plus", backbone, 2, 16))
    assert len(sd) == nkeys
    assert list(sd.keys()) == list(want.keys())
    for k, s in want.items():
        assert tuple(sd[k].shape) == tuple(s), k
    w = m.backbone.layer4[2].conv2.weight
    assert w.permute(0, 2, 3, 1).is_contiguous()          # OHWI in memory: what the kernels read
    assert m.backbone.layer4[1].conv2.dilation == (2, 2) and m.backbone.layer4[0].conv2.dilation == (1, 1)
    m8 = modeling._segm_resnet("deeplabv3plus", "resnet50", 2, 8, False)
    assert m8.backbone.layer3[1].conv2.dilation == (2, 2) and m8.backbone.layer4[1].conv2.dilation == (4, 4)
    assert [c[1][0].dilation[0] for c in list(enumerate(m8.classifier.aspp.convs))[1:4]] == [12, 24, 36]


def test_error_conventions():
    from iswm_amd.network import modeling, utils
    from iswm_amd.network.backbone import resnet
    with pytest.raises(NotImplementedError):
        modeling._load_model("deeplabv3plus", "xception", 2, 16, False)
    with pytest.raises(ValueError):
        utils.IntermediateLayerGetter(resnet.resnet50(), {"nope": "out"})
    with pytest.raises(ValueError):
        resnet.resnet50(replace_stride_with_dilation=[False, True])
    names = sorted(n for n in modeling.__dict__ if n.islower() and not n.startswith("_")
                   and callable(modeling.__dict__[n]))
    assert "deeplabv3plus_resnet50" in names and "deeplabv3plus_resnet101" in names   # train.py:284-289 scan


def test_no_cpu_fallback():
    from iswm_amd.network import modeling
    from iswm_amd.utils.loss import CrossEntropyLoss
    m = modeling.deeplabv3plus_resnet50(num_classes=2, output_stride=16)
    with pytest.raises(ValueError):
        m(torch.zeros(2, 3, 33, 33))
    with pytest.raises(ValueError):
        CrossEntropyLoss()(torch.zeros(1, 2, 4, 4), torch.zeros(1, 4, 4, dtype=torch.int64))


def test_oracle_is_not_imported_by_the_product():
    pkg = os.path.join(ROOT, "iswm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(dirpath, f)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iswm_amd.parallel import DistributedDataParallelHIP
    torch.manual_seed(rank)                       # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(300, 70), torch.nn.Linear(70, 5000), torch.nn.Linear(5000, 3))
    ddp = DistributedDataParallelHIP(net, bucket_mb=0.5)
    params = list(net.parameters())
    w0 = params[0].detach().clone()
    assert len(ddp.buckets) >= 2
    # buckets tile the arena back to front with no gaps
    assert ddp.buckets[0][1] == ddp.arena.numel and ddp.buckets[-1][0] == 0
    for (s0, e0), (s1, e1) in zip(ddp.buckets, ddp.buckets[1:]):
        assert e1 == s0
    ddp._left = list(ddp._need)
    for i in range(len(params) - 1, -1, -1):          # backward order
        p = params[i]
        p.grad = p._iswm_grad_view
        p.grad.fill_(float(rank + 1) * (i + 1))
        ddp._on_ready(p)
    ddp.finish_grad_sync()
    ok = all(bool((p.grad == 3.0 * (i + 1)).all()) for i, p in enumerate(params))   # 1 + 2 summed
    q.put((rank, ok, w0))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_bucketed_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] and res[1][1]
    assert torch.equal(res[0][2], res[1][2])          # rank-0 parameters were broadcast


def test_c_abi_argument_validation_without_a_gpu():
    """every entry point validates on the host before any launch: status 1 + a message, never a crash
    (null pointers / bad geometry never reach the device)"""
    import ctypes
    from iswm_amd import _lib
    lib = _lib.load()
    err = lambda: lib.iswm_last_error().decode()
    d = _lib.ConvDesc(2, 17, 17, 64, 17, 17, 64, 3, 3, 1, 1, 1, 64, 64)
    assert lib.iswm_conv2d_fwd(ctypes.byref(d), None, None, None, None, None, None) == 1 and "null" in err()
    assert lib.iswm_conv2d_fwd_packed(ctypes.byref(d), None, None, None, None, None, None) == 1 and "null" in err()
    assert lib.iswm_conv2d_dgrad_packed(ctypes.byref(d), None, None, None, 0, None) == 1
    assert lib.iswm_conv2d_pack_weights(ctypes.byref(d), 7, None, None, None) == 1 and "kind" in err()
    bad = _lib.ConvDesc(2, 17, 17, 64, 9, 9, 64, 3, 3, 1, 1, 1, 64, 64)          # Ho/Wo inconsistent with the geometry
    assert lib.iswm_conv2d_fwd(ctypes.byref(bad), None, None, None, None, None, None) == 1
    dw = _lib.ConvDesc(2, 17, 17, 64, 17, 17, 32, 3, 3, 1, 1, 1, 64, 64)         # depthwise needs Cin == Cout
    assert lib.iswm_dwconv2d_fwd(ctypes.byref(dw), None, None, 64, None, None, None) == 1 and "depthwise" in err()
    assert lib.iswm_confusion_matrix(None, 0, None, 1, 10, 2, None, None) == 1
    buf = (ctypes.c_longlong * 4)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.iswm_confusion_matrix(p, 0, p, 1, 10, 99, p, None) == 1 and "n_classes" in err()
    assert lib.iswm_confusion_matrix(p, 3, p, 1, 10, 2, p, None) == 1 and "dtype" in err()
    assert lib.iswm_augment_batch(None, None, None, None, 1, 8, 8, None, None, None, None, None) == 1
    assert lib.iswm_set_conv_math(5) == 1
    # packed-weight sizes are pure host arithmetic: [ceil(N/64)*2 column blocks][K/16][3 planes][64 lanes] x 16 B
    old = lib.iswm_get_conv_math()
    try:
        lib.iswm_set_conv_math(1)
        assert lib.iswm_conv2d_packed_weight_bytes(ctypes.byref(d), 0) == 2 * (9 * 64 // 16) * 3 * 64 * 16
        d48 = _lib.ConvDesc(2, 17, 17, 256, 17, 17, 48, 1, 1, 1, 0, 1, 256, 48)
        assert lib.iswm_conv2d_packed_weight_bytes(ctypes.byref(d48), 0) == 2 * (256 // 16) * 3 * 64 * 16
        assert lib.iswm_conv2d_packed_weight_bytes(ctypes.byref(d48), 1) == 0      # Cout = 48 is not 32-aligned
        tiles, rows = ctypes.c_int(0), ctypes.c_int(0)
        big = _lib.ConvDesc(16, 33, 33, 256, 33, 33, 256, 3, 3, 1, 1, 1, 256, 256)
        assert lib.iswm_conv2d_fwd_packed_stat_layout(ctypes.byref(big), ctypes.byref(tiles), ctypes.byref(rows)) == 0
        assert (tiles.value, rows.value) == (16 * 9, 0)                             # 11 x 11 patches, per-tile counts
        lib.iswm_set_conv_math(0)
        assert lib.iswm_conv2d_packed_weight_bytes(ctypes.byref(d), 0) == 0         # exact-fp32 path: no packed form
    finally:
        lib.iswm_set_conv_math(old)


def test_calculate_class_weights_matches_oracle():
    """the product's calculate_class_weights (train.py:388-410) on a loader of (image, label) batches and on dict
    batches, against the oracle's restatement of the same formula"""
    from iswm_amd.utils.loss import calculate_class_weights
    from oracle import loss as oloss
    g = torch.Generator().manual_seed(5)
    labs = [(torch.rand(3, 17, 19, generator=g) < 0.13).to(torch.uint8) for _ in range(4)]
    labs[1][0, :3] = 255                                           # ignored pixels count for neither class
    want = oloss.class_weights(torch.cat([x.reshape(-1) for x in labs]))
    got = calculate_class_weights([(torch.zeros(3), x) for x in labs])
    assert got.dtype == torch.float32 and torch.equal(got, want)
    assert torch.equal(calculate_class_weights([{"mask": x} for x in labs]), want)


def _cw_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iswm_amd.utils.loss import calculate_class_weights
    g = torch.Generator().manual_seed(11)
    full = [(torch.rand(2, 16, 16, generator=g) < (0.05 + 0.1 * i)).to(torch.uint8) for i in range(6)]
    shard = full[rank::world]                                      # what a DistributedSampler hands this rank
    q.put((rank, calculate_class_weights([(None, x) for x in shard]), calculate_class_weights.__module__))
    dist.barrier()
    dist.destroy_process_group()


def test_class_weights_are_global_under_data_parallelism():
    """every rank iterates its own shard but must end up with the weights of the WHOLE train set (the criterion's global
    normaliser assumes one weight vector): int64 all-reduce of the two pixel counts"""
    from iswm_amd.utils.loss import calculate_class_weights
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cw_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(11)
    full = [(torch.rand(2, 16, 16, generator=g) < (0.05 + 0.1 * i)).to(torch.uint8) for i in range(6)]
    want = calculate_class_weights([(None, x) for x in full])
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][1], want)


def _ddp_accum_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from iswm_amd.parallel import DistributedDataParallelHIP
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 30), torch.nn.Linear(30, 2000), torch.nn.Linear(2000, 3))
    for p in net[0].parameters():
        p.requires_grad_(False)                                   # a frozen layer: never reports, its bucket never fires
    ddp = DistributedDataParallelHIP(net, bucket_mb=0.1)
    params = list(net.parameters())

    def backward(scale):                                           # stands in for the hand-written backward
        for i in range(len(params) - 1, -1, -1):
            p = params[i]
            if not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = p._iswm_grad_view
                p.grad.zero_()
            p.grad.add_(scale * float(rank + 1) * (i + 1))
            ddp._on_ready(p)

    ddp._left = list(ddp._need)
    with ddp.no_sync():                                            # first micro-batch: local accumulation only
        backward(1.0)
    assert not ddp._works
    ddp._left = list(ddp._need)
    backward(10.0)                                                 # second micro-batch: the accumulated sum is all-reduced
    ddp.finish_grad_sync()
    ok = all(bool((p.grad == 11.0 * 3.0 * (i + 1)).all()) for i, p in enumerate(params) if p.requires_grad)
    frozen_untouched = all(p.grad is None for p in net[0].parameters())
    st = ddp.comm_stats()
    q.put((rank, ok and frozen_untouched, st["bytes_allreduced"], st["allreduces"], st["world_size"]))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gradient_accumulation_and_frozen_parameters_gloo_world2():
    """two backward calls per optimizer step (no_sync around the first) sum correctly across ranks; a frozen layer
    neither blocks its bucket nor receives a gradient; the communication counters add up"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_accum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in res:
        assert r[1] and r[4] == 2 and r[3] >= 1 and r[2] > 0
    assert res[0][2] == res[1][2]


def test_fused_optimizer_loads_stock_torch_state():
    """a checkpoint written by the reference holds torch.optim.AdamW / SGD state (train.py:567-582): the fused classes
    restore its moments into their arenas and keep their own hyper-parameter keys"""
    from iswm_amd.optim import FusedAdamW, FusedSGD
    torch.manual_seed(3)
    ref_p = [torch.nn.Parameter(torch.randn(7, 5)), torch.nn.Parameter(torch.randn(11))]
    ref = torch.optim.AdamW(ref_p, weight_decay=1e-2)
    for p in ref_p:
        p.grad = torch.randn_like(p)
    ref.step()
    sd = ref.state_dict()
    mine_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    opt = FusedAdamW(mine_p, weight_decay=1e-2)
    opt.load_state_dict(sd)
    assert opt.param_groups[0]["decoupled"] is True and opt._t == 1
    for p, q in zip(mine_p, ref_p):
        assert torch.equal(opt.state[p]["exp_avg"], ref.state[q]["exp_avg"])
        assert torch.equal(opt.state[p]["exp_avg_sq"], ref.state[q]["exp_avg_sq"])
        assert opt.state[p]["exp_avg"].data_ptr() >= opt._m.data_ptr()          # restored INTO the arena
    ref2 = torch.optim.SGD(ref_p, lr=1e-3, momentum=0.9, nesterov=True, weight_decay=1e-4)
    ref2.step()
    opt2 = FusedSGD([torch.nn.Parameter(p.detach().clone()) for p in ref_p], momentum=0.9, nesterov=True, weight_decay=1e-4)
    opt2.load_state_dict(ref2.state_dict())
    for p, q in zip(opt2.arena.params, ref_p):
        assert torch.equal(opt2.state[p]["momentum_buffer"], ref2.state[q]["momentum_buffer"])
    assert opt2.param_groups[0]["nesterov"] is True


def test_resume_scoring_of_a_reference_checkpoint():
    """the reference stores best_score as a dictionary of metrics next to `weighted_score` (train.py:567-582, 799-811)"""
    from iswm_amd.train import scalar_score
    assert scalar_score(0.71) == 0.71
    assert scalar_score({"Foreground IoU": 0.5, "Foreground F1": 0.7}, 0.63) == 0.63
    assert abs(scalar_score({"Foreground IoU": 0.5, "Foreground F1": 0.7}) - 0.6) < 1e-12


def test_planes_predicate_matches_the_kernel_preconditions():
    """iswm_conv2d_wgrad_planes_ok must say no exactly where iswm_conv2d_wgrad_planes would refuse:
    sizes beyond the 2^30-element indexing bound and unaligned pitches fall back to the fp32-input weight gradient"""
    import ctypes
    from iswm_amd import _lib
    from iswm_amd._lib import ConvDesc
    lib = _lib.load()
    ok = lambda *a: lib.iswm_conv2d_wgrad_planes_ok(ctypes.byref(ConvDesc(*a)))
    assert ok(16, 33, 33, 256, 33, 33, 256, 3, 3, 1, 1, 1, 256, 256) == 1
    assert ok(16, 33, 33, 256, 33, 33, 256, 3, 3, 1, 1, 1, 260, 256) == 0        # pitch not a multiple of 8
    assert ok(16, 33, 33, 252, 33, 33, 256, 3, 3, 1, 1, 1, 256, 256) == 0        # channels not a multiple of 8
    assert ok(64, 513, 513, 64, 513, 513, 64, 3, 3, 1, 1, 1, 64, 64) == 0        # 2^30 elements and beyond
    assert ok(16, 33, 33, 256, 32, 33, 256, 3, 3, 1, 1, 1, 256, 256) == 0        # output size does not match geometry


def test_fused_adam_loads_a_stock_checkpoint_with_stateless_parameters():
    """a torch.optim.AdamW state_dict has no entry for a parameter that never received a gradient (a frozen layer of the
    reference's run): loading it must not raise, the frozen parameter keeps zero moments, the others land in the arena"""
    from iswm_amd.optim import FusedAdamW
    torch.manual_seed(0)
    a, b, c = (torch.nn.Parameter(torch.randn(4, 3)) for _ in range(3))
    ref = torch.optim.AdamW([a, b, c], lr=1e-3, weight_decay=1e-2)
    a.grad, c.grad = torch.randn(4, 3), torch.randn(4, 3)           # b stays without a gradient -> no state entry
    ref.step()
    ref.step()
    sd = ref.state_dict()
    assert 1 not in sd["state"]
    p = [torch.nn.Parameter(t.detach().clone()) for t in (a, b, c)]
    opt = FusedAdamW(p, lr=1e-3, weight_decay=1e-2)
    opt.load_state_dict(sd)
    assert opt._t == 2
    assert torch.equal(opt.state[p[0]]["exp_avg"], ref.state[a]["exp_avg"])
    assert torch.equal(opt.state[p[2]]["exp_avg_sq"], ref.state[c]["exp_avg_sq"])
    assert float(opt.state[p[1]]["exp_avg"].abs().max()) == 0.0 and int(opt.state[p[1]]["step"]) == 0
    for i, q in enumerate(p):                                        # the state tensors ARE the arena views
        assert opt.state[q]["exp_avg"].data_ptr() == opt.arena.view_of(opt._m, i).data_ptr()


def test_aspp_plan_orders_rows_by_tap_set_and_covers_every_tap():
    """iswm_aspp_plan (host function): the row order of every job is a permutation of the pixels, rows are grouped by their set
    of in-bounds taps (heaviest first), every tile's tap mask is exactly the union of its rows' sets -- so no in-bounds
    (pixel, tap) pair is dropped and a uniform tile multiplies no padding -- and the tile table lists every (job, tile) once"""
    import ctypes
    import numpy as np
    from iswm_amd import _lib
    lib = _lib.load()
    n, h, w, cin, cout = 3, 33, 35, 128, 256
    ksize, dil = [1, 3, 3, 3], [1, 6, 12, 18]
    d = _lib.ConvDesc(n, h, w, cin, h, w, cout, 1, 1, 1, 0, 1, cin, cout)
    ks, dl = (ctypes.c_int * 4)(*ksize), (ctypes.c_int * 4)(*dil)
    for kind in (0, 1):
        nb = lib.iswm_aspp_plan_bytes(ctypes.byref(d), 4, ks, dl, kind)
        assert nb > 0
        buf = np.zeros(nb, dtype=np.uint8)
        assert lib.iswm_aspp_plan(ctypes.byref(d), 4, ks, dl, kind, buf.ctypes.data_as(ctypes.c_void_p), 256) == 0
        hdr = buf[:72].view(np.int32)                       # struct AsppPlan (csrc/conv_mfma_pl2t.hip): 18 ints, taps[32], jobs[4]
        magic, kd, nbr, ntaps, njobs, ntiles, N, H, W, M, MT, NT, GC, NC, rm_off, tl_off, total = hdr[:17]
        assert (kd, nbr, ntaps, N, H, W, M) == (kind, 4, 28, n, h, w, n * h * w) and total == nb
        assert njobs == (1 if kind else 4) and MT == (M + 143) // 144
        assert (GC, NC, NT) == ((cout, cin, 1) if kind else (cin, cout, 2))
        taps = buf[72:72 + 32 * 16].view(np.int32).reshape(32, 4)[:ntaps]          # dh, dw, branch, k32base
        jobs = buf[72 + 512:72 + 512 + 4 * 32].view(np.int32).reshape(4, 8)[:njobs]  # tap_begin, ntaps, branch, rowmap_off
        rowmap = buf[rm_off:rm_off + njobs * M * 4].view(np.int32).reshape(njobs, M)
        tiles = buf[tl_off:tl_off + ntiles * 16].view(np.int32).reshape(ntiles, 4)   # job, mt, nt, mask
        assert ntiles == njobs * MT * NT
        assert len({tuple(t[:3]) for t in tiles.tolist()}) == ntiles                   # every (job, mt, nt) once
        pop = np.array([bin(int(m) & 0xFFFFFFFF).count("1") for m in tiles[:, 3]])
        assert pop[0] == pop.max() and pop[:min(256, ntiles)].min() >= pop[min(256, ntiles):2 * 256].max(initial=0) - 0  # heaviest band first
        for j in range(njobs):
            tb, nt_ = int(jobs[j, 0]), int(jobs[j, 1])
            code = rowmap[j]
            pn, ph, pw = code >> 20, (code >> 10) & 1023, code & 1023
            assert len(np.unique(code)) == M and pn.max() == n - 1 and ph.max() == h - 1 and pw.max() == w - 1
            mask = np.zeros(M, dtype=np.int64)
            for t in range(nt_):
                ok = (ph + taps[tb + t, 0] >= 0) & (ph + taps[tb + t, 0] < h) & (pw + taps[tb + t, 1] >= 0) & (pw + taps[tb + t, 1] < w)
                mask |= ok.astype(np.int64) << t
            cnt = np.array([bin(int(m)).count("1") for m in mask])
            assert (np.diff(cnt) <= 0).all()                                           # heaviest tap sets first
            assert len(np.unique(mask)) == len(np.unique(mask[np.r_[True, np.diff(mask) != 0]]))   # equal sets are contiguous
            for tl in tiles[tiles[:, 0] == j]:
                rows = mask[tl[1] * 144:(tl[1] + 1) * 144]
                assert int(np.bitwise_or.reduce(rows)) == int(tl[3]) & 0xFFFFFFFF
    # geometries the fused kernel does not take answer 0 bytes (callers fall back to one launch per branch)
    bad = _lib.ConvDesc(n, h, w, 100, h, w, cout, 1, 1, 1, 0, 1, 100, cout)
    assert lib.iswm_aspp_plan_bytes(ctypes.byref(bad), 4, ks, dl, 0) == 0
