#!/usr/bin/env python3
"""Headline benchmark: training images/sec of DeepLabV3+-ResNet101 (output_stride 16) on
synthetic 513x513 tiles, 16 images per GPU (BASELINE.json configs[2]: global batch 128 on 8
GPUs = 16 per GPU; weak scaling), fp32, the train.py:1045-1049 step:

    logits = model(images); loss = criterion(logits, labels);
    optimizer.zero_grad(); loss.backward(); optimizer.step(); scheduler.step()

with SGD(momentum 0.9, nesterov, wd 1e-4, torch's default lr 1e-3), CosineAnnealingLR and the
class-weighted CE ([1, 3]) -- all on hand-written gfx950 kernels (libiswm_hip.so).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model resnet101|resnet50]
                    [--batch B] [--size S] [--no-cpu-baseline]

N > 1 is launched by the driver as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: bf16 MFMA, dense
# bf16x6 kernels issue six bf16 MFMAs per fp32 multiply-accumulate, so their roof in ALGORITHMIC fp32
# FLOP/s is the bf16 dense peak / 6
X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0


# HBM bytes per launch of each kernel, averaged over the launches of one training step of THIS workload, from
# two rocprofv3 PMC passes over tools/pmc_step.py (FETCH_SIZE, WRITE_SIZE; FETCH_SIZE doubled per the gfx950
# wide-read correction of the MI355X guide), aggregated by tools/pmc_aggregate.py into profiles/.  The table records a
# hash of the kernel sources it was measured on: `traffic` is reported only when that hash matches the sources of the
# build being timed; otherwise it is null and the stale figure is returned under its own name.
PMC_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_pmc", "step_traffic.json")


def csrc_hash():
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "iswm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """(bytes per launch | None, {"bytes", "csrc_sha16"} of a stale table | None)"""
    try:
        with open(PMC_TABLE) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, None
    ent = table.get(kernel.replace("+reduce", ""))
    if not ent:
        return None, None
    val = round(ent["hbm_bytes_per_launch"])
    sha = table.get("_csrc_sha16")
    if sha == csrc_hash():
        return val, None
    return None, {"bytes": val, "csrc_sha16": sha}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="resnet101", choices=["resnet50", "resnet101"])
    ap.add_argument("--output-stride", type=int, default=16, choices=[8, 16])
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=513)
    ap.add_argument("--torch-baseline", action="store_true",
                    help="also time the same step on stock PyTorch-ROCm ops (MIOpen) on this GPU (adds ~2 min)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--conv-math", default=None, choices=["f32", "bf16x6", "bf16"],
                    help="convolution arithmetic (default: library default = bf16x6, or $ISWM_CONV_MATH)")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra exact-fp32-MFMA measurement")
    return ap.parse_args()


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // period))
            break
        except Exception:
            continue
    return max(1, min(n, 64))


def cpu_baseline(model_name, output_stride, size):
    """The reference's CPU training step restated by oracle/ (kind "port"), timed on this host's
    cores on a bounded sample: batch 2 (the smallest the image-pooling BatchNorm accepts),
    1 warm-up + 10 timed steps of the same model and tile size (about 10-20 s)."""
    import torch
    from oracle import loss as oloss
    from oracle.deeplab import OracleDeepLab
    from oracle.optim import OracleSGD
    from oracle.synth import ArchCfg, synth_images, synth_labels, synth_state_dict
    cores = host_cores()
    torch.set_num_threads(cores)
    print("[bench] cpu_baseline: %d threads" % cores, file=sys.stderr, flush=True)
    cfg = ArchCfg("deeplabv3plus", model_name, 2, output_stride)
    o = OracleDeepLab(cfg, synth_state_dict(cfg), dropout_p=0.1).train()
    opt = OracleSGD(o.parameters())
    b = 2
    x = synth_images(b, size, size, seed=0)
    lab = synth_labels(b, size, size, seed=0)
    w = torch.tensor([1.0, 3.0])

    def step():
        loss = oloss.weighted_ce(o(x), lab, w)
        o.zero_grad()
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()
    print("[bench] cpu_baseline: warm-up step %.1f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    n = 10
    for i in range(n):
        step()
        print("[bench] cpu_baseline: step %d done at %.1f s" % (i, time.perf_counter() - t0), file=sys.stderr,
              flush=True)
    dt = time.perf_counter() - t0
    return {"value": round(b * n / dt, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d timed steps (1 warm-up) of oracle/ deeplabv3plus_%s os%d at %dx%d, batch %d, "
                      "fwd + weighted CE + bwd + SGD-nesterov, torch CPU fp32" %
                      (n, model_name, output_stride, size, size, b)}


def torch_rocm_baseline(model_name, output_stride, size, batch, dev):
    """The same training step as stock PyTorch-ROCm ops (MIOpen convolutions, ATen BatchNorm / loss / autograd) on THIS
    GPU: oracle/ -- the torch restatement of the reference model -- moved to the device.  It is what the reference
    code itself would run on an MI355X.  Optional (--torch-baseline): MIOpen's first-step kernel search takes ~1 min."""
    import torch
    from oracle import loss as oloss
    from oracle.deeplab import OracleDeepLab
    from oracle.optim import OracleSGD
    from oracle.synth import ArchCfg, synth_images, synth_labels, synth_state_dict
    cfg = ArchCfg("deeplabv3plus", model_name, 2, output_stride)
    sd = {k: v.to(dev) for k, v in synth_state_dict(cfg).items()}
    o = OracleDeepLab(cfg, sd, dropout_p=0.1).train()
    opt = OracleSGD(o.parameters())
    x = synth_images(batch, size, size, seed=0).to(dev)
    lab = synth_labels(batch, size, size, seed=0).to(dev)
    w = torch.tensor([1.0, 3.0], device=dev)

    def step():
        loss = oloss.weighted_ce(o(x), lab, w)
        o.zero_grad()
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    print("[bench] torch_rocm_baseline: first step (MIOpen search) %.1f s" % first, file=sys.stderr, flush=True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return {"value": round(batch / dt, 3), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 3),
            "first_step_s": round(first, 1),
            "note": "identical step on stock torch %s ops (MIOpen convs, NCHW fp32) on the same GPU: oracle/ on cuda" %
                    torch.__version__}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch one rank per GPU with torch.distributed.run)" %
                         (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # rehearsal knobs (single-GPU box): ISWM_DIST_BACKEND=gloo runs the N > 1 path with every rank on GPU
    # $ISWM_FORCE_DEVICE; the driver's real runs use RCCL ("nccl") with one GPU per rank
    backend = os.environ.get("ISWM_DIST_BACKEND", "nccl")
    if "ISWM_FORCE_DEVICE" in os.environ:
        local = int(os.environ["ISWM_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    from iswm_amd import _lib, ops
    from iswm_amd.network import modeling
    from iswm_amd.optim import FusedSGD
    from iswm_amd.parallel import DistributedDataParallelHIP
    from iswm_amd.utils.loss import CrossEntropyLoss
    lib = _lib.load()
    if args.conv_math is not None:
        lib.iswm_set_conv_math({"f32": 0, "bf16x6": 1, "bf16": 2}[args.conv_math])
    math_name = {0: "f32", 1: "bf16x6", 2: "bf16"}[lib.iswm_get_conv_math()]

    torch.manual_seed(1)                                   # --random_seed 1, train.py:322
    ctor = {"resnet50": modeling.deeplabv3plus_resnet50, "resnet101": modeling.deeplabv3plus_resnet101}[args.model]
    model = ctor(num_classes=2, output_stride=args.output_stride).to(dev).train()
    opt = FusedSGD(model.parameters(), momentum=0.9, weight_decay=1e-4, nesterov=True)   # train.py:426-432
    group = None
    net = model
    if world > 1:
        group = dist.group.WORLD
        net = DistributedDataParallelHIP(model, process_group=group, time_waits=True)
        net.attach(opt)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=30000, eta_min=0.01 * 0.01)   # train.py:446-452
    crit = CrossEntropyLoss(weight=torch.tensor([1.0, 3.0]), ignore_index=255, group=group).to(dev)

    g = torch.Generator(device="cpu").manual_seed(1000 + rank)
    B, S = args.batch, args.size
    images = torch.randn(B, 3, S, S, generator=g).to(dev)
    labels = (torch.rand(B, S, S, generator=g) < 0.10).to(torch.int64).to(dev)

    def step():
        logits = net(images)
        loss = crit(logits, labels)
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        return loss

    # warm-up; its last step is profiled per conv kernel to find the dominant one, so that the timed region
    # only brackets THAT kernel's launches with events (bracketing all ~350 conv launches costs ~3 % of a step)
    kprof, warm_summary, head_fwd = None, None, None
    head = model.classifier
    head_fwd_orig = head.fwd

    def head_fwd_scoped(*a, **k):          # stamps the ASPP + decoder forward convs of the profiled warm-up step
        if ops.KPROF is not None:
            ops.KPROF.scope = "head_fwd"
        try:
            return head_fwd_orig(*a, **k)
        finally:
            if ops.KPROF is not None:
                ops.KPROF.scope = None
    head.fwd = head_fwd_scoped
    for i in range(args.warmup):
        if i == args.warmup - 1 and not args.no_kernel_timing:
            torch.cuda.synchronize()
            ops.KPROF = ops.KernelProfile()
        step()
    if ops.KPROF is not None:
        torch.cuda.synchronize()
        warm_summary = ops.KPROF.summary()
        head_fwd = ops.KPROF.scope_total("head_fwd")
        ops.KPROF = None
        dominant = max(warm_summary.items(), key=lambda kv: kv[1]["ms"])[0]
        kprof = ops.KernelProfile(only=dominant)
    elif not args.no_kernel_timing:
        kprof = ops.KernelProfile()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.KPROF = kprof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_enqueued = time.perf_counter() - t0          # host time to enqueue the work (GPU still running)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.KPROF = None
    comm = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
        # what the first real multi-GPU run should tell about itself: volume, launches, the time the compute stream stood
        # still waiting for the last buckets (exposed = not hidden behind backward), and that every rank was there
        st = net.comm_stats()
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)
        nsteps = max(1, st["steps"])
        comm = {"backend": st["backend"], "ranks_seen": int(seen.item()), "buckets": st["buckets"],
                "allreduces_per_step": round(st["allreduces"] / nsteps, 2),
                "bytes_allreduced_per_step": int(st["bytes_allreduced"] / nsteps),
                "exposed_wait_ms_per_step": round(st["exposed_wait_ms"] / nsteps, 3),
                "note": "rank 0's view; exposed_wait = compute stream stalled in finish_grad_sync (includes warm-up steps)"}
    final_loss = float(loss.detach())
    # host cost of enqueuing ONE step into an idle queue (the timed loop's enqueue time is back-pressure: the host runs
    # into the launch queue limit and then advances at the GPU's pace)
    torch.cuda.synchronize()
    th = time.perf_counter()
    step()
    host_idle_ms = (time.perf_counter() - th) * 1e3
    torch.cuda.synchronize()

    if rank == 0:
        ms = dt / args.steps * 1e3
        out = {
            "metric": "training images/sec (%dx%d) DeepLabV3+-%s" % (S, S, {"resnet101": "ResNet101", "resnet50": "ResNet50"}[args.model]),
            "value": round(B * world * args.steps / dt, 3),
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # arithmetic type of the path: fp32 tensors and fp32-grade conv products by default; "bf16" only when the
            # mixed-precision conv math was asked for (--conv-math bf16: bf16-rounded MFMA inputs, fp32 accumulate/storage)
            "dtype": "bf16" if math_name == "bf16" else "f32",
            "data": "synthetic",
            "config": {
                "workload": "deeplabv3plus_%s output_stride=%d, %dx%d synthetic tiles, %d images/GPU "
                            "(%s), weighted CE [1,3], "
                            "SGD-nesterov + cosine LR; fp32 tensors, conv products via %s" %
                            (args.model, args.output_stride, S, S, B,
                             "BASELINE.json configs[2]: global batch 128 over 8 GPUs" if args.model == "resnet101"
                             else "BASELINE.json configs[1]",
                             {"bf16x6": "six bf16 MFMAs on an exact 3-way operand split (bf16x6, fp32-level error)",
                              "f32": "v_mfma_f32_32x32x2_f32",
                              "bf16": "ONE bf16 MFMA on bf16-stored activations (mixed precision, NOT fp32-grade)"}[math_name]),
                "conv_math": math_name,
                "global_batch": B * world,
                "parallelism": "dp%d" % world,
            },
            "final_loss": round(final_loss, 6),
            "host_enqueue_ms_idle": round(host_idle_ms, 3),
            "host_enqueue_ms_backpressured": round(t_enqueued / args.steps * 1e3, 3),
        }
        if comm is not None:
            out["comm"] = comm
        if kprof is not None:
            summ = kprof.summary()
            dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
            name, d = dom
            ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
            is_x6 = math_name == "bf16x6" and ("x6" in name or ", true, 3>" in name or name.startswith(("k_conv_pl", "k_wgrad_pl")))
            is_b16 = math_name == "bf16" and ("x6" in name or ", true, 1>" in name or name.startswith(("k_conv_pl", "k_wgrad_pl")))
            peak = BF16_MFMA_PEAK_TFLOPS if is_b16 else (X6_PEAK_TFLOPS if is_x6 else FP32_MFMA_PEAK_TFLOPS)
            out["roofline"] = {
                "kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1),
                "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "peak_basis": ("bf16 dense MFMA peak 2500 TFLOP/s" if is_b16 else
                               "bf16 dense MFMA peak 2500 TFLOP/s / 6 MFMAs per fp32 MAC (bf16x6)" if is_x6
                               else "fp32 MFMA dense peak"),
                "frac_of_fp32_mfma_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": pmc_traffic(name)[0],
                "launches": d["launches"], "avg_us": round(d["ms"] * 1e3 / d["launches"], 2),
                "flops_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
            }
            stale = pmc_traffic(name)[1]
            if stale is not None:
                out["roofline"]["traffic_profiled_build"] = stale      # measured on other kernel sources: not `traffic`
            if warm_summary is not None:
                out["roofline"]["by_kernel_warmup_step"] = {
                    k: {"ms_per_step": round(v["ms"], 3), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                        "launches_per_step": v["launches"]} for k, v in sorted(warm_summary.items())}
                out["roofline"]["conv_ms_per_step_warmup"] = round(sum(v["ms"] for v in warm_summary.values()), 3)
            if head_fwd is not None and head_fwd["ms"] > 0:
                # north_star target: MFMA-roofline utilisation of the ASPP + decoder FORWARD convolutions
                tf = head_fwd["flops"] / (head_fwd["ms"] * 1e-3) / 1e12
                out["roofline"]["aspp_decoder_forward"] = {
                    "launches": head_fwd["launches"], "ms": round(head_fwd["ms"], 3), "tflops": round(tf, 2),
                    "frac_of_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
                    "frac_of_bf16x6_peak": round(tf / X6_PEAK_TFLOPS, 4) if math_name == "bf16x6" else None}
        if world == 1 and math_name == "bf16x6" and not args.no_alt:
            # same step with the exact-fp32 MFMA kernels, for reference (outside the timed region above)
            lib.iswm_set_conv_math(0)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            dt1 = (time.perf_counter() - t1) / 3
            lib.iswm_set_conv_math(1)
            out["alt_conv_math_f32"] = {"value": round(B / dt1, 3), "unit": "images/sec",
                                        "ms_per_step": round(dt1 * 1e3, 3),
                                        "note": "identical step on v_mfma_f32_32x32x2_f32 kernels (ISWM_CONV_MATH=f32)"}
            # and with the mixed-precision conv math of BASELINE configs[4] (reduced precision: reported, never `value`)
            lib.iswm_set_conv_math(2)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t1) / 3
            lib.iswm_set_conv_math(1)
            out["alt_conv_math_bf16"] = {"value": round(B / dt2, 3), "unit": "images/sec",
                                         "ms_per_step": round(dt2 * 1e3, 3),
                                         "note": "identical step in mixed precision (ISWM_CONV_MATH=bf16, BASELINE configs[4]): "
                                                 "activations between convolutions stored as ONE bf16 plane, one bf16 MFMA per "
                                                 "product, fp32 accumulation / BatchNorm statistics / master weights / loss.  "
                                                 "Reduced precision: NOT the headline arithmetic; parity unpinned (the reference "
                                                 "has no such mode; checked against a same-rounding oracle)"}
        if world == 1 and args.torch_baseline:
            del model, net, opt, images, labels          # free the product's activations / arenas first
            torch.cuda.empty_cache()
            tb = torch_rocm_baseline(args.model, args.output_stride, S, B, dev)
            tb["speedup_of_this_path"] = round(out["value"] / tb["value"], 3)
            out["torch_rocm_baseline"] = tb
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model, args.output_stride, S)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
