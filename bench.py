#!/usr/bin/env python3
"""Headline benchmark: images/sec of the CIM per-image training step (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one synthetic image per GPU through the whole hot path: backbone fwd -> fused ROIAlign+mask-cat ->
MaskFuse -> 8 heads -> 3 x CIM mining incl. anti-noise sampling (on the device) -> 4 losses -> backward ->
gradient all-reduce (RCCL) -> SGD step.  Workload at N=1 = BASELINE configs[1] (resnet50_voc, bs=1,
~1000 proposals per image).  The timed loop cycles through --images (default 8) DISTINCT synthetic images per rank:
800 ... 1200 proposals (mean 1000), three of the training scales of configs/resnet50_voc.yaml:34 (688 = the median,
576, 864), different class sets, pseudo-GT counts and PRM clusters - nothing is cached per image.
Weak scaling: one image per rank.  Rank 0 prints ONE JSON line (contract in the task statement) with `roofline`,
`roofline_hbm`, `cpu_baseline` and `extra` (sustained >= 5 s loop, iter_size = 4, per-step host->device upload,
and for N > 1: all-reduce time and overlap).
"""
import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TF = 2516.6   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (256 CU x 4096 flop/clk x 2.4 GHz)

# the 8 images of a cycle: (proposals, longest side); mean N = 1000, mean pixel count 1.07 x the 688 scale
IMAGE_MIX = [(1000, 688), (800, 576), (1200, 864), (900, 688), (1100, 688), (1000, 864), (850, 576), (1150, 688)]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="resnet50_voc")
    ap.add_argument("--images", type=int, default=8, help="distinct synthetic images per rank the loop cycles through")
    ap.add_argument("--fixed-image", action="store_true", help="round-1 workload: ONE image (N = config default) repeated")
    ap.add_argument("--iter-size", type=int, default=1, help="images per optimizer step (tools/train.py --iter_size)")
    ap.add_argument("--sustained", type=float, default=5.0, help="seconds of the extra sustained loop (0 = skip)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra loops (sustained, iter_size=4, upload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap-update", action="store_true", help="A/B aid: the whole SGD update on the step's own stream (round 5)")
    ap.add_argument("--trail-wgs", type=int, default=None, help="A/B aid: workgroups of the side-stream update launch")
    ap.add_argument("--dw-form", type=int, default=None, help="A/B aid: kernel form of MaskFuse's late weight-gradient products "
                    "(cim_amd/ops/maskfuse_pair.py: DW_FORM; 0 = 256 x 256 tiles that own their CU, 1 = the co-resident 128 x 256 form)")
    ap.add_argument("--no-postpone-dw", action="store_true", help="A/B aid: MaskFuse's late weight gradients are launched at the end of its "
                    "backward node instead of behind the ROIAlign backward (cim_amd/ops/gemm.py: POSTPONE_DW)")
    ap.add_argument("--dw-wgs", type=int, default=None, help="A/B aid: workgroups per launch of those products (DW_WGS / DW_FORM1_WGS)")
    ap.add_argument("--late-cus", type=int, default=None, help="A/B aid: CUs of the stream MaskFuse's late weight-gradient launches run on "
                    "(cim_amd/ops/gemm.py: LATE_CUS; 0 = the whole chip); default: the package's setting")
    ap.add_argument("--phases", type=int, default=0, help="extra: N more steps with HIP events at the phase boundaries of the "
                    "step (main stream, no profiler attached) -> extra.phases")
    ap.add_argument("--cpu-sample", type=int, default=0, help="proposals in the CPU sample (0 = all)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a functional "
                                                       "multi-rank check on a box with fewer GPUs than ranks)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as CHILD processes (torch.distributed.run)
    before this process has touched the GPU, relay rank 0's JSON line, exit with the children's code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    sys.exit(proc.returncode if proc.returncode else (0 if line is not None else 1))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    os.environ.setdefault("TORCH_NCCL_ENABLE_TIMING", "1")        # Work._get_duration() for allreduce_ms
    run(args)


# ------------------------------------------------------------------------------------------------ the measurement
def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    from cim_amd import _lib, mask_iou, synthetic
    from cim_amd.core.config import cfg
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling import heads
    from cim_amd.modeling.model_builder import Generalized_RCNN
    from cim_amd.nn import DataParallel
    from cim_amd.ops import gemm as gemm_mod

    # The loop below draws nothing from np.random between its steps (inputs are resident, no sampler): the generator is settled
    # right before the next step's draw instead of at the end of backward (cim_amd/modeling/heads.py: LAZY_SETTLE) - the same
    # stream position, but the host's wait for the step's mining launches no longer caps its lead over the GPU at half a step
    # (with an 8 ms host hiccup every fourth step: 14.19 vs 14.46 ms per step; none on a quiet box: 13.95 vs 13.95).
    heads.LAZY_SETTLE = True
    if args.late_cus is not None:
        gemm_mod.LATE_CUS = args.late_cus
    if args.no_postpone_dw:
        gemm_mod.POSTPONE_DW = False
    if args.dw_form is not None or args.dw_wgs is not None:
        from cim_amd.ops import maskfuse_pair as _mp
        if args.dw_form is not None:
            _mp.DW_FORM = args.dw_form
        if args.dw_wgs is not None:
            if _mp.DW_FORM == 1:
                _mp.DW_FORM1_WGS = args.dw_wgs
            else:
                _mp.DW_WGS = args.dw_wgs
    if os.environ.get("CIM_BENCH_WATCHDOG"):        # debugging aid: dump all thread stacks if a phase stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CIM_BENCH_WATCHDOG"]), repeat=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    if args.backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)      # functional check: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    _lib.load()                                                # no HIP extension -> fail loudly

    apply_preset(args.config)
    torch.manual_seed(cfg.RNG_SEED)                            # identical initial weights on every rank
    model = Generalized_RCNN()
    init_for_synthetic(model)
    model = model.to(dev).train()
    timer = KernelTimer(torch, np)
    instrument(_lib, timer)
    dp = DataParallel(model, cpu_keywords=["im_info", "roidb"], minibatch=True)
    opt = make_optimizer(model, torch)
    # opt-in of the headline loop (extra.settings): the update of the three big MaskFuse weights (96 % of the update's 5.1 GB) runs
    # on the side stream UNDER the next step's backbone forward instead of in front of it (cim_amd/optim/sgd.py: overlap_update)
    opt.overlap_update = not args.no_overlap_update
    if args.trail_wgs is not None:
        opt.trail_workgroups = args.trail_wgs
    if os.environ.get("CIM_EARLY_STEP", "0") == "1":
        dp.attach_optimizer(opt)     # opt-in: MaskFuse / heads update overlapped with the ROIAlign + backbone backward
                                     # (measured at 1 GPU: 17.36-17.47 ms with, 17.37 without - HBM time only moves)
    Cf = model.Conv_Body.dim_out

    # ---- the images of the cycle: host (pinned) copies + device-resident copies
    base_n = synthetic.CONFIGS[args.config]["n"]
    base_t = synthetic.CONFIGS[args.config]["target"]
    mix = [(base_n, base_t)] if args.fixed_image else \
        [(int(round(n * base_n / 1000.0)), int(round(t * base_t / 688.0))) for n, t in (IMAGE_MIX * ((args.images + 7) // 8))[:args.images]]
    host_batches, dev_batches, infos = [], [], []
    map_build = []
    for j, (n, target) in enumerate(mix):
        inp = synthetic.make_image_inputs(args.config, seed=cfg.RNG_SEED + 1000 * rank + j, n=n, target=target)
        full = torch.from_numpy(inp["full_masks"]).to(dev)
        if j == 0:
            mask_iou.mask_iou_maps(full)                       # warm-up of the one-off map construction (a-7 / f-1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        iou, asy = mask_iou.mask_iou_maps(full)
        e1.record()
        torch.cuda.synchronize()
        hw = int(full[0].numel())
        map_build.append(dict(ms=e0.elapsed_time(e1), n=n, hw=hw, words=(hw + 63) // 64))
        del full
        t = lambda a: torch.from_numpy(a).unsqueeze(0)
        hb = dict(data=torch.from_numpy(inp["data"]), rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]),
                  mat=t(inp["mat"]), index=t(inp["index"]), iou_map=iou.cpu(), asy_iou_map=asy.cpu())
        hb = {k: v.pin_memory() for k, v in hb.items()}
        host_batches.append(hb)
        dev_batches.append({k: v.to(dev) for k, v in hb.items()})
        H, W = inp["image_hw"]
        infos.append(dict(n=n, H=H, W=W, n_cls=int((inp["labels"] > 0).sum())))
        del inp
    torch.cuda.synchronize()
    if args.late_cus:
        # hipExtStreamCreateWithCUMask only makes BLOCKING streams (they synchronise with the NULL stream, which is torch's default
        # stream): the step then runs on a stream of its own
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    np.random.seed(cfg.RNG_SEED + rank)                        # the anti-noise sampling stream
    state = dict(i=0, feat={})

    def one_image(upload):
        j = state["i"] % len(dev_batches)
        state["i"] += 1
        if upload:      # what nn.DataParallel does with the loader's CPU tensors (lib/nn/parallel/_functions.py:70-82)
            b = {k: v.to(dev, non_blocking=True) for k, v in host_batches[j].items()}
        else:
            b = dev_batches[j]
        timer.image = j
        out = dp(**{k: [v] for k, v in b.items()}, gtrois=[None])
        state["feat"][j] = tuple(out["blob_conv"].shape[-2:])
        # (tools/train.py:435 differentiates `total_loss`: the reference's training_stats adds the four losses up on the host side of
        # the loop; the model here returns that sum itself, from the loss launch)
        loss = out["total_loss"] if "total_loss" in out else sum(v.sum() for v in out["losses"].values())
        loss.backward()          # gradient all-reduce + NumPy-generator settle happen inside (end-of-backward callbacks)
        return loss

    def step(iter_size=1, upload=False):
        dp.zero_grad()
        for _ in range(iter_size):
            loss = one_image(upload)
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_steps, **kw):
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            loss = step(**kw)
        fence()
        el = time.perf_counter() - t0
        if world > 1:
            mine = torch.tensor([el], device=dev, dtype=torch.float64)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)                 # each rank's own clock; the line reports the MAX (the contract)
            state["rank_seconds"] = [float(t) for t in every]
            el = max(state["rank_seconds"])
        assert torch.isfinite(loss), "non-finite loss in the timed region"
        return el

    dp.iter_size = args.iter_size
    if os.environ.get("CIM_BENCH_PER_STEP") == "1" and rank == 0:      # debugging aid: synchronized time of every step from the first
        for i in range(args.warmup + args.steps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(args.iter_size)
            t2 = time.perf_counter()                    # the host has enqueued the whole step (the GPU was idle at t1)
            torch.cuda.synchronize()
            print("step %d: %.2f ms (host returned after %.2f ms)" % (i, 1e3 * (time.perf_counter() - t1), 1e3 * (t2 - t1)), file=sys.stderr)
    # one-off set-up, untimed and not counted as warm-up: every distinct image of the cycle once (allocator pools and
    # per-shape kernel attributes for its shapes), so that a short --warmup never leaves a first occurrence in the timed region
    # (two passes: in the second one the caching allocator has blocks for every size class of the cycle - a first occurrence of
    # an image still grows pools, measured as isolated 20+ ms steps in the first cycle after one pass)
    # (more set-up passes do not steady the short timed region: two passes before a warm-up of 8: 14.36 / 16.01 / 14.40 ms against
    # 15.60 / 15.06 / 14.38 without, one shared box, interleaved - what moves it there is the host: a step is enqueued in ~6.4 ms, and
    # on a loaded box single steps take the host 11-16 ms (CIM_BENCH_PER_STEP=1 prints both times); extra.sustained is the steadier figure)
    for _ in range(2 * len(dev_batches) if args.warmup < len(dev_batches) else 0):
        step(args.iter_size)
    for _ in range(args.warmup):
        step(args.iter_size)
    # Python's cyclic collector: a full (generation-2) collection walks every tracked object of the process - ~270 k here (torch,
    # the modules, the synthetic inputs) - and took 22-88 ms when it fell into the timed region (one such pause = +3.7 ms per
    # step on a 24-step region: 17.4 instead of 14.4 ms in one refresh of the round).  gc.freeze() moves what exists NOW to the
    # permanent generation: the per-step garbage (autograd contexts, closures) is still collected, a full collection only
    # walks what was created after this point.  A training loop does the same once after its first iterations.
    import gc
    gc.collect()
    gc.freeze()
    timer.enabled = True
    if world > 1:
        dp.comm_works = []
    elapsed = timed(args.steps, iter_size=args.iter_size)          # ---- THE timed region: exactly --steps steps
    if os.environ.get("CIM_BENCH_WINDOWS") == "1" and rank == 0:    # debugging aid: further windows of the same length (is the first one special?)
        timer.enabled = False
        print("window 0: %.3f ms" % (1e3 * elapsed / args.steps), file=sys.stderr)
        for w in range(1, 6):
            print("window %d: %.3f ms" % (w, 1e3 * timed(args.steps, iter_size=args.iter_size) / args.steps), file=sys.stderr)
    timer.enabled = False
    rank_seconds = state.get("rank_seconds", [elapsed])
    comm_works = getattr(dp, "comm_works", None)
    dp.comm_works = None
    images = world * args.steps * args.iter_size

    extra = {}
    if not args.no_extra:
        if args.sustained > 0:                                       # same loop for >= --sustained seconds (clocks settle)
            per = elapsed / args.steps
            k = max(args.steps, int(np.ceil(args.sustained / per)))
            el = timed(k, iter_size=args.iter_size)
            extra["sustained"] = dict(seconds=el, steps=k, images_per_s=world * k * args.iter_size / el,
                                      ms_per_step=1e3 * el / k)
        k4 = max(4, args.steps // 4)
        dp.iter_size = 4
        step(4)
        el = timed(k4, iter_size=4)                                  # the reference's operating point (scripts/train_CIM.sh:8)
        extra["iter_size_4"] = dict(images_per_s=world * k4 * 4 / el, ms_per_image=1e3 * el / (k4 * 4), optimizer_steps=k4)
        dp.iter_size = args.iter_size
        step(args.iter_size, upload=True)
        el = timed(args.steps, iter_size=args.iter_size, upload=True)   # inputs (image, rois, masks, labels, mat, both maps) from pinned host memory every step
        up_bytes = float(np.mean([sum(v.numel() * v.element_size() for v in hb.values()) for hb in host_batches]))
        extra["with_h2d_upload"] = dict(images_per_s=images / el, ms_per_step=1e3 * el / args.steps, bytes_per_image=up_bytes)
        if world == 1:
            extra["reference_loop"] = reference_loop(torch, np, heads, dp, opt, dev_batches, timed_fn=timed, state=state, timer=timer,
                                                     steps=args.steps, headline_ms=1e3 * elapsed / args.steps)
            gc.freeze()
            extra["tf32_class"] = tf32_class(torch, np, heads, model, dp, opt, dev_batches, step, timed, timer, infos, Cf, args,
                                             headline_ms=1e3 * elapsed / args.steps)
        if world > 1:                                                # compute-only steps (no collectives) -> what the all-reduce costs
            with dp.no_sync():
                step(args.iter_size)
                el_ns = timed(args.steps, iter_size=args.iter_size)
            ar_ms = None
            if comm_works:
                try:
                    ar_ms = float(sum(w._get_duration() for w in comm_works)) / args.steps
                except Exception as e:          # timing not available in this build of ProcessGroupNCCL
                    extra["allreduce_ms_error"] = "%s: %s" % (type(e).__name__, e)
            exposed = 1e3 * (elapsed - el_ns) / args.steps
            grad_bytes = 4.0 * sum(p.numel() for p in model.parameters() if p.requires_grad)
            extra["comm"] = dict(allreduce_ms=ar_ms, exposed_ms=exposed, ms_per_step_no_sync=1e3 * el_ns / args.steps,
                                 overlap_frac=(1.0 - exposed / ar_ms) if ar_ms else None, gradient_bytes=grad_bytes,
                                 ring_bound_ms=1e3 * 2.0 * (world - 1) / world * grad_bytes / 153e9,
                                 buckets=len(dp.buckets))
    # which backward schedule of the fused MaskFuse node ran (single-process: chunked late launches behind the ROIAlign backward;
    # several ranks: whole products published to nn.DataParallel at once) - counts over the whole run
    from cim_amd.ops import maskfuse_pair as _mfp
    extra.setdefault("comm", {})["maskfuse_backward_schedule"] = dict(_mfp.SCHEDULE)
    if args.phases > 0:
        extra["phases"] = phase_times(torch, model, opt, step, args.phases, args.iter_size)
    heads.settle_rng()
    # the measured run must have been a VALID training run to the end: every parameter finite (a finite last loss alone does not
    # show a weight gradient that went NaN a few steps ago - and NaN operands run the MFMAs faster)
    with torch.no_grad():
        worst = torch.stack([p.detach().abs().max() for p in model.parameters()]).max()
    assert torch.isfinite(worst), "non-finite parameters after the timed region: the measurement is invalid"
    # a-7 / f-1: the N x N mask-IoU and containment maps of every image of the cycle, built on the device from the full-
    # resolution proposal masks OUTSIDE the timed step (the reference builds them offline: tools/pre/create_cob_iou.py).
    # Integer work: N^2/2 * ceil(HW/64) 64-bit and + popcount pairs; bytes: N*HW mask bytes in, 4 N^2 out.
    mb_ms = float(np.mean([m["ms"] for m in map_build]))
    extra["mask_iou_build"] = dict(
        ms_per_image=mb_ms,
        and_popcount_Gops=float(np.sum([m["n"] ** 2 / 2.0 * m["words"] for m in map_build]) / np.sum([m["ms"] for m in map_build]) / 1e6),
        int_valu_peak_Gops=256 * 64 * 2.4 / 4.0 * 1.0,          # 256 CUs x 64 lanes x 2.4 GHz / (2 v_and_b32 + 2 v_bcnt per 64-bit pair)
        mask_GBs=float(np.sum([m["n"] * m["hw"] for m in map_build]) / np.sum([m["ms"] for m in map_build]) / 1e6),
        note="pack (16 px per lane) + areas + symmetric popcount pair kernel, bit-identical fp16 maps; outside the timed step")

    if rank == 0:
        line = report(args, world, elapsed, images, timer, infos, state["feat"], Cf, cfg, gemm_mod, np)
        from cim_amd.ops import fallback as _fb
        extra["aten_fallbacks"] = {"%s: %s" % k: v for k, v in _fb.counts().items()}      # GPU tensors that took a library branch (none expected)
        extra["settings"] = dict(lazy_settle=bool(heads.LAZY_SETTLE), gc_freeze_after_warmup=True, differentiates="model's total_loss key",
                                 retain_graph=False, optimizer_overlap_update=bool(opt.overlap_update),
                                 note="the headline loop's four departures from the reference's literal driver loop (the fourth: "
                                 "cim_amd.optim.SGD.overlap_update - the big weights' update runs on the side stream under the next "
                                 "backbone forward; same arithmetic, same weights bit for bit); extra.reference_loop measures that loop")
        line["extra"] = extra
        # what the process group itself reports (a SCALE run is checkable: ranks, backend, every rank's own rate)
        line["dist"] = dict(world_size=dist.get_world_size() if world > 1 else 1, backend=(dist.get_backend() if world > 1 else None),
                            devices=torch.cuda.device_count(),
                            per_rank_images_per_s=[args.steps * args.iter_size / t for t in rank_seconds])
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_sample)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def reference_loop(torch, np, heads, dp, opt, dev_batches, timed_fn, state, timer, steps, headline_ms):
    """The reference's LITERAL driver loop around the model (tools/train.py:418-438 with lib/utils/training_stats.py:65-118 between
    forward and backward), as a user of the unchanged train.py gets it: per image the four losses are reduced with .mean(0), summed
    into `total_loss` by ATen adds, each read back with .cpu() (iter_size 1: five blocking reads between forward and backward;
    iter_size 4, the shipped scripts/train_CIM.sh: five reads per four images), backward(retain_graph=True); the product's default
    generator settle (end of backward), no gc.freeze(), the whole optimizer update on the loop's stream.  The headline loop differs in
    exactly these points (extra.settings)."""
    import gc
    lazy = heads.LAZY_SETTLE
    heads.LAZY_SETTLE = False
    overlap, opt.overlap_update = opt.overlap_update, False          # (the whole update on the loop's own stream, as torch.optim.SGD)
    gc.unfreeze()
    stats = {"inner": {}, "log": []}

    def update_iter_stats(out, inner, iter_size):
        total = 0
        for k, loss in out["losses"].items():
            assert loss.shape[0] == 1                                  # cfg.NUM_GPUS
            loss = loss.mean(dim=0, keepdim=True)
            total = total + loss
            out["losses"][k] = loss
            if iter_size == 1:
                stats["log"].append(loss.data[0].cpu())
            else:
                stats["inner"].setdefault(k, []).append(loss.data[0])
                if inner == iter_size - 1:
                    stats["log"].append((sum(stats["inner"].pop(k)) / iter_size).cpu())
        out["total_loss"] = total
        if iter_size == 1:
            stats["log"].append(total.data[0].cpu())
        else:
            stats["inner"].setdefault("total", []).append(total.data[0])
            if inner == iter_size - 1:
                stats["log"].append((sum(stats["inner"].pop("total")) / iter_size).cpu())
        del stats["log"][:-16]

    def ref_step(iter_size=1, upload=False):
        opt.zero_grad()
        for inner in range(iter_size):
            j = state["i"] % len(dev_batches)
            state["i"] += 1
            timer.image = j
            out = dp(**{k: [v] for k, v in dev_batches[j].items()}, gtrois=[None])
            update_iter_stats(out, inner, iter_size)
            loss = out["total_loss"]
            loss.backward(retain_graph=True)
        opt.step()
        return loss.reshape(())

    res = {}
    try:
        for it in (1, 4):
            dp.iter_size = it
            for _ in range(3):
                ref_step(it)
            k = max(4, steps // it)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                loss = ref_step(it)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            assert torch.isfinite(loss)
            res["iter_size_%d" % it] = dict(images_per_s=k * it / el, ms_per_image=1e3 * el / (k * it), optimizer_steps=k)
    finally:
        heads.LAZY_SETTLE = lazy
        opt.overlap_update = overlap
        dp.iter_size = 1
    res["vs_headline"] = res["iter_size_1"]["ms_per_image"] / headline_ms
    res["note"] = ("tools/train.py:418-438 + training_stats.UpdateIterStats restated: .mean(0) / ATen total / .cpu() reads between "
                   "forward and backward, backward(retain_graph=True), generator settled at the end of backward, garbage collector not frozen")
    return res


def tf32_class(torch, np, heads, model, dp, opt, dev_batches, step, timed, timer, infos, Cf, args, headline_ms):
    """"decide explicitly and report both".  The same loop with ONE fp16 MFMA product per multiply-add in the MaskFuse
    contractions (ops.pair.PRODUCTS = 1: 11-bit operands, fp32 accumulation - TF32-class, the reference's own arithmetic on its
    hardware) instead of the headline's three (fp32-class).  Also: how far one step's losses / gradients move, and whether the mined
    pseudo labels stay the same."""
    from cim_amd.ops import pair
    res = {}
    # ---- deviation of one step (image 0, same weights, same generator state), three products against one: with the labels each run
    # mines itself (a flipped pseudo label moves a loss by percents: that figure swings from box to box) AND on the three-product run's
    # labels (model._fixed_mining: the arithmetic alone - reproducible)
    b = dev_batches[0]
    overlap, opt.overlap_update = opt.overlap_update, False

    class _Fixed:
        def __init__(self, m):
            self.pseudo, self.valid, self.status = m.pseudo, m.valid, m.status

        def commit(self):
            pass

    def one(products, fixed=None):
        pair.PRODUCTS = products
        model.__dict__["_fixed_mining"] = fixed
        try:
            dp.zero_grad()
            np.random.seed(12345)
            out = dp(**{k: [v] for k, v in b.items()}, gtrois=[None])
            out["total_loss"].backward()
            heads.settle_rng()
            torch.cuda.synchronize()
            mined = model.__dict__["_last_mining"]
            return ({k: float(v) for k, v in out["losses"].items()},
                    {n: p.grad.detach().double().clone() for n, p in model.named_parameters() if p.grad is not None},
                    [tuple(x.clone() for x in ps) for ps in mined.pseudo], mined.valid.clone(), mined)
        finally:
            pair.PRODUCTS = 3
            model.__dict__.pop("_fixed_mining", None)

    def deviation(l3, g3, l1, g1):
        worst = max(((float((g1[n] - g3[n]).norm() / g3[n].norm()), n) for n in g3 if float(g3[n].norm()) > 1e-6 * g3[n].numel() ** 0.5),
                    default=(0.0, ""))
        return dict(loss_rel=max(abs(l1[k] - l3[k]) / max(abs(l3[k]), 1e-12) for k in l3), worst_gradient_rel=worst[0], worst_parameter=worst[1])

    try:
        l3, g3, p3, v3, m3 = one(3)
        l1, g1, p1, v1, _ = one(1)
        lf, gf, _, _, _ = one(1, _Fixed(m3))
    finally:
        opt.overlap_update = overlap
    res["one_step_deviation_on_the_same_pseudo_labels"] = deviation(l3, g3, lf, gf)
    res["one_step_deviation_with_its_own_mining"] = dict(
        deviation(l3, g3, l1, g1),
        pseudo_labels_identical=bool(torch.equal(v3, v1)) and all(torch.equal(x, y) for a, c in zip(p3, p1) for x, y in zip(a, c)))
    # ---- throughput
    try:
        pair.PRODUCTS = 1
        for _ in range(len(dev_batches)):
            step(args.iter_size)
        saved, timer.spans = timer.spans, {}
        timer.enabled = True
        el = timed(args.steps, iter_size=args.iter_size)
        timer.enabled = False
        ls = timer.launches("wino_gemm_fwd")
        timer.spans = saved
    finally:
        pair.PRODUCTS = 3
    res.update(images_per_s=args.steps * args.iter_size / el, ms_per_step=1e3 * el / args.steps, vs_headline_ms=(1e3 * el / args.steps) / headline_ms)
    if ls:
        ms = float(np.sum([m for m, _ in ls]))
        fl = float(np.sum([121 * 2.0 * infos[j]["n"] * (2 * Cf) * Cf for _, j in ls]))
        res["dominant_kernel"] = dict(kernel="gemm_pair_kernel<L_KC,L_KC,ONEP> x121 (MaskFuse conv3x3 fwd, one f16 product)", ms=ms / len(ls),
                                      achieved_tflops=fl / (ms * 1e-3) / 1e12, frac_of_f16_mfma_peak=fl / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TF,
                                      note="the l halves of the pair images are still read (interleaved with the h halves): this launch is HBM-bound")
    res["note"] = ("never the default: the headline is the fp32-class line.  one_step_deviation_on_the_same_pseudo_labels is the arithmetic's "
                   "own deviation (reproducible); with its own mining a single flipped pseudo label dominates.  Against the REFERENCE's cfg1 run "
                   "and the cfg2 CPU oracle step: tests/test_gpu_tolerance.py::test_tf32_class_single_product_deviation -> "
                   "profiles/r6/parity_deviation.json")
    return res


def phase_times(torch, model, opt, step, n, iter_size):
    """Average ms per phase over n steps, from HIP events recorded on the step's stream at the phase boundaries (module hooks;
    the gradient hook of the body's output marks the start of the body's backward).  Unlike a rocprofv3 trace this does not slow
    the host down, so launch-dense phases (the body: ~90 + ~200 launches) show their real wall time."""
    ev = {}
    mark = lambda k: ev.setdefault(k, []).append(_rec(torch))
    hooks = [model.Conv_Body.register_forward_pre_hook(lambda m, i: mark("body0")),
             model.Conv_Body.register_forward_hook(lambda m, i, o: (mark("body1"), o.register_hook(lambda g: mark("bwd_body0")) if o.requires_grad else None) and None),
             model.Box_Head.register_forward_hook(lambda m, i, o: mark("maskfuse1")),
             model.register_forward_hook(lambda m, i, o: mark("fwd1")),
             opt.register_step_pre_hook(lambda o, a, k: mark("opt0"))]
    try:
        for _ in range(n):
            mark("start")
            step(iter_size)
            mark("end")
    finally:
        for h in hooks:
            h.remove()
    torch.cuda.synchronize()
    names = [("zero_grad .. body", "start", "body0"), ("body forward", "body0", "body1"), ("ROIAlign + MaskFuse forward", "body1", "maskfuse1"),
             ("heads + mining + losses", "maskfuse1", "fwd1"), ("backward: losses .. ROIAlign", "fwd1", "bwd_body0"),
             ("backward: body (+ late weight gradients, join)", "bwd_body0", "opt0"), ("optimizer", "opt0", "end")]
    out = {}
    for label, a, b in names:
        if len(ev.get(a, [])) == len(ev.get(b, [])) == n * (iter_size if a not in ("start", "opt0") and b not in ("end", "opt0") else 1):
            out[label] = float(sum(x.elapsed_time(y) for x, y in zip(ev[a], ev[b])) / len(ev[a]))
    out["step"] = float(sum(x.elapsed_time(y) for x, y in zip(ev["start"], ev["end"])) / n)
    return out


def _rec(torch):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


class KernelTimer:
    """HIP events around selected C-ABI launches, recorded on the launch stream inside the timed region
    (torch.cuda.Event records on torch's current stream, which is the stream every launch here uses)."""

    def __init__(self, torch, np):
        self.torch, self.np = torch, np
        self.spans = {}
        self.enabled = False
        self.image = 0

    def span(self, name):
        timer = self

        class _Ctx:
            def __enter__(self):
                if timer.enabled:
                    self.a = timer.torch.cuda.Event(enable_timing=True)
                    self.b = timer.torch.cuda.Event(enable_timing=True)
                    self.a.record()

            def __exit__(self, *exc):
                if timer.enabled:
                    self.b.record()
                    timer.spans.setdefault(name, []).append((self.a, self.b, timer.image))
        return _Ctx()

    def launches(self, name):
        """[(ms, image index)] of every timed launch."""
        return [(a.elapsed_time(b), j) for a, b, j in self.spans.get(name, [])]


MINING_CALLS = ("cim_asy_prep", "cim_mining_step")


def instrument(_lib, timer):
    orig_call = _lib.call
    state = {"conv": 0, "bg": 0}

    def call(name, *args):
        if name in ("cim_roi_align_maskcat_fwd", "cim_roi_align_maskcat_fwd_ws", "cim_roi_align_maskcat_bwd",
                    "cim_roi_align_maskcat_bwd_ws", "cim_roi_align_wino7_pair_fwd", "cim_roi_align_bwd_ws"):
            if "bwd" not in name:
                state["conv"] = state["bg"] = 0
            with timer.span("cim_roi_align_bwd" if name == "cim_roi_align_bwd_ws" else "cim_roi_align_maskcat_bwd" if "bwd" in name else
                            "cim_roi_align_wino7_pair_fwd" if "wino7" in name else "cim_roi_align_maskcat_fwd"):
                return orig_call(name, *args)
        if name == "cim_gemm_pair_batched":  # Winograd-domain GEMMs, per image: forward, data grad, weight grad
            state["bg"] += 1
            with timer.span(("wino_gemm_fwd", "wino_gemm_dgrad", "wino_gemm_wgrad")[min(state["bg"], 3) - 1]):
                return orig_call(name, *args)
        if name in MINING_CALLS:            # a-4 ... a-6: every mining / sampling / assignment launch of the three CIM layers
            with timer.span("mining"):
                return orig_call(name, *args)
        return orig_call(name, *args)

    _lib.call = call          # every wrapper resolves `_lib.call` at call time


def init_for_synthetic(model):
    """Random-init weights of the same architecture (no checkpoints on the box).  The ResNet body
    keeps BatchNorm in eval mode (resnet50.py:63-68), so with untrained running statistics it is an
    identity and the residual sum would grow ~2x in variance per block; damp the last BN scale of
    every bottleneck (as zero-init-residual schemes do) so activations stay O(1) and the
    synthetic training steps stay finite.  Architecture and work per step are unchanged."""
    import torch
    for m in model.modules():
        if hasattr(m, "bn3"):                                   # bottleneck (ResNet-50, HRNet stem / head)
            torch.nn.init.constant_(m.bn3.weight, 0.25)
        elif hasattr(m, "bn2") and hasattr(m, "downsample"):    # HRNet basic block
            torch.nn.init.constant_(m.bn2.weight, 0.25)


def pmc_traffic(config):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs of this
    script, tools/pmc_traffic.py).  Only a profile of THIS workload counts: files are named per config and record the
    image mix they were taken on; anything else -> {} and `traffic: null`."""
    for rnd in ("r6", "r5", "r4"):                       # the newest committed profile of this configuration
        path = os.path.join(REPO, "profiles", rnd, "pmc_traffic_%s.json" % config)
        if os.path.exists(path):
            with open(path) as f:
                out = json.load(f)
            out["_source"] = "profiles/%s/pmc_traffic_%s.json" % (rnd, config)
            return out
    return {}


def make_optimizer(model, torch):
    """Param groups of tools/train.py:282-311 (bias: lr x2, no weight decay), SGD momentum 0.9."""
    bias, nonbias = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (bias if "bias" in name else nonbias).append(p)
    lr, wd = 0.0005, 0.0005                                   # configs/resnet50_voc.yaml SOLVER
    groups = [dict(params=nonbias, lr=lr, weight_decay=wd), dict(params=bias, lr=2 * lr, weight_decay=0.0)]
    from cim_amd.optim import SGD          # fused multi-tensor SGD, one HIP launch per step (cim_amd/csrc/sgd.hip)
    return SGD(groups, lr=lr, momentum=0.9)


def report(args, world, elapsed, images, timer, infos, feat, Cf, cfg, gemm_mod, np):
    engine, conv_algo, products = "f16x2p", "winograd7", 3.0     # the product's ONE engine / algorithm (ops/maskfuse_pair.py)
    # dominant kernel: the MFMA GEMM of the MaskFuse 3x3 conv forward (a-2).  Algorithmic flops of ONE launch on image j: the mixed
    # 4 + 3 Winograd tiling's 121 batched GEMMs [N_j x 2Cf] x [2Cf x Cf]                                          
    wino = timer.launches("wino_gemm_fwd")
    ms = [m for m, _ in wino]
    flops = [2.0 * 121 * infos[j]["n"] * (2 * Cf) * Cf for _, j in wino]
    kname = "gemm_pair_kernel<L_KC,L_KC> x121 (MaskFuse conv3x3 fwd, Winograd 4+3 mixed tiling domain)"
    # every algorithmic fp32 multiply-add is executed as three fp16 MFMA products with fp32 accumulation, so the kernel is priced
    # against the f16 MFMA peak with achieved = 3 x algorithmic flops / time
    conv_ms = float(np.mean(ms))
    alg_tf = float(np.sum(flops)) / (float(np.sum(ms)) * 1e-3) / 1e12
    roofline = dict(bound="mfma", kernel=kname, achieved=products * alg_tf, peak=BF16_MFMA_PEAK_TF, unit="TFLOP/s", traffic=None,
                    ms=conv_ms, launches=len(ms), algorithmic_flops_per_launch=float(np.mean(flops)),
                    engine="f16x2p: 3 f16 MFMA products per fp32 multiply-add on operands pre-split by their producers (two fp16 terms, "
                           "one power-of-two scale per matrix), LDS-DMA staging, fp32 accumulate",
                    algorithmic_tflops=alg_tf, fp32_mfma_peak=FP32_MFMA_PEAK_TF)
    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    pmc = pmc_traffic(args.config)
    mix_tag = "fixed" if args.fixed_image else "mix%d" % len(infos)
    pmc_ok = pmc.get("_workload") == mix_tag and pmc.get("_conv_algo") == conv_algo and pmc.get("_engine") == engine
    # `traffic` is NOT measured by this run: it is the calibrated FETCH_SIZE + WRITE_SIZE of the same launch from the committed
    # rocprofv3 --pmc passes of this command (two separate profiler runs), replayed here when workload, algorithm and engine match
    if pmc_ok and wino and pmc.get("wino_gemm_fwd"):
        roofline["traffic"] = pmc["wino_gemm_fwd"]["hbm_bytes_mean"]
        roofline["traffic_source"] = pmc["_source"]
    # HBM-bound hand-written kernels: fused ROIAlign+mask-cat fwd / bwd; algorithmic bytes of one launch on image j:
    #   4 (Cf Hf Wf + 5N + 49N) + 4 N 2Cf 49                                                                   
    hbm = []

    def ra_bytes(j):
        Hf, Wf = feat[j]
        n = infos[j]["n"]
        return 4.0 * (Cf * Hf * Wf + 5 * n + 49 * n) + 4.0 * n * 2 * Cf * 49

    def ra_wino_bytes(j):          # the fused forward's own traffic: the map in, the Winograd input pair image out (no `cat`)
        Hf, Wf = feat[j]
        n = infos[j]["n"]
        return 4.0 * (Cf * Hf * Wf + 5 * n + 49 * n) + 4.0 * 121 * ((n + 31) // 32 * 32) * 2 * Cf

    ls = timer.launches("cim_roi_align_wino7_pair_fwd")
    if ls:
        tot_b = float(np.sum([ra_wino_bytes(j) for _, j in ls]))
        tot_8d = float(np.sum([ra_bytes(j) for _, j in ls]))
        tot_ms = float(np.sum([m for m, _ in ls]))
        ach = tot_b / (tot_ms * 1e-3) / 1e9
        hbm.append(dict(kernel="cim_roi_align_wino7_pair_fwd (ROIAlign + mask multiply + concat + Winograd 4+3 input transform: the conv "
                               "input `cat` is never stored; replaces cim_roi_align_maskcat_fwd_ws + cim_wino7_input_pair)",
                        bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, ms=tot_ms / len(ls),
                        launches=len(ls), algorithmic_bytes=tot_b / len(ls),
                        traffic=(pmc["cim_roi_align_wino7_pair_fwd"]["hbm_bytes_mean"] if pmc_ok and pmc.get("cim_roi_align_wino7_pair_fwd") else None),
                        traffic_source=(pmc["_source"] if pmc_ok and pmc.get("cim_roi_align_wino7_pair_fwd") else None),
                        algorithmic_bytes_note="feature map + rois + masks in, pair image [121][N padded to 32][2 Cf] x 4 B out",
                        survey_8d=dict(algorithmic_bytes=tot_8d / len(ls), frac=tot_8d / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       note="SURVEY.md 8(d)'s formula prices an operator that writes `cat` (4 N 2Cf 49 bytes); this "
                                            "launch writes the 2.47x larger Winograd image instead and the separate transform launch "
                                            "(round 4: 0.288 ms for 401 MB in + 991 MB out) is gone")))
    ls = timer.launches("cim_roi_align_bwd")
    if ls:      # the plain ROIAlign backward on dbox = dcat_lo + mask * dcat_hi (the fold happens in the convolution's last backward stage)
        def rb_bytes(j):
            Hf, Wf = feat[j]
            n = infos[j]["n"]
            return 4.0 * (Cf * Hf * Wf + 5 * n) + 4.0 * n * Cf * 49
        tot_b = float(np.sum([rb_bytes(j) for _, j in ls]))
        tot_8d = float(np.sum([ra_bytes(j) for _, j in ls]))
        tot_ms = float(np.sum([m for m, _ in ls]))
        ach = tot_b / (tot_ms * 1e-3) / 1e9
        hbm.append(dict(kernel="cim_roi_align_bwd_ws (ROIAlign backward on dbox [N,7,7,Cf]: the mask multiply + concat backward is folded into "
                               "cim_wino7_dx_maskfold, which writes dbox instead of dcat [N,7,7,2Cf])",
                        bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, ms=tot_ms / len(ls),
                        launches=len(ls), algorithmic_bytes=tot_b / len(ls),
                        traffic=(pmc["cim_roi_align_bwd"]["hbm_bytes_mean"] + pmc.get("roi_partial_reduce", {}).get("hbm_bytes_mean", 0.0)
                                 if pmc_ok and pmc.get("cim_roi_align_bwd") else None),
                        traffic_source=(pmc["_source"] if pmc_ok and pmc.get("cim_roi_align_bwd") else None),
                        survey_8d=dict(algorithmic_bytes=tot_8d / len(ls), frac=tot_8d / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       note="SURVEY.md 8(d)'s backward reads dcat (4 N 2Cf 49 bytes); this launch reads half of that - the "
                                            "same gradient, combined upstream - so its own algorithmic bytes are 4 (Cf Hf Wf + 5N + 49 N Cf)")))
    for name in ("cim_roi_align_maskcat_fwd", "cim_roi_align_maskcat_bwd"):
        ls = timer.launches(name)
        if ls:
            tot_b = float(np.sum([ra_bytes(j) for _, j in ls]))
            tot_ms = float(np.sum([m for m, _ in ls]))
            ach = tot_b / (tot_ms * 1e-3) / 1e9
            pk = pmc.get(name) if pmc_ok else None
            hbm.append(dict(kernel=name, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                            ms=tot_ms / len(ls), launches=len(ls), algorithmic_bytes=tot_b / len(ls),
                            traffic=pk["hbm_bytes_mean"] if pk else None, traffic_source=pmc["_source"] if pk else None))
    # mining + sampling + assignment (a-4 ... a-6), all launches of an image summed.  Algorithmic bytes after SURVEY.md 8(d),
    # with the per-class seed count and the pseudo-GT count at their upper bounds (S_c = K, G = classes x K):
    #   2N^2 (containment flags, once per image) + layers x [classes x (4N + 2K^2 + 2NK) + 2N G + 12 N (C+1)]
    ls = timer.launches("mining")
    if ls:
        C1 = int(cfg.MODEL.NUM_CLASSES) + 1
        n_img = len(set((i // len(MINING_CALLS)) for i in range(len(ls))))
        tot_ms = float(np.sum([m for m, _ in ls]))
        tot_b = 0.0
        for _, j in ls[::len(MINING_CALLS)]:
            n, n_cls = infos[j]["n"], infos[j]["n_cls"]
            K = int(np.ceil(cfg.p_seed * n))
            tot_b += 2.0 * n * n + cfg.REFINE_TIMES * (n_cls * (4.0 * n + 2.0 * K * K + 2.0 * n * K) + 2.0 * n * n_cls * K + 12.0 * n * C1)
        ach = tot_b / (tot_ms * 1e-3) / 1e9
        step_ms = float(np.sum([m for m, _ in ls[1::len(MINING_CALLS)]]))
        hbm.append(dict(kernel="mining + sampling + assignment (cim_asy_prep: flags + transposed containment map, on the side stream under "
                               "the backbone forward; cim_mining_step: 2 launches on the step's stream; no host round trip)",
                        bound="hbm", achieved=tot_b / (step_ms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=tot_b / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        ms=step_ms / n_img, ms_prep_side_stream=(tot_ms - step_ms) / n_img, ms_both_calls=tot_ms / n_img,
                        algorithmic_bytes=tot_b / n_img, traffic=None, kernel_launches_per_image=4,
                        note="latency-bound: 2 dependent launches over a few MB per image on the step's stream (SURVEY.md 8d).  `ms` = "
                             "cim_mining_step on the step's stream (what the step waits for); the input-only prep (2 launches: flags + "
                             "transposed map) runs on the side stream under the backbone forward - its HIP-event time there "
                             "(ms_prep_side_stream, mostly launch latency of an idle stream) costs no step time; ms_both_calls sums them"))
    # ROIAlign + mining together (north_star: ">= 40 % of the HBM roofline on ROIAlign + mining"): all three calls' algorithmic bytes
    # over the sum of their times - once with the launches' OWN bytes (the forward writes the Winograd image, the backward reads the
    # folded gradient), once with SURVEY.md 8(d)'s formulas for an operator pair that moves `cat` both ways
    trio = [h for h in hbm if h["kernel"].startswith(("cim_roi_align", "mining"))]
    if len(trio) >= 3:
        tms = sum(h["ms"] for h in trio)
        own = sum(h["algorithmic_bytes"] for h in trio)
        s8d = sum(h.get("survey_8d", {}).get("algorithmic_bytes", h["algorithmic_bytes"]) for h in trio)
        hbm.append(dict(kernel="ROIAlign forward + backward + mining (sum of the three calls above)", bound="hbm", ms=tms,
                        algorithmic_bytes=own, achieved=own / (tms * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=own / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS, traffic=None,
                        survey_8d=dict(algorithmic_bytes=s8d, frac=s8d / (tms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       note="round 4, two more launches on the same data (cat written + re-read by the Winograd input "
                                            "transform: 0.288 ms; dcat written in full): 0.131 + 0.208 + 0.090 ms = 23.9 %")))
    metric = "images/sec training step (ResNet-50 VOC, ~1k proposals/img) at 1/2/4/8 GPU" \
        if args.config == "resnet50_voc" else "images/sec training step (%s)" % args.config      # BASELINE.json
    ns = [i["n"] for i in infos]
    shapes = sorted(set("%dx%d" % (i["H"], i["W"]) for i in infos))
    workload = "%s bs=1/GPU, %d distinct images cycled: %d-%d proposals (mean %d), images 3x{%s}, feature %d ch, iter_size=%d" \
        % (args.config, len(infos), min(ns), max(ns), round(float(np.mean(ns))), ",".join(shapes), Cf, args.iter_size)
    return dict(metric=metric, value=images / elapsed, unit="images/s", n_gpus=world, steps=args.steps,
                warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True,
                scaling="weak", vs_baseline=None, dtype="fp32", data="synthetic",
                dtype_note="fp32 tensors and accumulation everywhere; MaskFuse GEMM products evaluated on a scaled two-term fp16 operand "
                           "split written by the operands' producers (3 MFMA products, dropped term <= 2^-22, one power-of-two scale per "
                           "matrix) - measured deviation of losses / gradients from the reference in tests/test_gpu_tolerance.py "
                           "(profiles/r6/parity_deviation.json); the TF32-class single-product line: extra.tf32_class",
                config=dict(workload=workload, parallelism="dp%d" % world, inputs="resident in HBM (extra.with_h2d_upload: "
                            "uploaded from pinned host memory inside every step)"),
                roofline=roofline, roofline_hbm=hbm)


def cpu_baseline(config, budget_n=0):
    """The oracle's CPU restatement of the same step, timed on this host (bounded sample)."""
    import torch
    from cim_amd import synthetic
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling.model_builder import Generalized_RCNN
    from oracle import cpu_step, mask_iou as omi
    apply_preset(config)
    torch.manual_seed(3)
    # 32 threads: measured best on the GPU box's host (2 x EPYC 9575F, 256 hw threads): ATen's CPU
    # convs / GEMMs at these sizes get slower past 32 threads and collapse at 256 (backbone fwd
    # 0.03 s @32, 0.39 s @128, 33 s @256).
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    model = Generalized_RCNN().train()
    inp = synthetic.make_image_inputs(config, seed=3)
    n = inp["rois"].shape[0]
    budget_n = n if budget_n <= 0 else min(budget_n, n)
    iou, asy = omi.mask_iou_maps(inp["full_masks"])
    cpu_step.step(model, inp, iou, asy, n_sub=8, seed=3)          # untimed warm-up (oneDNN primitive creation)
    for p in model.parameters():
        p.grad = None
    tm = {}
    t0 = time.perf_counter()
    cpu_step.step(model, inp, iou, asy, n_sub=budget_n, seed=3, timings=tm)
    wall = time.perf_counter() - t0
    # backbone + mining run on the full workload inside the sample; the proposal-linear phases
    # (ROIAlign, MaskFuse, heads, losses and their backward) ran on n_sub of n proposals.
    scale = n / float(budget_n)
    fixed = tm["backbone_fwd"] + tm["backbone_bwd"]
    linear = tm["roialign_fwd"] + tm["head_fwd"] + tm["losses_fwd"] + tm["head_bwd"] + tm["mining"]
    est = fixed + linear * scale
    return dict(value=1.0 / est, unit="images/s", cores=threads, kind="port",
                sample="oracle/cpu_step.py fwd+bwd of %s (the 1000-proposal 516x688 image of the mix): backbone on the full "
                       "image, ROIAlign/MaskFuse/heads/mining/losses/backward on the first %d of %d proposals, "
                       "proposal-linear phases scaled x%.1f (measured %.1f s; phases %s)"
                       % (config, budget_n, n, scale, wall, {k: round(v, 2) for k, v in tm.items()}))


if __name__ == "__main__":
    main()
