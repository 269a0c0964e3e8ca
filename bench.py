#!/usr/bin/env python3
"""Headline benchmark: images/sec of the CIM per-image training step (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one synthetic image per GPU through the whole hot path: backbone fwd -> fused
ROIAlign+mask-cat -> MaskFuse -> 8 heads -> 3 x CIM mining (+ host anti-noise sampling) -> 4 losses
-> backward -> gradient all-reduce (RCCL) -> SGD step.  Workload at N=1 = BASELINE configs[1]
(resnet50_voc, bs=1, 1000 proposals, 516x688 image).  Weak scaling: one image per rank.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from cim_amd import _lib, mask_iou, synthetic  # noqa: E402
from cim_amd.core.config import cfg  # noqa: E402
from cim_amd.core.presets import apply_preset  # noqa: E402
from cim_amd.modeling.model_builder import Generalized_RCNN  # noqa: E402
from cim_amd.nn import DataParallel  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TF = 2516.6   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (256 CU x 4096 flop/clk x 2.4 GHz)


class KernelTimer:
    """HIP events around selected C-ABI launches / module calls, recorded on the launch stream
    inside the timed region (torch.cuda.Event records on torch's current stream, which is the
    stream every launch here uses)."""

    def __init__(self):
        self.spans = {}
        self.enabled = False

    def span(self, name):
        timer = self

        class _Ctx:
            def __enter__(self):
                if timer.enabled:
                    self.a = torch.cuda.Event(enable_timing=True)
                    self.b = torch.cuda.Event(enable_timing=True)
                    self.a.record()

            def __exit__(self, *exc):
                if timer.enabled:
                    self.b.record()
                    timer.spans.setdefault(name, []).append((self.a, self.b))
        return _Ctx()

    def mean_ms(self, name):
        ev = self.spans.get(name, [])
        return float(np.mean([a.elapsed_time(b) for a, b in ev])) if ev else None


MINING_CALLS = ("cim_asy_flag", "cim_seed_select", "cim_contain_argmax", "cim_arbitrate", "cim_assign")


def instrument(model, timer):
    orig_call = _lib.call

    state = {"conv": 0, "bg": 0}

    def call(name, *args):
        if name in ("cim_roi_align_maskcat_fwd", "cim_roi_align_maskcat_fwd_ws", "cim_roi_align_maskcat_bwd",
                    "cim_roi_align_maskcat_bwd_ws"):
            if "bwd" not in name:
                state["conv"] = state["bg"] = 0
            with timer.span("cim_roi_align_maskcat_bwd" if "bwd" in name else "cim_roi_align_maskcat_fwd"):
                return orig_call(name, *args)
        if name == "cim_conv3x3_f32":       # per step: 1st launch = forward, 2nd = data gradient
            state["conv"] += 1
            with timer.span("maskfuse_conv_fwd" if state["conv"] == 1 else "maskfuse_conv_dgrad"):
                return orig_call(name, *args)
        if name in ("cim_gemm_f32_batched", "cim_gemm_f16x2_batched"):  # Winograd-domain GEMMs, per step: forward, data grad, weight grad
            state["bg"] += 1
            with timer.span(("wino_gemm_fwd", "wino_gemm_dgrad", "wino_gemm_wgrad")[min(state["bg"], 3) - 1]):
                return orig_call(name, *args)
        if name == "cim_conv3x3_wgrad_f32":
            with timer.span("maskfuse_conv_wgrad"):
                return orig_call(name, *args)
        if name in MINING_CALLS:            # a-4 ... a-6: every mining / assignment launch of the three CIM layers
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
    for m in model.modules():
        if hasattr(m, "bn3"):                                   # bottleneck (ResNet-50, HRNet stem / head)
            torch.nn.init.constant_(m.bn3.weight, 0.25)
        elif hasattr(m, "bn2") and hasattr(m, "downsample"):    # HRNet basic block
            torch.nn.init.constant_(m.bn2.weight, 0.25)


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in
    separate runs of this script, tools/pmc_traffic.py); None when the file is absent."""
    path = os.path.join(REPO, "profiles", "r1", "pmc_traffic.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def make_optimizer(model):
    """Param groups of tools/train.py:282-311 (bias: lr x2, no weight decay), SGD momentum 0.9."""
    bias, nonbias = [], []
    for name, p in model.named_parameters():
        if p.requires_grad:
            (bias if "bias" in name else nonbias).append(p)
    lr, wd = 0.0005, 0.0005                                   # configs/resnet50_voc.yaml SOLVER
    groups = [dict(params=nonbias, lr=lr, weight_decay=wd), dict(params=bias, lr=2 * lr, weight_decay=0.0)]
    if os.environ.get("CIM_OPTIM", "hip") == "aten":
        return torch.optim.SGD(groups, lr=lr, momentum=0.9, fused=True)
    from cim_amd.optim import SGD          # fused multi-tensor SGD, one HIP launch per step (cim_amd/csrc/sgd.hip)
    return SGD(groups, lr=lr, momentum=0.9)


def cpu_baseline(config, budget_n=0):
    """The oracle's CPU restatement of the same step, timed on this host (bounded sample)."""
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
                sample="oracle/cpu_step.py fwd+bwd of %s: backbone on the full image, ROIAlign/MaskFuse/heads/"
                       "mining/losses/backward on the first %d of %d proposals, proposal-linear phases scaled x%.1f "
                       "(measured %.1f s; phases %s)" % (config, budget_n, n, scale, wall,
                                                         {k: round(v, 2) for k, v in tm.items()}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="resnet50_voc")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="proposals in the CPU sample (0 = all)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a functional "
                                                       "multi-rank check on a box with fewer GPUs than ranks)")
    ap.add_argument("--miopen-find", action="store_true",
                    help="let MIOpen time its solvers for the backbone convs (default: immediate mode)")
    args = ap.parse_args()

    if os.environ.get("CIM_BENCH_WATCHDOG"):        # debugging aid: dump all thread stacks if a phase stalls
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["CIM_BENCH_WATCHDOG"]), repeat=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    if args.backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)      # functional check: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.backends.cudnn.benchmark = args.miopen_find      # backbone convs (a-11) go through MIOpen
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
    timer = KernelTimer()
    instrument(model, timer)
    dp = DataParallel(model, cpu_keywords=["im_info", "roidb"], minibatch=True)
    opt = make_optimizer(model)

    inp = synthetic.make_image_inputs(args.config, seed=cfg.RNG_SEED + rank)     # one image per rank
    n = inp["rois"].shape[0]
    iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(dev))
    t = lambda a: [torch.from_numpy(a).unsqueeze(0).to(dev)]
    batch = dict(data=[torch.from_numpy(inp["data"]).to(dev)], rois=t(inp["rois"]), masks=t(inp["masks"]),
                 labels=t(inp["labels"]), gtrois=[None], mat=t(inp["mat"]), index=t(inp["index"]),
                 iou_map=[iou], asy_iou_map=[asy])
    np.random.seed(cfg.RNG_SEED + rank)                        # the anti-noise sampling stream

    feat_hw = [0, 0]

    def step():
        dp.zero_grad()
        out = dp(**batch)
        feat_hw[:] = out["blob_conv"].shape[-2:]
        loss = sum(v.sum() for v in out["losses"].values()) * dp.loss_scale()
        loss.backward()
        dp.finish_gradient_sync()
        opt.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if os.environ.get("CIM_BENCH_PER_STEP") == "1" and rank == 0:      # debugging aid: synchronized time of every step from the first
        for i in range(args.warmup + args.steps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            print("step %d: %.2f ms" % (i, 1e3 * (time.perf_counter() - t1)), file=sys.stderr)
    for _ in range(args.warmup):
        step()
    timer.enabled = True
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    assert torch.isfinite(loss), "non-finite loss in the timed region"

    if rank == 0:
        H, W = inp["image_hw"]
        Cf = model.Conv_Body.dim_out
        Hf, Wf = feat_hw
        # dominant kernel: the exact-fp32 MFMA GEMM of the MaskFuse 3x3 conv forward (a-2).
        #   Winograd F(2x2,3x3) (default): 16 batched GEMMs [16N x 2Cf] x [2Cf x Cf] = 2*16*16N*2Cf*Cf flops
        #   direct implicit GEMM (CIM_CONV_ALGO=direct): 2*49N*18Cf*Cf flops        
        #   Winograd F(4x4,3x3) (CIM_CONV_ALGO=winograd4): 36 batched GEMMs [4N x 2Cf] x [2Cf x Cf]
        wino_ms = timer.mean_ms("wino_gemm_fwd")
        from cim_amd.ops import gemm as gemm_mod
        #   mixed tiling (CIM_CONV_ALGO=winograd7, default): 121 batched GEMMs [N x 2Cf] x [2Cf x Cf]
        if wino_ms:
            _, npos, rows = gemm_mod._wino_geometry(gemm_mod.CONV_ALGO, 7, n)
            what = {"winograd7": "Winograd 4+3 mixed tiling", "winograd4": "Winograd F(4x4,3x3)", "winograd": "Winograd F(2x2,3x3)"}
            kname = "gemm_f32_kernel<A_KCONTIG,B_NCONTIG> x%d (MaskFuse conv3x3 fwd, %s domain)" \
                % (npos, what.get(gemm_mod.CONV_ALGO, gemm_mod.CONV_ALGO))
            conv_ms, conv_flops = wino_ms, 2.0 * npos * rows * (2 * Cf) * Cf
        else:
            kname = "gemm_f32_kernel<A_CONV_K,B_NCONTIG> (MaskFuse conv3x3 fwd, implicit GEMM)"
            conv_ms, conv_flops = timer.mean_ms("maskfuse_conv_fwd"), 2.0 * 49 * n * (2 * Cf * 9) * Cf
        # split engines: every algorithmic fp32 multiply-add is executed as several half-precision MFMA products with
        # fp32 accumulation, so the kernel is priced against the bf16/f16 MFMA peak with
        # achieved = products x algorithmic flops / time.  f16x2 (default): 3 products (scaled two-term fp16 split);
        # bf16x3: 6 products (exact three-term bf16 split); fp32: v_mfma_f32_32x32x2_f32 against its own peak.
        engine = gemm_mod.ENGINE
        products = {"f16x2": 3.0, "bf16x3": 6.0, "fp32": 1.0}[engine]
        kern = {"f16x2": "gemm_f16x2_kernel", "bf16x3": "gemm_bf16x3_kernel", "fp32": "gemm_f32_kernel"}[engine]
        alg_tf = conv_flops / (conv_ms * 1e-3) / 1e12
        kname = kname.replace("gemm_f32_kernel", kern)
        roofline = dict(bound="mfma", kernel=kname, achieved=products * alg_tf,
                        peak=FP32_MFMA_PEAK_TF if engine == "fp32" else BF16_MFMA_PEAK_TF, unit="TFLOP/s", traffic=None,
                        ms=conv_ms,
                        engine={"f16x2": "f16x2: 3 f16 MFMA products per fp32 multiply-add (scaled two-term split), fp32 accumulate",
                                "bf16x3": "bf16x3: 6 bf16 MFMA products per fp32 multiply-add, fp32 accumulate",
                                "fp32": "fp32: v_mfma_f32_32x32x2_f32"}[engine],
                        algorithmic_tflops=alg_tf, fp32_mfma_peak=FP32_MFMA_PEAK_TF)
        roofline["frac"] = roofline["achieved"] / roofline["peak"]
        pmc = pmc_traffic()
        g00 = pmc.get(kern + "<0, 0>")
        if wino_ms and g00 and pmc.get("_conv_algo", "winograd") == gemm_mod.CONV_ALGO:      # first <A_KCONTIG,B_NCONTIG> launch of a step = the Winograd forward GEMM
            roofline["traffic"] = (g00["fetch_kib_per_dispatch"][0] + g00["write_kib_per_dispatch"][0]) * 1024
        # HBM-bound hand-written kernels: fused ROIAlign+mask-cat fwd / bwd
        ra_bytes = 4.0 * (Cf * Hf * Wf + 5 * n + 49 * n) + 4.0 * n * 2 * Cf * 49
        hbm = []
        for name in ("cim_roi_align_maskcat_fwd", "cim_roi_align_maskcat_bwd"):
            ms = timer.mean_ms(name)
            if ms:
                ach = ra_bytes / (ms * 1e-3) / 1e9
                pk = None
                for kn, kv in pmc.items():        # whichever forward / backward kernel variant the profiled run used
                    if kn.startswith("roi_align_fwd" if name.endswith("fwd") else "roi_align_bwd"):
                        pk = kv
                hbm.append(dict(kernel=name, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                                frac=ach / HBM_PEAK_GBS, ms=ms, algorithmic_bytes=ra_bytes,
                                traffic=pk["hbm_bytes_mean"] if pk else None))
        # mining + assignment (a-4 ... a-6), all launches of a step summed.  Algorithmic bytes after SURVEY.md 8(d), with the
        # per-class seed count and the pseudo-GT count at their upper bounds (S_c = K, G = classes x K):
        #   2N^2 (containment flags, once per step) + layers x [classes x (4N + 2K^2 + 2NK) + 2N G + 12 N (C+1)]
        mining_spans = timer.spans.get("mining", [])
        if mining_spans:
            per_step = len(mining_spans) / args.steps
            mining_ms = float(np.sum([a.elapsed_time(b) for a, b in mining_spans])) / args.steps
            n_cls = int((inp["labels"] > 0).sum())
            K = int(np.ceil(cfg.p_seed * n))
            C1 = int(cfg.MODEL.NUM_CLASSES) + 1
            mining_bytes = 2.0 * n * n + cfg.REFINE_TIMES * (n_cls * (4.0 * n + 2.0 * K * K + 2.0 * n * K) + 2.0 * n * n_cls * K + 12.0 * n * C1)
            ach = mining_bytes / (mining_ms * 1e-3) / 1e9
            hbm.append(dict(kernel="mining + assignment (cim_asy_flag, cim_seed_select, cim_contain_argmax, cim_arbitrate, cim_assign)",
                            bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS, ms=mining_ms,
                            algorithmic_bytes=mining_bytes, traffic=None, launches_per_step=per_step,
                            note="launch-latency-bound: ~%d dependent launches of a few MB per step (SURVEY.md 8d)" % round(per_step)))
        metric = "images/sec training step (ResNet-50 VOC, ~1k proposals/img) at 1/2/4/8 GPU" \
            if args.config == "resnet50_voc" else "images/sec training step (%s)" % args.config      # BASELINE.json
        line = dict(metric=metric,
                    value=world * args.steps / elapsed, unit="images/s", n_gpus=world, steps=args.steps,
                    warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True,
                    scaling="weak", vs_baseline=None, dtype="fp32", data="synthetic",
                    dtype_note={"f16x2": "fp32 tensors and accumulation everywhere; MaskFuse GEMM products evaluated on a scaled "
                                         "two-term fp16 operand split (3 MFMA products, dropped term <= 2^-22, rms 2^-25.6) - "
                                         "measured error vs fp64 in the class of the f32-multiply engine (CIM_GEMM_ENGINE=fp32)",
                                "bf16x3": "fp32 tensors and accumulation everywhere; MaskFuse GEMM products evaluated as an exact "
                                          "3 x bf16 operand split (6 MFMA products, dropped terms < 2^-23) - measured error vs fp64 "
                                          "below the f32-multiply engine's (CIM_GEMM_ENGINE=fp32)",
                                "fp32": "fp32 multiplies and accumulation"}[engine],
                    config=dict(workload="%s bs=1/GPU, %d proposals, image 3x%dx%d, feature %dx%dx%d, iter_size=1"
                                         % (args.config, n, H, W, Cf, Hf, Wf), parallelism="dp%d" % world),
                    roofline=roofline, roofline_hbm=hbm)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.config, args.cpu_sample)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
