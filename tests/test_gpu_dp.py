"""-m gpu: data parallel on the REAL model.

  * iter_size = 4 gradient accumulation (the reference's operating point, scripts/train_CIM.sh:8,
    tools/train.py:420-438) on Generalized_RCNN: four different images, the unchanged driver calls.
  * two ranks, each with its own image handed over as CPU tensors through nn.DataParallel(minibatch=True):
    gradients == mean of the per-rank gradients.  `nccl` (RCCL, one rank per GPU) when the box has >= 2 GPUs - skipped
    otherwise -, and always `gloo` with both ranks on cuda:0 (the hooks, buckets, in-place big-tensor reduction and the
    construction-time broadcast are backend-independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _small_batch(seed, n=40, dev=None):
    from cim_amd import mask_iou, synthetic
    inp = synthetic.make_image_inputs("resnet50_voc", seed=seed, n=n)
    inp["data"] = inp["data"][:, :, :160, :224].copy()
    inp["rois"][:, 1:] *= np.float32(0.3)
    t = lambda a: torch.from_numpy(a).unsqueeze(0)                # CPU tensors with the loader's batch dimension
    iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(dev))
    return dict(data=[torch.from_numpy(inp["data"])], rois=[t(inp["rois"])], masks=[t(inp["masks"])], labels=[t(inp["labels"])],
                gtrois=[None], mat=[t(inp["mat"])], index=[t(inp["index"])], iou_map=[iou], asy_iou_map=[asy])


def _model(dev, seed=0):
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling.model_builder import Generalized_RCNN
    apply_preset("resnet50_voc")
    torch.manual_seed(seed)
    return Generalized_RCNN().to(dev).train()


def _flat_grads(model):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float()
                      for p in model.parameters() if p.requires_grad])


def _loss(out):
    return sum(v.sum() for v in out["losses"].values())


def test_total_loss_key_is_the_sum_the_driver_differentiates():
    """model_out['total_loss'] (made by the loss launch's finishing kernel): equal to the sum of the four reported losses the
    reference's loop builds (lib/utils/training_stats.py:72-83, tools/train.py:435), and differentiating it gives the same
    gradients bit for bit as differentiating that sum."""
    from cim_amd.nn import DataParallel
    dev = torch.device("cuda:0")
    model = _model(dev)
    dp = DataParallel(model, cpu_keywords=["im_info", "roidb"], minibatch=True)
    batch = _small_batch(77, n=40, dev=dev)
    grads, vals = [], []
    for use_total in (False, True):
        dp.zero_grad()
        np.random.seed(9)
        out = dp(**batch)
        assert out["total_loss"].shape == (1,) and all(v.shape == (1,) for v in out["losses"].values())
        total = 0
        for v in out["losses"].values():          # the driver's own accumulation
            total = total + v.mean(dim=0, keepdim=True)
        assert torch.equal(total.detach(), out["total_loss"].detach())
        (out["total_loss"] if use_total else total).backward()
        torch.cuda.synchronize()
        grads.append(_flat_grads(model).clone())
        vals.append(float(out["total_loss"].detach()))
    assert vals[0] == vals[1] and vals[0] > 0
    assert torch.equal(grads[0], grads[1]) and float(grads[0].abs().max()) > 0


def test_lazy_settle_keeps_the_numpy_stream(monkeypatch):
    """heads.LAZY_SETTLE (bench.py's mode): the generator is not settled at the end of backward but right before the next step's
    draw - three steps give the same losses, the same parameters and the same final np.random position as the default mode, and
    between a backward pass and settle_rng() the generator stands where the pre-drawn uniforms left it."""
    from cim_amd.modeling import heads
    from cim_amd.nn import DataParallel
    from cim_amd.optim import SGD
    dev = torch.device("cuda:0")
    batches = [_small_batch(400 + i, n=40, dev=dev) for i in range(3)]

    def run(lazy):
        monkeypatch.setattr(heads, "LAZY_SETTLE", lazy)
        model = _model(dev, seed=3)
        dp = DataParallel(model, cpu_keywords=["im_info", "roidb"], minibatch=True)
        opt = SGD([p for p in model.parameters() if p.requires_grad], lr=0.01, momentum=0.9, weight_decay=5e-4)
        np.random.seed(123)
        losses, unsettled = [], 0
        for b in batches:
            dp.zero_grad()
            out = dp(**b)
            out["total_loss"].backward()
            unsettled += int(heads._rng.pending is not None)
            opt.step()
            losses.append(float(out["total_loss"].detach()))
        heads.settle_rng()
        probe = np.random.random_sample()
        torch.cuda.synchronize()
        return losses, probe, unsettled, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()

    l0, p0, u0, w0 = run(False)
    l1, p1, u1, w1 = run(True)
    assert u0 == 0 and u1 == 3            # eager: settled by the end-of-backward callback; lazy: still pending after every backward
    assert l0 == l1 and p0 == p1 and torch.equal(w0, w1)


def test_iter_size_4_accumulation_real_model():
    from cim_amd.nn import DataParallel
    dev = torch.device("cuda:0")
    model = _model(dev)
    batches = [_small_batch(100 + i, n=32 + 8 * i, dev=dev) for i in range(4)]       # different N per image
    dp = DataParallel(model, cpu_keywords=["im_info", "roidb"], minibatch=True, iter_size=4)
    singles = []
    for i, b in enumerate(batches):
        dp.zero_grad()
        np.random.seed(500 + i)
        _loss(dp(**b)).backward()
        singles.append(_flat_grads(model).clone())
    dp.zero_grad()
    for i, b in enumerate(batches):                      # tools/train.py:419-438 verbatim: no extra calls
        np.random.seed(500 + i)
        _loss(dp(**b)).backward(retain_graph=True)
    acc = _flat_grads(model)
    want = sum(singles)
    # every kernel of the ResNet-50 step is the build's own and deterministic (no library convolution left, fixed-order
    # split-K and partial-map reductions), and autograd accumulates g0 + g1 + g2 + g3 in the order `sum` does: bit-equal
    assert torch.equal(acc, want), float((acc - want).norm() / want.norm())
    assert float((acc - singles[0]).norm() / want.norm()) > 0.1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, backend, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from cim_amd.modeling import heads
    from cim_amd.nn import DataParallel
    model = _model(dev, seed=rank)                       # DIFFERENT initial weights per rank: construction must broadcast rank 0's
    dp = DataParallel(model, cpu_keywords=["im_info", "roidb"], minibatch=True, big_bytes=1 << 20, bucket_bytes=4 << 20)
    ref = _model(dev, seed=0)
    for (n, a), (_, b) in zip(model.named_parameters(), ref.named_parameters()):
        assert torch.equal(a, b), "parameter %s was not broadcast from rank 0" % n
    assert any("tensor" in b for b in dp.buckets) and len(dp.buckets) > 3
    checks, affine = [], []
    for step in range(2):                                # two steps: a different image per rank AND per step
        batch = _small_batch(10 * step + rank, n=40 + 8 * rank, dev=dev)
        # local gradient of this rank's image on an unwrapped copy
        ref.load_state_dict(model.state_dict())
        ref.zero_grad(set_to_none=True)
        np.random.seed(7 + rank)
        _loss(ref(**{k: (v[0].to(dev) if torch.is_tensor(v[0]) else v[0]) for k, v in batch.items()})).backward()
        local = _flat_grads(ref)
        mean = local.clone()
        dist.all_reduce(mean)
        mean /= world
        dp.zero_grad()
        np.random.seed(7 + rank)
        _loss(dp(**batch)).backward()                    # reduction completes inside (end-of-backward callback)
        assert not dp._pending
        got = _flat_grads(model)
        torch.cuda.synchronize()
        checks.append((float((got - mean).norm() / mean.norm()), float((local - mean).norm() / mean.norm())))
        # the gamma / beta gradients of chained BatchNorm layers (bn1, bn2 of a bottleneck) do not come through autograd: they
        # are installed at the end of the backward pass (ops/conv1x1.py: finish_affine) like the deferred weight gradients, and
        # go out with the backbone's bucket at the forced flush - each of them must be the MEAN over the ranks, on every rank
        off = 0
        for (name, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            if not p.requires_grad:
                continue
            n = p.numel()
            if (".bn1." in name or ".bn2." in name) and name.startswith("Conv_Body.res"):
                m, l, g = mean[off:off + n], local[off:off + n], got[off:off + n]
                affine.append((name, float((g - m).abs().max() / m.abs().max().clamp(min=1e-20)),
                               float((l - m).abs().max() / m.abs().max().clamp(min=1e-20))))
            off += n
        mine = got.cpu() if backend == "gloo" else got
        both = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        assert torch.equal(both[0], both[1]), "the ranks hold different gradients after the reduction"
        heads.settle_rng()
    if rank == 0:
        torch.save(dict(checks=checks, affine=affine), out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_ranks_real_model(tmp_path, backend):
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL test needs >= 2 GPUs (this box has %d)" % torch.cuda.device_count())
    out = str(tmp_path / "checks.pt")
    mp.spawn(_worker, args=(2, _free_port(), backend, out), nprocs=2, join=True)
    r = torch.load(out)
    for err, spread in r["checks"]:
        assert spread > 1e-2, "the two ranks' gradients should differ (different images)"
        assert err < 2e-3, "all-reduced gradient differs from the mean of the per-rank gradients: %.3g" % err
    # chained BatchNorm affine gradients (installed at the end of the pass, not by autograd) are all-reduced too
    assert len(r["affine"]) >= 2 * 2 * (4 + 6), len(r["affine"])          # 2 steps x (bn1, bn2) x (weight, bias) x res3 + res4 blocks
    for name, err, spread in r["affine"]:
        assert err < 5e-3, "%s: all-reduced gradient is not the mean of the ranks' gradients (%.3g)" % (name, err)
    assert max(sp for _, _, sp in r["affine"]) > 1e-2, "the ranks' BatchNorm affine gradients should differ before the reduction"


def _schedule_worker(rank, world, port, ref_path, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cim_amd.modeling import heads
    from cim_amd.nn import DataParallel
    from cim_amd.ops import gemm, maskfuse_pair
    model = _model(dev, seed=0)
    dp = DataParallel(model, cpu_keywords=["im_info", "roidb"], minibatch=True)          # production bucket sizes
    assert gemm.publisher_for(model.Box_Head.seg_fc[0].weight) is not None
    batch = _small_batch(5, n=48, dev=dev)               # the SAME image on both ranks: mean of the two gradients == each of them
    maskfuse_pair.SCHEDULE.clear()
    dp.zero_grad()
    np.random.seed(11)
    _loss(dp(**batch)).backward()
    assert not dp._pending
    torch.cuda.synchronize()
    heads.settle_rng()
    got = _flat_grads(model).cpu()
    want = torch.load(ref_path)
    sched = dict(maskfuse_pair.SCHEDULE)
    if rank == 0:
        torch.save(dict(equal=bool(torch.equal(got, want)), rel=float((got - want).norm() / want.norm()), schedule=sched,
                        big_buckets=sum(1 for b in dp.buckets if "tensor" in b)), out)
    dist.barrier()
    dist.destroy_process_group()


def test_multi_rank_schedule_runs_and_equals_the_single_rank_schedule(tmp_path):
    """VERDICT r3 item 6.  The backward schedule of the fused MaskFuse node differs with several ranks (ops/maskfuse_pair.py: the
    three weight-gradient products go out WHOLE and each is handed to nn.DataParallel the moment it is enqueued - its all-reduce
    starts behind it on the side stream; no launches of 256 workgroups, nothing postponed behind the ROIAlign backward).  Two gloo
    ranks on cuda:0 with the same image: (i) those branches ran - three early publications, no chunked or postponed launch - and
    (ii) the averaged gradients equal the single-process schedule's bit for bit (same kernels on the same operands; the mean of two
    equal fp32 values is exact), computed beforehand in THIS process without a process group."""
    from cim_amd.modeling import heads
    from cim_amd.ops import gemm, maskfuse_pair
    dev = torch.device("cuda:0")
    model = _model(dev, seed=0)
    batch = _small_batch(5, n=48, dev=dev)
    maskfuse_pair.SCHEDULE.clear()
    np.random.seed(11)
    _loss(model(**{k: (v[0].to(dev) if torch.is_tensor(v[0]) else v[0]) for k, v in batch.items()})).backward()
    torch.cuda.synchronize()
    heads.settle_rng()
    single = dict(maskfuse_pair.SCHEDULE)
    assert gemm.publisher_for(model.Box_Head.seg_fc[0].weight) is None and single.get("late_launches_chunked", 0) == 1 and single.get("weight_gradients_published_early", 0) == 0, single
    assert single.get("late_launches_postponed_behind_roi_align", 0) == 1, single
    ref_path, out = str(tmp_path / "single.pt"), str(tmp_path / "multi.pt")
    torch.save(_flat_grads(model).cpu(), ref_path)
    del model
    torch.cuda.empty_cache()
    mp.spawn(_schedule_worker, args=(2, _free_port(), ref_path, out), nprocs=2, join=True)
    r = torch.load(out)
    s = r["schedule"]
    assert r["big_buckets"] == 3, r
    assert s.get("weight_gradients_published_early", 0) == 3 and s.get("late_launches_whole_products", 0) == 1, s
    assert s.get("late_launches_chunked", 0) == 0 and s.get("late_launches_postponed_behind_roi_align", 0) == 0, s
    assert r["equal"], "multi-rank schedule's gradients differ from the single-rank schedule's: %.3g" % r["rel"]


def test_two_models_in_one_process_step_alternately():
    """The scheduling state behind a step is keyed by what it belongs to - deferred weight gradients by parameter, the chained
    BatchNorm hand-over by the producer's autograd node, the early publisher by the wrapper's parameters, cached weight images by
    weight tensor - so two models alive in one process that step ALTERNATELY (and once with both backward passes in ONE autograd
    run) end with the parameters each of them reaches alone, bit for bit."""
    from cim_amd.optim import SGD
    dev = torch.device("cuda:0")
    batches = [_small_batch(700 + i, n=40 + 8 * (i % 2), dev=dev) for i in range(4)]
    unwrap = lambda b: {k: (v[0].to(dev) if torch.is_tensor(v[0]) else v[0]) for k, v in b.items()}

    def make(seed):
        m = _model(dev, seed=seed)
        bias = [p for n, p in m.named_parameters() if p.requires_grad and "bias" in n]
        rest = [p for n, p in m.named_parameters() if p.requires_grad and "bias" not in n]
        return m, SGD([dict(params=rest, lr=0.01, weight_decay=5e-4), dict(params=bias, lr=0.02, weight_decay=0.0)], lr=0.01, momentum=0.9)

    def step(m, opt, b, seed):
        opt.zero_grad(set_to_none=True)
        np.random.seed(seed)
        _loss(m(**unwrap(b))).backward()
        opt.step()

    solo = {}
    for tag, seed in (("a", 1), ("b", 2)):
        m, opt = make(seed)
        for i in range(3):
            step(m, opt, batches[i if tag == "a" else 3 - i], 60 + i)
        torch.cuda.synchronize()
        solo[tag] = {n: p.detach().clone() for n, p in m.named_parameters()}
        del m, opt
    (ma, oa), (mb, ob) = make(1), make(2)
    for i in range(3):
        step(ma, oa, batches[i], 60 + i)
        step(mb, ob, batches[3 - i], 60 + i)
    torch.cuda.synchronize()
    for tag, m in (("a", ma), ("b", mb)):
        for n, p in m.named_parameters():
            assert torch.equal(p.detach(), solo[tag][n]), (tag, n)
    # both models in ONE backward pass: each still gets its own gradients (compare with separate passes)
    from cim_amd.modeling import heads
    grads = {}
    for mode in ("separate", "joint"):
        for m in (ma, mb):
            m.zero_grad(set_to_none=True)
        np.random.seed(99)                  # ONE NumPy stream, as in the reference: settled right after each forward
        la = _loss(ma(**unwrap(batches[0])))
        heads.settle_rng()
        if mode == "separate":
            la.backward()
        lb = _loss(mb(**unwrap(batches[1])))
        heads.settle_rng()
        if mode == "separate":
            lb.backward()
        else:
            (la + lb).backward()
        torch.cuda.synchronize()
        grads[mode] = (_flat_grads(ma).clone(), _flat_grads(mb).clone())
    assert torch.equal(grads["joint"][1], grads["separate"][1])
    assert torch.equal(grads["joint"][0], grads["separate"][0])


def test_early_optimizer_step_is_identical():
    """nn.DataParallel.attach_optimizer: the MaskFuse / heads parameters are updated inside the last backward pass on a
    side stream (overlapped with the backbone backward), the rest by optimizer.step() - same kernel, same arithmetic: the
    parameters and momentum buffers after three steps (the third with iter_size = 2) equal the plain optimizer.step() run
    bit for bit."""
    from cim_amd.nn import DataParallel
    from cim_amd.optim import SGD
    dev = torch.device("cuda:0")
    batches = [_small_batch(300 + i, n=40, dev=dev) for i in range(4)]

    def run(early):
        model = _model(dev, seed=5)
        dp = DataParallel(model, cpu_keywords=["im_info", "roidb"], minibatch=True)
        bias = [p for n, p in model.named_parameters() if p.requires_grad and "bias" in n]
        rest = [p for n, p in model.named_parameters() if p.requires_grad and "bias" not in n]
        opt = SGD([dict(params=rest, lr=0.01, weight_decay=5e-4), dict(params=bias, lr=0.02, weight_decay=0.0)], lr=0.01, momentum=0.9)
        if early:
            assert dp.attach_optimizer(opt)
        k = 0
        for step, iters in enumerate((1, 1, 2)):
            dp.iter_size = iters
            dp.zero_grad()
            for _ in range(iters):
                np.random.seed(40 + k)
                _loss(dp(**batches[k])).backward()
                k += 1
            if early:
                assert opt._early is not None and len(opt._early[0]) > 10, "the early update did not start inside backward"
            opt.step()
            assert opt._early is None
        torch.cuda.synchronize()
        return ({n: p.detach().clone() for n, p in model.named_parameters()},
                {n: opt.state[p]["momentum_buffer"].clone() for n, p in model.named_parameters() if p in opt.state})

    p0, m0 = run(False)
    p1, m1 = run(True)
    # same kernels, same arithmetic, deterministic step (no library convolution in the ResNet-50 body): bit for bit
    for n in p0:
        assert torch.equal(p1[n], p0[n]), n
    for n in m0:
        assert torch.equal(m1[n], m0[n]), n


def test_bench_two_ranks_launch_path():
    """`python bench.py --gpus 2` from a bare shell - the exact command the driver's SCALE run issues -
    must start its ranks as children before touching the GPU, run the step with real collectives and print ONE JSON line
    with n_gpus = 2, the process group's own world size and the communication figures.  gloo on a 1-GPU box (both ranks
    share cuda:0), nccl = RCCL when two devices are visible."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--backend", backend, "--steps", "2", "--warmup", "1",
           "--images", "2", "--no-cpu-baseline", "--sustained", "0"]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["dist"]["world_size"] == 2 and line["dist"]["backend"] == backend and len(line["dist"]["per_rank_images_per_s"]) == 2
    assert abs(line["value"] - 2 * min(line["dist"]["per_rank_images_per_s"])) < 1e-6 * line["value"]     # MAX over ranks
    comm = line["extra"]["comm"]
    assert comm["buckets"] >= 3 and comm["gradient_bytes"] > 9e8 and comm["ms_per_step_no_sync"] > 0
    assert "roofline" in line and "cpu_baseline" not in line
