"""CPU-side checks: the C-ABI library loads and exports every symbol include/cim_hip.h declares,
the config loader, and the torch-level losses / heads against the reference goldens."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from cases import MINING_CASES, case_inputs, procedural

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from cim_amd import _lib, build
    build.build()
    header = open(os.path.join(REPO, "include", "cim_hip.h")).read()
    declared = set(re.findall(r"\b(cim_[a-z0-9_]+)\s*\(", header))
    assert {"cim_roi_align_fwd", "cim_mining_step", "cim_mask_iou_pair", "cim_asy_flag", "cim_losses_fwd"} <= declared
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name
    assert declared - {"cim_last_error", "cim_abi_version"} == set(_lib.SIGNATURES), "ctypes table out of sync with the header"
    assert _lib.load().cim_abi_version() >= 1


def test_experiments_library_builds_and_exports_only_its_own_header():
    """experiments/ (superseded engines, test comparators) builds against the CURRENT product header - a changed product signature
    must not break it - and exports what experiments/include/cim_exp.h declares, no product entry point besides the error pair."""
    import subprocess
    from experiments import build as xbuild
    xbuild.build()
    header = open(os.path.join(REPO, "experiments", "include", "cim_exp.h")).read()
    declared = set(re.findall(r"\b(cim_[a-z0-9_]+)\s*\(", header))
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(REPO, "experiments", "libcim_exp.so")], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (cim_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert exported - declared <= {"cim_last_error", "cim_abi_version", "cim_set_last_error"}, exported - declared


def test_no_cpu_fallback():
    from cim_amd import _lib
    from cim_amd.modeling import heads
    from cim_amd.ops import roi_align
    with pytest.raises(_lib.CimHipError):
        roi_align(torch.randn(1, 4, 5, 5), torch.zeros(1, 5), 7, 0.0625)
    with pytest.raises(_lib.CimHipError):
        heads.CIM_layer()(torch.rand(8, 21), torch.rand(8, 21), None, torch.zeros(1, 20), None, None)


def test_presets_and_model_construction():
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling.model_builder import Generalized_RCNN
    cfg = apply_preset("vgg16_voc")
    m = Generalized_RCNN()
    keys = set(m.state_dict().keys())
    assert {"Conv_Body.conv5.4.weight", "Box_Head.mask_branch.0.weight", "Box_Head.seg_fc.0.weight",
            "Box_Head.seg_fc.2.bias", "cls_iou_model.refine_iou.2.weight", "cls_iou_model.detector.bias"} <= keys
    assert m.Box_Head.mask_branch[0].weight.shape == (512, 1024, 3, 3)
    assert not any(k.startswith("CIM_layer_list") for k in keys)
    thr = [(l.cls_thr, l.iou_thr) for l in m.CIM_layer_list]
    np.testing.assert_allclose(thr, [(0.25, 0.5), (0.35, 0.6), (0.45, 0.7)])
    assert not m.Conv_Body.conv1[0].weight.requires_grad and m.Conv_Body.conv3[0].weight.requires_grad
    wmap, orphans = m.detectron_weight_mapping
    assert "Conv_Body.conv1.0.weight" in wmap and orphans == []
    apply_preset("resnet50_voc")
    assert cfg.MODEL.CONV_BODY == "resnet50.torch_resnet50"


@pytest.mark.parametrize("name", ["n300_c20_k2", "n1000_c80_k3"])
def test_torch_losses_match_reference(name, golden_dir):
    from cim_amd.modeling import heads
    g = np.load(os.path.join(golden_dir, "losses_%s.npz" % name))
    m = np.load(os.path.join(golden_dir, "mining_%s.npz" % name))
    inp = case_inputs(MINING_CASES[name])
    t = torch.from_numpy
    for dt, tag, rtol in ((torch.float32, "f32", 2e-5), (torch.float64, "f64", 1e-10)):
        labels = t(inp["labels"]).to(dt)
        for li in range(3):
            lmda = 3 if li == 0 else 1
            cls, _, iou = inp["layers"][li]
            got = heads.cls_iou_loss(t(cls).to(dt), t(iou).to(dt), t(m["l%d_pseudo_labels" % li]).to(dt),
                                     t(m["l%d_pseudo_iou_labels" % li]), lmda * t(m["l%d_loss_weights" % li]).to(dt), labels)
            np.testing.assert_allclose([float(x) for x in got], g["%s_l%d_cls_iou_bag" % (tag, li)], rtol=rtol)
        cls, det, _ = inp["layers"][0]
        np.testing.assert_allclose(float(heads.mil_bag_loss(t(cls).to(dt), t(det).to(dt), labels)), g[tag + "_mil_bag"], rtol=rtol)
        np.testing.assert_allclose(float(heads.PCL_loss(t(cls).to(dt), t(inp["mat"]).to(dt), labels)), g[tag + "_pcl"],
                                   rtol=max(rtol, 2e-7))
        cls, _, iou = inp["layers"][1]
        bg = torch.zeros(cls.shape, dtype=dt)
        bg[::3, 0] = 1
        got = heads.cls_iou_loss(t(cls).to(dt), t(iou).to(dt), bg, t(m["l1_pseudo_iou_labels"]),
                                 t(m["l1_loss_weights"]).to(dt), labels)
        np.testing.assert_allclose([float(x) for x in got], g[tag + "_bgonly_cls_iou_bag"], rtol=rtol, atol=1e-12)
        with pytest.raises(AssertionError):           # same error behaviour as heads.py:51
            heads.cls_iou_loss(t(cls).to(dt), t(iou).to(dt), torch.zeros(cls.shape, dtype=dt),
                               t(m["l1_pseudo_iou_labels"]), t(m["l1_loss_weights"]).to(dt), labels)


def test_cls_iou_model_matches_reference(golden_dir):
    from cim_amd.modeling import heads
    g = np.load(os.path.join(golden_dir, "heads_small.npz"))
    model = heads.cls_iou_model(64, 21, 3)
    with torch.no_grad():
        for k, (_, p) in enumerate(model.named_parameters()):
            p.copy_(torch.from_numpy(procedural(tuple(p.shape), k + 1)))
        pc, pd, rc, ri = model(torch.from_numpy(procedural((50, 64), 99) * 20))
    np.testing.assert_allclose(pc.numpy(), g["predict_cls"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pd.numpy(), g["predict_det"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(torch.stack(rc).numpy(), g["refine_cls"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(torch.stack(ri).numpy(), g["refine_iou"], rtol=1e-5, atol=1e-7)
    assert [n for n, _ in model.named_parameters()][:4] == ["classifier.weight", "classifier.bias", "detector.weight", "detector.bias"]


def test_hrnet_w48_matches_reference(golden_dir):
    """cfg5 body: same state_dict keys, frozen set and output as the reference's HRNet.py on procedural weights."""
    from cases import procedural_init
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling.HRNet import get_HRNet
    g = np.load(os.path.join(golden_dir, "hrnet_w48.npz"))
    apply_preset("hrnet48_voc")
    m = get_HRNet().train()
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert [n for n, p in m.named_parameters() if p.requires_grad] == [str(k) for k in g["trainable"]]
    procedural_init(m)
    y = m(torch.from_numpy(procedural((1, 3, 150, 220), 4242) * 10.0))
    assert y.shape == g["out"].shape
    np.testing.assert_allclose(y.detach().numpy(), g["out"], rtol=1e-4, atol=1e-5)
    y.sum().backward()
    assert [n for n, p in m.named_parameters() if p.grad is not None] == [str(k) for k in g["with_grad"]]
    np.testing.assert_allclose(float(m.final_layer[0].weight.grad.norm()), float(g["grad_norm_final"]), rtol=1e-3)
    np.testing.assert_allclose(float(m.stage3[0].branches[0][0].conv1.weight.grad.norm()), float(g["grad_norm_stage3"]), rtol=1e-3)
    assert all(not b.training for b in m.modules() if isinstance(b, torch.nn.BatchNorm2d))


def _run_lr_schedule(make_opt, device="cpu"):
    """The loop of tools/train.py:385-432 reduced to schedule + optimizer step, through cim_amd's host mirror."""
    from cases import LR_CASE, lr_toy_grads, lr_toy_model
    from cim_amd.core.config import cfg, reset_cfg
    from cim_amd.optim.solver import LRSchedule
    from cim_amd.utils import net as net_utils
    reset_cfg()
    for k in ("BASE_LR", "WARM_UP_ITERS", "WARM_UP_FACTOR", "WARM_UP_METHOD", "STEPS", "GAMMA", "MOMENTUM", "WEIGHT_DECAY"):
        cfg.SOLVER[k] = LR_CASE[k]
    m = lr_toy_model().to(device)
    opt = make_opt(m)
    sched = LRSchedule(opt)
    lrs, flat, hist = [], [], []

    def history():
        return torch.cat([opt.state[p]["momentum_buffer"].reshape(-1) for p in m.parameters()]).cpu().numpy()

    for step in range(LR_CASE["n_steps"]):
        sched.before_step(step)
        lr_toy_grads(m, step)
        opt.step()
        lrs.append([g["lr"] for g in opt.param_groups])
        flat.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu().numpy())
        hist.append(history())
    out = dict(lrs=np.array(lrs), params=np.stack(flat), history=np.stack(hist))
    net_utils.decay_learning_rate(opt, sched.lr, 0.1)
    out["decay_lrs"] = np.array([g["lr"] for g in opt.param_groups])
    out["decay_history"] = history()
    for name, clip in (("clip_small", 0.5), ("clip_large", 100.0)):
        lr_toy_grads(m, 3)
        net_utils.clip_gradient(m, clip)
        out[name] = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu().numpy()
    reset_cfg()
    return out


def test_lr_schedule_and_momentum_correction_match_reference(golden_dir):
    """cim_amd.utils.net + cim_amd.optim.solver against lib/utils/net.py driven by train.py's schedule (golden:
    learning rates of both groups bit-equal, parameters / momentum history of torch.optim.SGD bit-equal)."""
    from cim_amd.core.config import cfg
    from cim_amd.optim.solver import param_groups
    g = np.load(os.path.join(golden_dir, "lr_schedule.npz"))
    out = _run_lr_schedule(lambda m: torch.optim.SGD(param_groups(m), momentum=cfg.SOLVER.MOMENTUM))
    assert np.array_equal(out["lrs"], g["lrs"]) and np.array_equal(out["decay_lrs"], g["decay_lrs"])
    assert out["lrs"][0, 1] == 2 * out["lrs"][0, 0]                     # BIAS_DOUBLE_LR
    for k in ("params", "history", "decay_history", "clip_small", "clip_large"):
        assert np.array_equal(out[k], g[k]), k


def test_chained_bn_backward_protocol():
    """cim_amd/ops/chain.py (host side of the BatchNorm backward folded into the next layer's data gradient): a producer is only
    tagged when it is eligible (ReLU, no residual, no trainable folded bias, a differentiable output) - with TRAINABLE gamma / beta
    (the reference's configuration, lib/modeling/resnet50.py:59-60) it is, and the consumer is told to write partial sums; the
    consumer marks it as taken; the producer recognises the handed-over gradient by its storage, exactly once, and refuses anything
    else once it was taken.  The marks live in the producer's own state (no process-global set)."""
    import pytest
    import torch
    from cim_amd.ops import chain
    gamma, beta, mean, var = (torch.ones(4) for _ in range(4))
    y = torch.zeros(2, 4, 3, 3, requires_grad=True)
    st = lambda: {"taken": False, "xr": torch.zeros(2, 4, 3, 3)}
    assert chain.tag(y, gamma, beta, mean, var, 1e-5, False, False, st()) is None and not hasattr(y, "_cim_bn")     # no ReLU
    assert chain.tag(y, gamma, beta, mean, var, 1e-5, True, True, st()) is None                                       # residual
    assert chain.tag(y, gamma, beta, mean.clone().requires_grad_(), var, 1e-5, True, False, st()) is None            # trainable folded bias
    assert chain.tag(torch.zeros(3), gamma, beta, mean, var, 1e-5, True, False, st()) is None                        # nothing to differentiate
    assert chain.tag(y, gamma, beta, mean, var, 1e-5, True, False, None) is None                                      # no-grad call
    state = st()
    assert chain.tag(y, gamma, beta, mean, var, 1e-5, True, False, state) is state and y._cim_bn.state is state and not y._cim_bn.affine
    assert chain.input_bn(y, enabled=False) is None and not state["taken"]
    assert chain.input_bn(torch.zeros(3), enabled=True) is None                                             # untagged input
    ib = chain.input_bn(y, enabled=True)
    assert ib.gamma is gamma and ib.var is var and ib.eps == 1e-5 and ib.mean is mean and state["taken"]
    assert chain.c_args(None, None) == (None, None, 0.0, None, None, None)
    assert chain.c_args(ib, None)[3:] == (None, None, None) and chain.c_args(ib, None)[0] == gamma.data_ptr()
    part = torch.zeros(2, 2, 1, 4)
    assert chain.c_args(ib, part)[3:] == (state["xr"].data_ptr(), mean.data_ptr(), part.data_ptr())
    dx, other = torch.zeros(2, 4, 3, 3), torch.zeros(2, 4, 3, 3)
    chain.hand_over(ib, dx, part)
    ok, got = chain.take(dx.view(2, 4, 9), state)                # the same storage, whatever the wrapper
    assert ok and got is part
    with pytest.raises(RuntimeError, match="second consumer"):   # consumed: a second take of it is "a different gradient"
        chain.take(dx, state)
    with pytest.raises(RuntimeError, match="second consumer"):
        chain.take(other, state)
    assert chain.take(other, {"taken": False}) == (False, None) and chain.take(other, None) == (False, None)   # nobody chained: the ordinary path
    # trainable affine parameters: eligible, flagged
    y2 = torch.zeros(2, 4, 3, 3, requires_grad=True)
    s2 = st()
    chain.tag(y2, gamma.clone().requires_grad_(), beta, mean, var, 1e-5, True, False, s2)
    assert y2._cim_bn.affine
    # a mark handed to ANOTHER producer's state is not this layer's (no recycled-address confusion across layers)
    chain.hand_over(chain.input_bn(y2, True), dx)
    assert chain.take(dx, {"taken": False}) == (False, None)


def test_postponed_launches_run_once_in_order_and_are_dropped_by_a_discarding_join():
    """ops.gemm.postpone / run_postponed (MaskFuse's late weight gradients wait for the next node): closures run once, in order,
    at run_postponed() or at any join; a discarding join (start of a training forward after an aborted backward) drops them."""
    from cim_amd.ops import gemm as G
    from cim_amd.utils import engine
    calls, queued = [], []
    orig = engine.queue_callback
    engine.queue_callback = queued.append                       # (no backward pass is running here)
    try:
        G._POSTPONED.clear()
        G.postpone("dev0", lambda: calls.append(1))
        G.postpone("dev0", lambda: calls.append(2))
        assert queued == [G.run_postponed]                      # one end-of-backward fallback per batch of closures
        G.run_postponed("dev1")
        assert calls == []
        G.run_postponed("dev0")
        G.run_postponed()
        assert calls == [1, 2]
        G.postpone("dev0", lambda: calls.append(3))
        G.join_side()
        assert calls == [1, 2, 3]
        G.postpone("dev0", lambda: calls.append(4))
        G.join_side(discard=True)
        G.run_postponed()
        assert calls == [1, 2, 3]
    finally:
        engine.queue_callback = orig
        G._POSTPONED.clear()


def test_pair_tile_balancing_rules():
    """ops/pair.py: which products run their last tiles as separate split-K launches (pure host arithmetic on the 256 x 256 tile
    grid of csrc/gemm_pair.hip and the chip's 256 CUs)."""
    from cim_amd.ops import pair
    # fc1's data gradient: 4 x 196 = 784 tiles = 3 rounds + 16 -> the last 4 column tiles, 4 k-splits (8 slabs of 32 each)
    assert pair.tail_columns(1000, 50176, 1024) == (192 * 256, 4)
    assert pair.tail_columns(1200, 50176, 1024) is None          # 5 x 196 = 980: the last round is 83 % full
    assert pair.tail_columns(1000, 12544, 1024) is None          # 196 tiles: less than one round
    assert pair.tail_columns(1000, 50176 + 8, 1024) is None      # ragged last column tile: not split
    assert pair.tail_columns(1024, 256 * 64, 1024) is None       # exactly one round
    # Winograd data gradient, 121 positions of 4 x 8 tiles = 15 rounds + 32 tiles -> the last position alone
    assert pair.tail_entries(1000, 2048, 1024, 121) == (1, 4)
    assert pair.tail_entries(1000, 1024, 2048, 121) is None      # forward: 1936 = 7 rounds + 144 tiles (56 % of a round)
    assert pair.tail_entries(1200, 2048, 1024, 121) is None
    assert pair.tail_entries(300, 1024, 1024, 33) == (1, 4)
    assert pair.tail_entries(300, 1024, 1024, 32) is None        # exactly one round
    assert pair.tail_entries(300, 1024, 128, 33) is None         # K too short to split


def test_mining_reads_score_blocks_in_place():
    """modeling/heads.py: a column block of the fused score matrix is handed to the mining launch with its row stride; anything
    that is not row-contiguous fp32 is copied."""
    import torch
    from cim_amd.modeling.heads import _rows_in_place
    base = torch.arange(5 * 12, dtype=torch.float32).reshape(5, 12)
    blk = base[:, 4:8]
    t, ld = _rows_in_place(blk)
    assert ld == 12 and t.data_ptr() == blk.data_ptr() and not t.requires_grad
    t, ld = _rows_in_place(base.t())                             # column-major view: copied
    assert ld == 5 and t.is_contiguous() and torch.equal(t, base.t())
    t, ld = _rows_in_place(blk.double())
    assert ld == 4 and t.dtype == torch.float32


def test_library_sources_keep_no_state():
    """include/cim_hip.h's contract (no allocation, no sync, no global state; re-entrant), checked on the sources: no
    device allocation, no library-made events / streams, no process-wide or per-thread switches, no environment reads.  The only
    per-thread datum is the message behind cim_last_error() (csrc/common.cpp)."""
    import glob
    banned = re.compile(r"hipMalloc|hipFree|hipEventCreate|hipStreamCreate|hipDeviceSynchronize|hipStreamSynchronize|"
                        r"\bthread_local\b|\bgetenv\b|std::mutex|^\s*static\s+(?!const)[^()\n]*\[\d+\]\s*(=|;)|^\s*(static\s+)?int\s+g_[a-z_]+\s*=", re.M)
    for path in sorted(glob.glob(os.path.join(REPO, "cim_amd", "csrc", "*.hip")) + glob.glob(os.path.join(REPO, "cim_amd", "csrc", "*.cpp"))
                       + glob.glob(os.path.join(REPO, "cim_amd", "csrc", "*.h"))):
        src = open(path).read()
        if os.path.basename(path) == "common.cpp":
            src = src.replace('static thread_local char g_err[512] = "";', "")
        hits = [m.group(0).strip() for m in banned.finditer(src)]
        assert not hits, (os.path.basename(path), hits)
    header = open(os.path.join(REPO, "include", "cim_hip.h")).read()
    for gone in ("cim_gemm_set_engine", "cim_gemm_get_engine", "cim_gemm_pair_limit"):
        assert gone + "(" not in header
    assert "cim_mining_sync_bytes" in header and "fork_event" in header and "max_workgroups" in header


def test_pool_wrappers_take_the_module_on_cpu_tensors_and_parse_geometry():
    """ops/pool.py: CPU tensors run the nn module itself (host-side tests of the model code); the geometry the HIP kernels take is
    read off the module the way torch stores it (ints or pairs)."""
    import torch.nn as nn
    from cim_amd.ops import max_pool2d, upsample_nearest
    from cim_amd.ops.pool import _one
    x = torch.randn(1, 3, 9, 11)
    m = nn.MaxPool2d(3, 2, 1)
    assert torch.equal(max_pool2d(x, m), m(x))
    u = nn.Upsample(scale_factor=2, mode="nearest")
    assert torch.equal(upsample_nearest(x, u), u(x))
    assert _one(3) == 3 and _one((2, 2)) == 2 and _one((2, 3)) is None
    from cim_amd import _lib
    lib = _lib.load()
    for n, k, s, p in ((258, 3, 2, 1), (7, 3, 2, 1), (33, 2, 2, 0), (5, 5, 3, 2)):
        want = nn.MaxPool2d(k, s, p)(torch.zeros(1, 1, n, n)).shape[-1]
        assert lib.cim_maxpool2d_out_size(n, k, s, p) == want
