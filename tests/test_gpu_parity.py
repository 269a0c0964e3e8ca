"""-m gpu: the HIP path (through the C ABI of include/cim_hip.h) against the oracle and the
golden vectors captured from the reference.  Bit-exact for indices / fp16 maps / ROIAlign
forward; stated tolerances for fp32 reductions with a different summation order."""
import os

import numpy as np
import pytest
import torch

from cases import MINING_CASES, THRESHOLDS, case_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from cim_amd import _lib
    _lib.load()          # fail loudly if the HIP extension is missing
    return torch.device("cuda:0")


def _cl(x_nchw, dev):
    return torch.from_numpy(x_nchw).to(dev).contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------ ROIAlign (a-1, a-2 prologue)
def _roi_case(seed, C, H, W, K, img_scale=16.0):
    rng = np.random.RandomState(seed)
    feat = rng.randn(1, C, H, W).astype(np.float32)
    x1 = rng.uniform(-20, W * img_scale * 0.8, K)
    y1 = rng.uniform(-20, H * img_scale * 0.8, K)
    w = rng.uniform(1, W * img_scale, K)
    h = rng.uniform(1, H * img_scale, K)
    rois = np.stack([np.zeros(K), x1, y1, x1 + w, y1 + h], 1).astype(np.float32)
    rois[0, 1:] = (0, 0, W * img_scale, H * img_scale)             # full image
    rois[1, 1:] = (40, 40, 40.5, 40.25)                            # < 1 feature px
    rois[2, 1:] = (W * img_scale + 500, 10, W * img_scale + 900, 200)   # outside
    return feat, rois


@pytest.mark.parametrize("C,H,W,K,aligned", [(8, 13, 17, 24, True), (3, 9, 11, 16, True), (64, 33, 43, 40, True),
                                              (8, 13, 17, 24, False), (4, 13, 17, 24, True), (4, 13, 17, 24, False),
                                              (16, 45, 60, 33, True), (16, 45, 60, 33, False),
                                              (16, 57, 75, 33, True), (8, 100, 128, 20, True)])      # (maps beyond 64 x 64: round 6)
def test_roi_align_fwd_bwd_vs_oracle(dev, C, H, W, K, aligned, monkeypatch):
    from cim_amd.ops import roi_align
    import sys
    ra_mod = sys.modules["cim_amd.ops.roi_align"]        # (the package attribute of that name is the function)
    from oracle import roi_align as oracle
    feat, rois = _roi_case(C * 7 + K, C, H, W, K)
    ref = oracle.roi_align_fwd(feat, rois, P=7, scale=1 / 16.0, aligned=aligned)
    x = _cl(feat, dev).requires_grad_(True)
    monkeypatch.setattr(ra_mod, "EXACT", True)                             # the reference's sample order
    out = roi_align(x, torch.from_numpy(rois).to(dev), 7, 1 / 16.0, 0, "avg", aligned)
    assert out.shape == (K, C, 7, 7)
    np.testing.assert_array_equal(out.detach().cpu().numpy(), ref)         # bit-identical forward
    monkeypatch.setattr(ra_mod, "EXACT", False)                            # default: aggregated weights (reassociated sum)
    out = roi_align(x, torch.from_numpy(rois).to(dev), 7, 1 / 16.0, 0, "avg", aligned)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=0, atol=1e-6 * float(np.abs(feat).max()))
    rng = np.random.RandomState(1)
    go = rng.randn(K, C, 7, 7).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    gref = oracle.roi_align_bwd(go, rois, feat.shape, P=7, scale=1 / 16.0, aligned=aligned)
    # fp32 atomics in arbitrary order vs exact fp64 accumulation
    np.testing.assert_allclose(x.grad.cpu().numpy(), gref, rtol=1e-4, atol=1e-4)


def test_roi_align_module_matches_reference_call_convention(dev):
    """model_builder.py:230-231: RoIAlign(resolution, spatial_scale, sampling_ratio)(feat, rois)."""
    from cim_amd.ops import RoIAlign
    from oracle import roi_align as oracle
    feat, rois = _roi_case(5, 16, 12, 15, 10)
    tol = dict(rtol=0, atol=1e-6 * float(np.abs(feat).max()))
    out = RoIAlign(7, 1.0 / 16.0, 0)(torch.from_numpy(feat).to(dev).contiguous(), torch.from_numpy(rois).to(dev).contiguous())
    np.testing.assert_allclose(out.cpu().numpy(), oracle.roi_align_fwd(feat, rois), **tol)
    out2 = RoIAlign(7, 1.0 / 16.0, 2)(torch.from_numpy(feat).to(dev), torch.from_numpy(rois).to(dev))
    np.testing.assert_allclose(out2.cpu().numpy(), oracle.roi_align_fwd(feat, rois, sampling_ratio=2), **tol)


def test_roi_align_maskcat_fused_vs_oracle(dev, monkeypatch):
    from cim_amd.ops import roi_align_maskcat
    import sys
    ra_mod = sys.modules["cim_amd.ops.roi_align"]        # (the package attribute of that name is the function)
    from oracle import roi_align as oracle
    C, H, W, K = 32, 20, 25, 30
    feat, rois = _roi_case(77, C, H, W, K)
    rng = np.random.RandomState(2)
    masks = (rng.rand(K, 7, 7) > 0.4).astype(np.float32)
    box = oracle.roi_align_fwd(feat, rois)
    ref = np.concatenate([box, box * masks[:, None]], axis=1)
    x = _cl(feat, dev).requires_grad_(True)
    monkeypatch.setattr(ra_mod, "EXACT", True)
    cat = roi_align_maskcat(x, torch.from_numpy(rois).to(dev), torch.from_numpy(masks).to(dev), 7, 1 / 16.0, 0, True)
    assert cat.shape == (K, 2 * C, 7, 7)
    np.testing.assert_array_equal(cat.detach().cpu().numpy(), ref)
    monkeypatch.setattr(ra_mod, "EXACT", False)
    cat = roi_align_maskcat(x, torch.from_numpy(rois).to(dev), torch.from_numpy(masks).to(dev), 7, 1 / 16.0, 0, True)
    np.testing.assert_allclose(cat.detach().cpu().numpy(), ref, rtol=0, atol=1e-6 * float(np.abs(feat).max()))
    g = rng.randn(K, 2 * C, 7, 7).astype(np.float32)
    cat.backward(torch.from_numpy(g).to(dev))
    g_box = g[:, :C] + g[:, C:] * masks[:, None]
    gref = oracle.roi_align_bwd(g_box, rois, feat.shape)
    np.testing.assert_allclose(x.grad.cpu().numpy(), gref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,C,H,W,K", [(2, 8, 13, 17, 300), (3, 1028, 9, 11, 40), (1, 2048, 17, 22, 70)])
def test_roi_align_batches_groups_and_channel_chunks(dev, B, C, H, W, K):
    """Table-driven forward / region backward beyond the benchmark shape: several images per call, several ROI groups,
    channel counts past one 1024-lane chunk and not a multiple of the 256-channel slice."""
    from cim_amd.ops import roi_align
    from oracle import roi_align as oracle
    feat0, rois = _roi_case(B * 31 + K, C, H, W, K)
    rng = np.random.RandomState(K)
    feat = rng.randn(B, C, H, W).astype(np.float32)
    rois[:, 0] = rng.randint(0, B, K)
    ref = oracle.roi_align_fwd(feat, rois)
    x = _cl(feat, dev).requires_grad_(True)
    out = roi_align(x, torch.from_numpy(rois).to(dev), 7, 1 / 16.0)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref, rtol=0, atol=1e-6 * float(np.abs(feat).max()))
    go = rng.randn(K, C, 7, 7).astype(np.float32)
    out.backward(torch.from_numpy(go).to(dev))
    gref = oracle.roi_align_bwd(go, rois, feat.shape)
    np.testing.assert_allclose(x.grad.cpu().numpy(), gref, rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("B,C,H,W,K", [(2, 8, 13, 17, 300), (1, 516, 33, 43, 200), (1, 512, 37, 49, 1900)])
def test_roi_align_backward_forms_agree_with_oracle(dev, B, C, H, W, K):
    """The backward behind cim_roi_align(_maskcat)_bwd_ws - region form with partial maps (interleaved entry order) - and the
    entry points WITHOUT partial-map scratch (cim_roi_align_maskcat_bwd: the ROI groups then meet through atomics) against the
    fp64 oracle, with several images, several ROI groups (128-ROI groups in the last case: 30 x 16 x 2 = 960 workgroups of 64 ROIs
    are more than the launcher's limit of 900) and a channel count that is not a multiple of the slice."""
    from cim_amd import _lib
    from cim_amd.ops import roi_align_maskcat
    from oracle import roi_align as oracle
    _, rois = _roi_case(B * 31 + K, C, H, W, K)
    rng = np.random.RandomState(K)
    feat = rng.randn(B, C, H, W).astype(np.float32)
    rois[:, 0] = rng.randint(0, B, K)
    masks = (rng.rand(K, 7, 7) > 0.4).astype(np.float32)
    g = rng.randn(K, 2 * C, 7, 7).astype(np.float32)
    gref = oracle.roi_align_bwd(g[:, :C] + g[:, C:] * masks[:, None], rois, feat.shape)
    tol = dict(rtol=1e-4, atol=2e-4 * max(1.0, K / 300.0))
    x = _cl(feat, dev).requires_grad_(True)
    r_d, m_d = torch.from_numpy(rois).to(dev), torch.from_numpy(masks).to(dev)
    cat = roi_align_maskcat(x, r_d, m_d, 7, 1 / 16.0, 0, True)
    cat.backward(torch.from_numpy(g).to(dev))
    np.testing.assert_allclose(x.grad.cpu().numpy(), gref, err_msg="region form, partial maps", **tol)
    # the same gradient through the C entry point without scratch
    g_nhwc = torch.from_numpy(g).to(dev).permute(0, 2, 3, 1).contiguous()
    gin = torch.empty(B, H, W, C, device=dev)
    ws = torch.empty(_lib.call("cim_roi_align_bwd_workspace", K, 7, H, W) // 4 + 1, device=dev)
    _lib.call("cim_roi_align_maskcat_bwd", g_nhwc.data_ptr(), r_d.data_ptr(), m_d.data_ptr(), gin.data_ptr(), B, C, H, W, K, 7,
              1 / 16.0, 0, 1, ws.data_ptr(), _lib.stream_ptr())
    np.testing.assert_allclose(gin.permute(0, 3, 1, 2).cpu().numpy(), gref, err_msg="no scratch: groups meet through atomics", **tol)


def test_roi_align_empty_and_bad_args(dev):
    from cim_amd import _lib
    from cim_amd.ops import roi_align
    feat = torch.randn(1, 8, 5, 5, device=dev).contiguous(memory_format=torch.channels_last)
    out = roi_align(feat, torch.zeros(0, 5, device=dev), 7, 1 / 16.0)
    assert out.shape == (0, 8, 7, 7)
    with pytest.raises(_lib.CimHipError):
        _lib.call("cim_roi_align_fwd", feat.data_ptr(), None, None, 1, 8, 5, 5, 3, 7, 0.0625, 0, 1, None)
    with pytest.raises(_lib.CimHipError):
        roi_align(torch.randn(1, 8, 5, 5), torch.zeros(1, 5), 7, 1 / 16.0)     # CPU tensors: no fallback


def test_roi_align_full_size_properties(dev):
    """BASELINE cfg2 size (1024 x 33 x 43, 1000 ROIs): linearity and the constant-map identity."""
    from cim_amd.ops import roi_align
    from cim_amd import synthetic
    inp = synthetic.make_image_inputs("resnet50_voc", seed=3, with_image=False)
    rois = torch.from_numpy(inp["rois"]).to(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    a = torch.randn(1, 1024, 33, 43, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn(1, 1024, 33, 43, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    fa, fb, fab = (roi_align(t, rois, 7, 1 / 16.0) for t in (a, b, 2 * a + b))
    assert fa.shape == (1000, 1024, 7, 7)
    torch.testing.assert_close(fab, 2 * fa + fb, rtol=1e-4, atol=1e-4)
    ones = roi_align(torch.ones_like(a), rois, 7, 1 / 16.0)
    inside = (ones - 1).abs().max()
    assert inside < 1e-5 or ones.min() >= 0         # bins partly outside the map average in zeros
    assert torch.isfinite(fa).all()


# ------------------------------------------------------------------ mask IoU maps (a-7)
@pytest.mark.parametrize("n", [6, 64, "witness"])
def test_mask_iou_matches_reference_golden(dev, n, golden_dir):
    from cim_amd import mask_iou
    g = np.load(os.path.join(golden_dir, "mask_iou_%s.npz" % n))
    n = g["iou"].shape[0]
    h, w = int(g["h"]), int(g["w"])
    masks = np.unpackbits(g["masks_packed"], axis=1)[:, :h * w].reshape(n, h, w).astype(bool)
    iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(masks).to(dev))
    assert iou.dtype == torch.float16
    np.testing.assert_array_equal(iou.cpu().numpy(), g["iou"])
    np.testing.assert_array_equal(asy.cpu().numpy(), g["asy"])


def test_mask_iou_vs_oracle_ragged_and_full_size(dev):
    from cim_amd import mask_iou, synthetic
    from oracle import mask_iou as oracle
    rng = np.random.RandomState(4)
    masks, _ = synthetic.make_masks(301, 67, 93, rng, min_side=4)      # N, H*W not multiples of 64
    iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(masks).to(dev))
    ri, ra = oracle.mask_iou_maps(masks)
    np.testing.assert_array_equal(iou.cpu().numpy(), ri)
    np.testing.assert_array_equal(asy.cpu().numpy(), ra)
    # cfg2 size: 1000 masks of 375 x 500
    inp = synthetic.make_image_inputs("resnet50_voc", seed=3, with_image=False)
    iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(dev))
    ri, ra = oracle.mask_iou_maps(inp["full_masks"])
    np.testing.assert_array_equal(iou.cpu().numpy(), ri)
    np.testing.assert_array_equal(asy.cpu().numpy(), ra)
    assert torch.equal(iou, iou.t()) and bool((iou.diagonal() == 1).all()) and bool((asy.diagonal() == 1).all())


# ------------------------------------------------------------------ mining + assignment (a-4..a-6)
def _run_layer(dev, g, prefix, cls, det, labels, iou, asy, cls_thr, iou_thr, seed, sampling=True, using_cim=True):
    from cim_amd.modeling import heads
    layer = heads.CIM_layer(p_seed=0.1, cls_thr=cls_thr, iou_thr=iou_thr, Anti_noise_sampling=sampling)
    np.random.seed(seed)
    t = lambda a: torch.from_numpy(a).to(dev)
    out = layer(t(cls), t(det), torch.zeros(cls.shape[0], 5, device=dev), t(labels), t(iou), t(asy), using_CIM=using_cim)
    probe = np.random.random_sample()
    assert bool(g[prefix + "is_none"]) == (out[0] is None)
    assert probe == float(g[prefix + "rng_probe"]), "NumPy RNG stream position differs from the reference"
    last = layer.last
    order = g[prefix + "class_order"]
    K = int(np.ceil(0.1 * cls.shape[0]))
    for k, c in enumerate(order):
        np.testing.assert_array_equal(last["topk"][k].cpu().numpy(), g["%sc%d_keep_sort_idx" % (prefix, c)])
        ns = int(last["n_seeds"][k])
        np.testing.assert_array_equal(last["seeds"][k, :ns].cpu().numpy(), g["%sc%d_keep_nms_idx" % (prefix, c)])
        assert (last["seeds"][k, ns:] == -1).all()
        if using_cim:
            res = last["res"][k, :ns].cpu().numpy()
            key = "%sc%d_res_idx" % (prefix, c)
            got = np.unique(res[res >= 0])
            if key in g.files:
                np.testing.assert_array_equal(got, g[key])
            else:
                assert got.size == 0
    gt_idxs = (last["gt_class"] > 0).cpu().numpy()
    np.testing.assert_array_equal(gt_idxs, g[prefix + "label_gt_idxs"])
    gl = g[prefix + "label_gt_labels"]
    np.testing.assert_array_equal(last["gt_class"].cpu().numpy()[gt_idxs], gl.argmax(1) if gl.size else np.zeros(0))
    np.testing.assert_array_equal(last["gt_weight"].cpu().numpy()[gt_idxs], g[prefix + "label_gt_weights"])
    if using_cim:
        np.testing.assert_array_equal(last["asy_flag"].cpu().numpy().astype(bool)[:, None], g[prefix + "asy_iou_flag"])
    if out[0] is None:
        return
    if sampling:
        np.testing.assert_array_equal(last["sample_keep"], g[prefix + "sample_keep"])
    np.testing.assert_array_equal(last["max_overlap_idx"].cpu().numpy(), g[prefix + "max_overlap_idx"])
    np.testing.assert_array_equal(out[0].cpu().numpy(), g[prefix + "pseudo_labels"])
    assert out[1].dtype == torch.float16
    np.testing.assert_array_equal(out[1].cpu().numpy(), g[prefix + "pseudo_iou_labels"])
    np.testing.assert_array_equal(out[2].cpu().numpy(), g[prefix + "loss_weights"])


@pytest.mark.parametrize("name", list(MINING_CASES))
def test_cim_layer_bit_identical_to_reference(dev, name, golden_dir):
    g = np.load(os.path.join(golden_dir, "mining_%s.npz" % name))
    inp = case_inputs(MINING_CASES[name])
    for li, (cls_thr, iou_thr) in enumerate(THRESHOLDS):
        cls, det, _ = inp["layers"][li]
        _run_layer(dev, g, "l%d_" % li, cls, det, inp["labels"], inp["iou"], inp["asy"], cls_thr, iou_thr, 100 + li)
    cls, det, _ = inp["layers"][0]
    _run_layer(dev, g, "nosample_", cls, det, inp["labels"], inp["iou"], inp["asy"], 0.25, 0.5, 7, sampling=False)
    _run_layer(dev, g, "mist_", cls, det, inp["labels"], inp["iou"], inp["asy"], 0.25, 0.5, 8, using_cim=False)


def test_cim_layer_degenerate_cases(dev, golden_dir):
    from cim_amd.modeling import heads
    g = np.load(os.path.join(golden_dir, "mining_degenerate.npz"))
    _run_layer(dev, g, "huge_", g["in_cls"], g["in_det"], g["labels"], g["in_iou"], g["in_asy"], 0.25, 0.5, 9)
    layer = heads.CIM_layer()
    t = lambda a: torch.from_numpy(a).to(dev)
    out = layer(t(g["in_cls"]), t(g["in_det"]), None, t(np.zeros_like(g["labels"])), t(g["in_iou"]), t(g["in_asy"]))
    assert out == (None, None, None)
    with pytest.raises(NotImplementedError):                      # missing maps: model_builder.py:150-152
        layer(t(g["in_cls"]), t(g["in_det"]), None, t(g["labels"]), None, None)


def test_cim_layer_full_size_vs_oracle(dev):
    """cfg2 / cfg4 sizes against the NumPy oracle on the same seeded inputs (index-exact)."""
    from cim_amd import synthetic
    from cim_amd.modeling import heads
    from oracle import mask_iou as omi, mining as om
    for config, n in (("resnet50_voc", 1000), ("resnet50_coco2017", 2000)):
        inp = synthetic.make_image_inputs(config, seed=3, with_image=False)
        iou, asy = omi.mask_iou_maps(inp["full_masks"])
        C = inp["labels"].shape[1]
        rng = np.random.RandomState(0)
        t = lambda a: torch.from_numpy(a).to(dev)
        for li, (cls_thr, iou_thr) in enumerate(THRESHOLDS):
            cls, det, _ = synthetic.make_scores(n, C, rng)
            np.random.seed(50 + li)
            ref = om.cim_layer_forward(cls, det, inp["labels"], iou, asy, cls_thr=cls_thr, iou_thr=iou_thr)
            probe_ref = np.random.random_sample()
            np.random.seed(50 + li)
            layer = heads.CIM_layer(cls_thr=cls_thr, iou_thr=iou_thr)
            out = layer(t(cls), t(det), None, t(inp["labels"]), t(iou), t(asy))
            assert np.random.random_sample() == probe_ref
            assert (out[0] is None) == (ref[0] is None)
            if ref[0] is not None:
                np.testing.assert_array_equal(out[0].cpu().numpy(), ref[0])
                np.testing.assert_array_equal(out[1].cpu().numpy(), ref[1])
                np.testing.assert_array_equal(out[2].cpu().numpy(), ref[2])


# ------------------------------------------------------------------ heads + losses (a-3, a-8..a-10)
@pytest.mark.parametrize("name", ["n300_c20_k2", "n1000_c80_k3"])
def test_losses_on_device_match_reference(dev, name, golden_dir):
    from cim_amd.modeling import heads
    g = np.load(os.path.join(golden_dir, "losses_%s.npz" % name))
    m = np.load(os.path.join(golden_dir, "mining_%s.npz" % name))
    inp = case_inputs(MINING_CASES[name])
    t = lambda a: torch.from_numpy(a).to(dev)
    # against the reference's own fp32 results (its fp64 results differ by up to 1e-3 where the
    # 1e-6 clamp saturates: 1 - 1e-6 is not representable in fp32); fp32 reductions run in a
    # different order on the device, hence 2e-5.
    rtol = 2e-5
    for li in range(3):
        lmda = 3 if li == 0 else 1
        cls, _, iou = inp["layers"][li]
        got = heads.cls_iou_loss(t(cls), t(iou), t(m["l%d_pseudo_labels" % li]), t(m["l%d_pseudo_iou_labels" % li]),
                                 lmda * t(m["l%d_loss_weights" % li]), t(inp["labels"]))
        np.testing.assert_allclose([float(x) for x in got], g["f32_l%d_cls_iou_bag" % li], rtol=rtol)
    cls, det, _ = inp["layers"][0]
    np.testing.assert_allclose(float(heads.mil_bag_loss(t(cls), t(det), t(inp["labels"]))), g["f32_mil_bag"], rtol=rtol)
    np.testing.assert_allclose(float(heads.PCL_loss(t(cls), t(inp["mat"]), t(inp["labels"]))), g["f32_pcl"], rtol=rtol)


# ------------------------------------------------------------------ end to end (a-11, a-12): model vs oracle/cpu_step.py
@pytest.mark.parametrize("config", ["resnet50_voc", "vgg16_voc", "hrnet48_voc"])
def test_training_step_matches_cpu_oracle(dev, config):
    """Same weights, same synthetic image: the 4 losses and the parameter gradients of the HIP
    model against the host restatement (torch CPU ops + oracle ROIAlign + oracle mining).
    fp32 conv / GEMM summation order differs between the HIP kernels and ATen's CPU kernels,
    hence a tolerance; the pseudo labels are index-exact, so the same loss terms are active."""
    import copy
    from cim_amd import mask_iou, synthetic
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling.model_builder import Generalized_RCNN
    from oracle import cpu_step, mask_iou as omi
    apply_preset(config)
    torch.manual_seed(0)
    n = 48
    inp = synthetic.make_image_inputs(config, seed=5, n=n)
    inp["data"] = inp["data"][:, :, :192, :256].copy()
    inp["rois"][:, 1:] *= np.float32(0.35)
    cpu_model = Generalized_RCNN().train()
    gpu_model = copy.deepcopy(cpu_model).to(dev).train()
    iou, asy = omi.mask_iou_maps(inp["full_masks"])
    ref = cpu_step.step(cpu_model, inp, iou, asy, seed=11)
    t = lambda a: torch.from_numpy(a).unsqueeze(0).to(dev)
    np.random.seed(11)
    diou, dasy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(dev))
    out = gpu_model(data=torch.from_numpy(inp["data"]).to(dev), rois=t(inp["rois"]), masks=t(inp["masks"]),
                    labels=t(inp["labels"]), gtrois=None, mat=t(inp["mat"]), index=t(inp["index"]),
                    iou_map=diou, asy_iou_map=dasy)
    sum(v.sum() for v in out["losses"].values()).backward()
    for k, v in ref.items():
        assert out["losses"][k].shape == (1,)
        np.testing.assert_allclose(float(out["losses"][k]), v, rtol=1e-4, atol=1e-6, err_msg=k)
    from cases import gradient_deviation
    cpu_p = dict(cpu_model.named_parameters())
    got_g, ref_g = {}, {}
    for name, p in gpu_model.named_parameters():
        if not p.requires_grad:
            assert cpu_p[name].grad is None
            continue
        g_ref = cpu_p[name].grad
        if p.grad is None or g_ref is None:       # e.g. HRNet's unused `classifier`
            assert p.grad is None and g_ref is None, name
            continue
        got_g[name], ref_g[name] = p.grad, g_ref
    # every gradient within 2e-3 of its own norm; gradients that cancel to nothing by an absolute bound (cases.py).  (The full-size
    # runs of tests/test_gpu_fullsize.py hold 5e-4, measured 1.8e-4; on these SMALL images a BatchNorm scale / shift gradient - a
    # signed sum over few pixels that cancels to a small remainder - measures 5.3e-4 (ResNet-50 res3) and 1.1e-3 (an HRNet fuse
    # layer at 1/32 resolution) against ATen's CPU summation order; VGG16, without BatchNorm: 8e-7.)
    worst, where, _ = gradient_deviation(got_g, ref_g, rtol=2e-3)
    print("%s: worst gradient deviation %.3g at %s" % (config, worst, where))
    assert len(got_g) > 20


@pytest.mark.parametrize("config", ["resnet50_voc", "vgg16_voc"])
def test_backbone_hip_graph_replay_matches_eager(dev, config, monkeypatch):
    """From the 3rd step on a recurring image shape the backbone runs as captured HIP graphs (forward and backward):
    same kernels, so the losses and every parameter gradient must equal the eager steps'; a different shape in
    between stays eager and does not disturb the cached graph."""
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    from cim_amd import mask_iou, synthetic
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling import model_builder
    apply_preset(config)
    monkeypatch.setattr(model_builder, "GRAPH_BACKBONE", True)      # opt-in feature (CIM_GRAPH_BACKBONE=1)
    torch.manual_seed(0)
    inp = synthetic.make_image_inputs(config, seed=5, n=32)
    inp["data"] = inp["data"][:, :, :160, :224].copy()
    inp["rois"][:, 1:] *= np.float32(0.3)
    model = model_builder.Generalized_RCNN().to(dev).train()
    t = lambda a: torch.from_numpy(a).unsqueeze(0).to(dev)
    diou, dasy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(dev))
    data = torch.from_numpy(inp["data"]).to(dev)

    def step(d):
        model.zero_grad(set_to_none=True)
        np.random.seed(11)
        out = model(data=d, rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]), gtrois=None,
                    mat=t(inp["mat"]), index=t(inp["index"]), iou_map=diou, asy_iou_map=dasy)
        sum(v.sum() for v in out["losses"].values()).backward()
        torch.cuda.synchronize()
        return ({k: float(v) for k, v in out["losses"].items()}, out["blob_conv"].detach().clone(),
                {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})

    eager = step(data)
    step(data)
    step(data[:, :, :128, :192].contiguous())                       # another shape in between: eager
    graphed = step(data)                                             # 3rd occurrence of the shape: captured + replayed
    replay = step(data)
    st = model.__dict__["_graphed_bodies"]
    assert any(g not in (None, False) for g in st["graphs"].values()), "backbone was not captured"
    for got in (graphed, replay):
        # (split-K workspaces differ under capture: fp32 summation-order noise)
        torch.testing.assert_close(got[1], eager[1], rtol=1e-4, atol=1e-4)
        for k in eager[0]:
            np.testing.assert_allclose(got[0][k], eager[0][k], rtol=1e-4, atol=1e-6)
        assert got[2].keys() == eager[2].keys()
        for n in eager[2]:
            d, ref = (got[2][n] - eager[2][n]).norm(), eager[2][n].norm()
            assert float(d) <= 2e-3 * float(ref) + 1e-7 * eager[2][n].numel() ** 0.5, n
    # An optimizer step BETWEEN replays: the captured backward must see the new weights (a prefetched weight transpose baked
    # into the graph would keep the capture-time weights: ADVICE round 4, ops/conv3x3.py: _transposed).
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.grad is not None:
                p.add_(p.grad, alpha=-0.05 / (float(p.grad.abs().max()) + 1e-12) * float(p.abs().max()))
    after = step(data)                                               # replay on the updated weights
    monkeypatch.setattr(model_builder, "GRAPH_BACKBONE", False)
    ref_after = step(data)                                           # eager, same weights
    assert any(float((after[2][n] - replay[2][n]).norm()) > 1e-3 * float(replay[2][n].norm()) for n in after[2]), \
        "the update did not change the gradients: vacuous check"
    torch.testing.assert_close(after[1], ref_after[1], rtol=1e-4, atol=1e-4)
    for n in ref_after[2]:
        d, ref = (after[2][n] - ref_after[2][n]).norm(), ref_after[2][n].norm()
        assert float(d) <= 2e-3 * float(ref) + 1e-7 * ref_after[2][n].numel() ** 0.5, n


# ------------------------------------------------------------------ fused HIP losses (csrc/losses.hip)
@pytest.mark.parametrize("name", ["n300_c20_k2", "n1000_c80_k3"])
def test_fused_losses_match_reference_and_autograd(dev, name, golden_dir):
    """One-launch HIP losses vs (a) the reference's fp32 values (goldens) and (b) the ATen formulation's
    autograd gradients, including a skipped refinement layer and non-unit upstream gradients."""
    from cim_amd.modeling import heads
    g = np.load(os.path.join(golden_dir, "losses_%s.npz" % name))
    m = np.load(os.path.join(golden_dir, "mining_%s.npz" % name))
    inp = case_inputs(MINING_CASES[name])
    t = lambda a: torch.from_numpy(a).to(dev)
    labels = t(inp["labels"])
    mat = t(inp["mat"])
    for valid in ([True, True, True], [True, False, True]):
        pc = t(inp["layers"][0][0]).requires_grad_(True)
        pd = t(inp["layers"][0][1]).requires_grad_(True)
        rc = [t(inp["layers"][i][0]).requires_grad_(True) for i in range(3)]
        ri = [t(inp["layers"][i][2]).requires_grad_(True) for i in range(3)]
        pseudo = [(t(m["l%d_pseudo_labels" % i]), t(m["l%d_pseudo_iou_labels" % i]), t(m["l%d_loss_weights" % i]))
                  if valid[i] else None for i in range(3)]
        scales = [3, 1, 1]
        bag, pcl, cls_l, iou_l = heads.fused_losses(pc, pd, rc, ri, labels, pseudo, scales, mat)
        up = torch.tensor([0.7, 1.3, 2.0, 0.5], device=dev)
        (up[0] * bag + up[1] * pcl + up[2] * cls_l + up[3] * iou_l).backward()
        got = [x.grad.clone() for x in [pc, pd] + rc + ri]
        for x in [pc, pd] + rc + ri:
            x.grad = None
        # ATen formulation (cim_amd.modeling.heads functions mirror the reference one to one)
        rb = rcl = rio = 0
        for i in range(3):
            if pseudo[i] is None:
                continue
            c, io, b = heads.cls_iou_loss(rc[i], ri[i], pseudo[i][0], pseudo[i][1], scales[i] * pseudo[i][2], labels)
            rb, rcl, rio = rb + b, rcl + c, rio + io
        rb = rb + heads.mil_bag_loss(pc, pd, labels)
        rp = heads.PCL_loss(pc, t(inp["mat"]), labels)
        (up[0] * rb + up[1] * rp + up[2] * rcl + up[3] * rio).backward()
        for a, b_ in zip((bag, pcl, cls_l, iou_l), (rb, rp, rcl, rio)):
            np.testing.assert_allclose(float(a), float(b_), rtol=2e-5, atol=1e-7)
        for x, gg in zip([pc, pd] + rc + ri, got):
            ref = x.grad if x.grad is not None else torch.zeros_like(x)
            scale = float(ref.abs().max()) + 1e-12
            assert float((gg - ref).abs().max()) <= 2e-5 * scale + 1e-9
        if all(valid):   # and against the reference's own numbers
            want = [g["f32_l%d_cls_iou_bag" % i] for i in range(3)]
            np.testing.assert_allclose(float(cls_l), sum(w[0] for w in want), rtol=2e-5)
            np.testing.assert_allclose(float(iou_l), sum(w[1] for w in want), rtol=2e-5)
            np.testing.assert_allclose(float(bag), sum(w[2] for w in want) + float(g["f32_mil_bag"]), rtol=2e-5)
            np.testing.assert_allclose(float(pcl), float(g["f32_pcl"]), rtol=2e-5)


# ------------------------------------------------------------------ model boundary (a-12): pickle path, eval branch
def test_model_pickle_path_and_eval_branch(dev, tmp_path):
    """forward(path=...) reads <iou_dir>/<stem>.pkl / <asy_iou_dir>/<stem>.pkl like model_builder.py:147-159
    and gives the same losses as passing the maps; eval mode returns refine_score (model_builder.py:60-68)."""
    import pickle
    from cim_amd import synthetic
    from cim_amd.core.config import cfg
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling.model_builder import Generalized_RCNN
    from oracle import mask_iou as omi
    apply_preset("resnet50_voc")
    torch.manual_seed(1)
    inp = synthetic.make_image_inputs("resnet50_voc", seed=9, n=40)
    inp["data"] = inp["data"][:, :, :160, :224].copy()
    inp["rois"][:, 1:] *= np.float32(0.3)
    iou, asy = omi.mask_iou_maps(inp["full_masks"])
    (tmp_path / "iou").mkdir()
    (tmp_path / "asy").mkdir()
    pickle.dump(iou, open(tmp_path / "iou" / "2008_000123.pkl", "wb"))
    pickle.dump(asy, open(tmp_path / "asy" / "2008_000123.pkl", "wb"))
    cfg.iou_dir, cfg.asy_iou_dir = str(tmp_path / "iou"), str(tmp_path / "asy")
    model = Generalized_RCNN().to(dev).train()
    t = lambda a: torch.from_numpy(a).unsqueeze(0)             # CPU tensors, as the loader hands them over
    kw = dict(data=torch.from_numpy(inp["data"]).to(dev), rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]),
              gtrois=None, mat=t(inp["mat"]), index=t(inp["index"]))
    np.random.seed(3)
    a = model(path="/data/VOC2012/JPEGImages/2008_000123.jpg", **kw)
    from cim_amd.modeling import heads
    heads.settle_rng()                                        # no backward here: settle the generator by hand
    np.random.seed(3)
    b = model(iou_map=torch.from_numpy(iou).to(dev), asy_iou_map=torch.from_numpy(asy).to(dev), **kw)
    heads.settle_rng()
    for k in ("bag_loss", "pcl_loss", "cls_loss", "iou_loss"):
        assert a["losses"][k].shape == (1,)
        assert float(a["losses"][k]) == float(b["losses"][k])      # same maps, deterministic kernels: bit-equal
    with pytest.raises(NotImplementedError):                  # missing pickle: model_builder.py:150-152
        model(path="/x/missing.jpg", **kw)
    with pytest.raises(ValueError):                           # check_inference: model_builder.py:45-58
        model.convbody_net(kw["data"])
    model.eval()
    out = model(data=kw["data"], rois=torch.from_numpy(inp["rois"]).to(dev), masks=torch.from_numpy(inp["masks"]).to(dev),
                labels=None, gtrois=None, mat=None)
    assert set(out) == {"blob_conv", "refine_score"} and len(out["refine_score"]) == 3
    assert out["refine_score"][0].shape == (40, 20) and not out["refine_score"][0].requires_grad
    assert model.convbody_net(kw["data"]).shape[1] == 1024


def test_head_activations_hip_vs_reference_and_autograd(dev, golden_dir):
    """cls_iou_model on the device (fused HIP epilogue) vs the reference's outputs (heads_small.npz) and vs
    ATen autograd for the backward."""
    import torch.nn.functional as F
    from cases import procedural
    from cim_amd.modeling import heads
    g = np.load(os.path.join(golden_dir, "heads_small.npz"))
    model = heads.cls_iou_model(64, 21, 3)
    with torch.no_grad():
        for k, (_, p) in enumerate(model.named_parameters()):
            p.copy_(torch.from_numpy(procedural(tuple(p.shape), k + 1)))
    model = model.to(dev)
    x = torch.from_numpy(procedural((50, 64), 99) * 20).to(dev)
    pc, pd, rc, ri = model(x)
    np.testing.assert_allclose(pc.detach().cpu().numpy(), g["predict_cls"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pd.detach().cpu().numpy(), g["predict_det"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(torch.stack(rc).detach().cpu().numpy(), g["refine_cls"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(torch.stack(ri).detach().cpu().numpy(), g["refine_iou"], rtol=1e-5, atol=1e-7)
    # backward vs ATen on random logits, N and C1 not multiples of 64
    torch.manual_seed(0)
    for n, c1, r in ((301, 21, 3), (77, 81, 3), (5, 3, 1)):
        lg = (torch.randn(n, (2 + 2 * r) * c1, device=dev) * 3).requires_grad_(True)
        up = torch.randn(n, (2 + 2 * r) * c1, device=dev)
        s = heads.HeadActFunction.apply(lg, c1, r)
        (s * up).sum().backward()
        got = lg.grad.clone()
        lg.grad = None
        parts = lg.split(c1, dim=1)
        ref = torch.cat([F.softmax(parts[0], -1), F.softmax(parts[1], 0)] + [F.softmax(p, -1) for p in parts[2:2 + r]]
                        + [torch.sigmoid(p) for p in parts[2 + r:]], dim=1)
        (ref * up).sum().backward()
        torch.testing.assert_close(s, ref, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(got, lg.grad, rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------ backbone BN chains (csrc/bn_act.hip)
@pytest.mark.parametrize("relu,with_res,affine_grad", [(True, False, True), (True, True, True), (False, False, True),
                                                        (True, True, False)])
def test_bn_act_matches_aten(dev, relu, with_res, affine_grad):
    """relu?(bn_eval(x) + residual) fused, forward and backward, against the ATen formulation in fp64."""
    import torch.nn.functional as F
    from cim_amd.ops import bn_act
    g = torch.Generator().manual_seed(7)
    N, C, H, W = 2, 24, 13, 17                                       # odd plane size: no 16-byte alignment
    bn = torch.nn.BatchNorm2d(C).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.randn(C, generator=g)); bn.bias.copy_(torch.randn(C, generator=g))
        bn.running_mean.copy_(torch.randn(C, generator=g)); bn.running_var.copy_(torch.rand(C, generator=g) + 0.3)
    if not affine_grad:
        for p in bn.parameters():
            p.requires_grad = False
    x = torch.randn(N, C, H, W, generator=g)
    r = torch.randn(N, C, H, W, generator=g) if with_res else None
    go = torch.randn(N, C, H, W, generator=g)
    import copy
    bn64 = copy.deepcopy(bn).double()
    x64 = x.double().requires_grad_(True)
    r64 = r.double().requires_grad_(True) if with_res else None
    out = bn64(x64) + (r64 if with_res else 0)
    out = F.relu(out) if relu else out
    out.backward(go.double())
    bnd = copy.deepcopy(bn).to(dev)
    xd = x.to(dev).requires_grad_(True)
    rd = r.to(dev).requires_grad_(True) if with_res else None
    y = bn_act(xd, bnd, residual=rd, relu=relu)
    y.backward(go.to(dev))
    tol = dict(rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(y.detach().cpu().double(), out.detach(), **tol)
    torch.testing.assert_close(xd.grad.cpu().double(), x64.grad, **tol)
    if with_res:
        torch.testing.assert_close(rd.grad.cpu().double(), r64.grad, **tol)
    if affine_grad:
        torch.testing.assert_close(bnd.weight.grad.cpu().double(), bn64.weight.grad, rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(bnd.bias.grad.cpu().double(), bn64.bias.grad, rtol=1e-4, atol=1e-4)
    else:
        assert bnd.weight.grad is None and bnd.bias.grad is None
    bnd.train()                                                      # training-mode BN: ATen path, batch statistics (a counted fallback)
    from cim_amd.ops import fallback
    with fallback.allowed("bn_act"):
        y2 = bn_act(x.to(dev), bnd, relu=relu)
    ref2 = copy.deepcopy(bn).train()(x)
    torch.testing.assert_close(y2.detach().cpu(), (F.relu(ref2) if relu else ref2).detach(), rtol=1e-4, atol=1e-4)


def test_stream_scheduling_does_not_change_a_training_run(dev, monkeypatch):
    """The step's scheduling options - training forward / backward on a high-priority stream, MaskFuse's and the backbone's weight
    gradients left on the side stream until the end of the backward pass, MaskFuse's launched late in chunks of 256 workgroups,
    the big weights' update on the side stream under the next forward (optim.SGD.overlap_update) -
    must not change a single bit of a training run:
    40 optimizer steps (fused SGD) with them ON (default) and OFF give the same loss trajectory, and every gradient stays finite.
    (A buffer the side stream still reads being handed out by the caching allocator showed up here as a NaN weight gradient
    around step 17 and a derailed run - nothing a one-step test sees.)"""
    import bench
    from cim_amd import mask_iou, synthetic
    from cim_amd.core.config import cfg
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling import heads
    from cim_amd.modeling.model_builder import Generalized_RCNN
    from cim_amd.ops import gemm, maskfuse_pair

    def run(flag):
        monkeypatch.setattr(gemm, "HIGH_PRIO", flag)
        monkeypatch.setattr(maskfuse_pair, "DEFER_DW", flag)
        monkeypatch.setattr(maskfuse_pair, "DW_WGS", 256 if flag else 0)
        apply_preset("resnet50_voc")
        torch.manual_seed(cfg.RNG_SEED)
        model = Generalized_RCNN()
        bench.init_for_synthetic(model)
        model = model.to(dev).train()
        opt = bench.make_optimizer(model, torch)
        opt.overlap_update = flag          # (round 6) the big weights' update on the side stream, under the next backbone forward
        inp = synthetic.make_image_inputs("resnet50_voc", seed=3, n=300)
        iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(dev))
        t = lambda a: torch.from_numpy(a).unsqueeze(0).to(dev)
        batch = dict(data=torch.from_numpy(inp["data"]).to(dev), rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]),
                     mat=t(inp["mat"]), index=t(inp["index"]), iou_map=iou, asy_iou_map=asy, gtrois=None)
        np.random.seed(cfg.RNG_SEED)
        hist = []
        for s in range(40):
            opt.zero_grad(set_to_none=True)
            out = model(**batch)
            loss = sum(v.sum() for v in out["losses"].values())
            loss.backward()
            if s % 8 == 7:
                torch.cuda.synchronize()
                for n, p in model.named_parameters():
                    assert p.grad is None or bool(torch.isfinite(p.grad).all()), (s, n)
            opt.step()
            hist.append(float(loss))
        heads.settle_rng()
        sd = model.state_dict()            # (waits for an update still running on the side stream by itself)
        osd = opt.state_dict()
        big = max(model.parameters(), key=lambda p: p.numel())
        return hist, {k: v.detach().clone() for k, v in sd.items()}, osd["state"][max(osd["state"], key=lambda k: osd["state"][k]["momentum_buffer"].numel())]["momentum_buffer"].clone(), big.numel()

    (on, sd_on, mom_on, nbig), (off, sd_off, mom_off, _) = run(True), run(False)
    assert all(np.isfinite(on)) and on[-1] < on[0]
    assert on == off, [(i, a, b) for i, (a, b) in enumerate(zip(on, off)) if a != b][:3]
    from cim_amd.optim import sgd as _sgd
    assert nbig >= _sgd.TRAIL_MIN                                   # (the run did take the side-stream update)
    for k in sd_on:
        assert torch.equal(sd_on[k], sd_off[k]), k                  # the trained weights, bit for bit
    assert torch.equal(mom_on, mom_off)


# ------------------------------------------------------------------ optimizer (csrc/sgd.hip)
def test_fused_sgd_matches_torch_sgd(dev):
    """cim_amd.optim.SGD == torch.optim.SGD (momentum, weight decay, two groups as tools/train.py:282-311) over several
    steps, on tensors of odd sizes (tails, chunk boundaries) and an unaligned view; momentum buffers stay addressable
    the way lib/utils/net.py:_CorrectMomentum scales them."""
    from cim_amd.optim import SGD
    g = torch.Generator().manual_seed(3)
    shapes = [(1000, 50), (16384,), (16385,), (7,), (3, 5, 7), (40000,)]
    base = [torch.randn(*s, generator=g) for s in shapes]
    storage = torch.randn(1001, generator=g)
    ref_p = [b.clone().requires_grad_(True) for b in base] + [storage.clone()[1:].requires_grad_(True)]
    big = storage.clone().to(dev)
    hip_p = [b.clone().to(dev).requires_grad_(True) for b in base] + [big[1:].detach().requires_grad_(True)]   # 4-byte aligned view
    mk = lambda ps, cls: cls([dict(params=ps[:3], lr=0.05, weight_decay=0.01), dict(params=ps[3:], lr=0.1, weight_decay=0.0)],
                             lr=0.05, momentum=0.9)
    ref, hip = mk(ref_p, torch.optim.SGD), mk(hip_p, SGD)
    for step in range(4):
        for rp, hp in zip(ref_p, hip_p):
            gr = torch.randn(rp.shape, generator=g)
            rp.grad, hp.grad = gr.clone(), gr.clone().to(dev)
        if step == 2:                                   # lr change + momentum correction as net.py does it
            for opt in (ref, hip):
                for grp in opt.param_groups:
                    grp["lr"] *= 0.1
                    for p in grp["params"]:
                        opt.state[p]["momentum_buffer"] *= 0.1
        ref.step()
        v0 = [p._version for p in hip_p]
        hip.step()
        assert all(p._version > v for p, v in zip(hip_p, v0))          # raw-pointer update is visible to version counters
    for rp, hp in zip(ref_p, hip_p):
        torch.testing.assert_close(hp.detach().cpu(), rp.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(hip.state[hp]["momentum_buffer"].cpu(), ref.state[rp]["momentum_buffer"], rtol=1e-5, atol=1e-6)
    with pytest.raises(NotImplementedError):
        SGD(hip_p, lr=0.1, momentum=0.9, nesterov=True)


def test_fused_sgd_overlapped_update_is_the_same_update(dev):
    """optim.SGD.overlap_update (round 6): the big weights (>= TRAIL_MIN elements, matrix mode) are updated by a WALKING launch
    (cim_sgd_multi max_workgroups) on the package's side stream while the caller's stream goes on; the result - weights, momentum
    buffers, the row / column |max| by-products - equals the plain one-stream update bit for bit, and every documented way to read the
    weights afterwards (wait_update, state_dict, utils.net's momentum correction) is ordered behind the side stream's launch."""
    from cim_amd.optim import SGD, sgd as sgd_mod
    from cim_amd.ops import gemm
    from cim_amd.utils import net as net_utils
    g = torch.Generator().manual_seed(5)
    rows, cols = 4096, sgd_mod.TRAIL_MIN // 4096                   # exactly TRAIL_MIN elements
    base = [torch.randn(rows, cols, generator=g) * 0.1, torch.randn(1000, 50, generator=g), torch.randn(777, generator=g)]

    def run(overlap, wgs):
        ps = [b.clone().to(dev).requires_grad_(True) for b in base]
        opt = SGD([dict(params=ps[:2], lr=0.05, weight_decay=0.01), dict(params=ps[2:], lr=0.1, weight_decay=0.0)], lr=0.05, momentum=0.9)
        opt.overlap_update, opt.trail_workgroups = overlap, wgs
        gg = torch.Generator().manual_seed(6)
        scales = None
        for step in range(3):
            for p_ in ps:
                p_.grad = torch.randn(p_.shape, generator=gg).to(dev)
            if step == 2:                                           # lr change with momentum correction (lib/utils/net.py:65-82)
                for grp in opt.param_groups:
                    grp["lr"] *= 0.5
                net_utils._CorrectMomentum(opt, [q for grp in opt.param_groups for q in grp["params"]], 0.5)
            opt.step()
            if overlap:
                assert gemm._PENDING_UPDATES, "the big weight's update did not go to the side stream"
            opt.zero_grad()
            scales = gemm._registered_scales(ps[0], rows, cols)
        sd = opt.state_dict()                                       # (waits for the side stream by itself)
        assert not gemm._PENDING_UPDATES
        out = [p_.detach().clone() for p_ in ps] + [sd["state"][i]["momentum_buffer"].clone() for i in range(3)]
        torch.cuda.synchronize()
        return out + [scales[0].clone(), scales[1].clone()]

    ref = run(False, 0)
    for wgs in (256, 7, 0):
        got = run(True, wgs)
        for i, (a, b) in enumerate(zip(got, ref)):
            assert torch.equal(a, b), (wgs, i)


def test_fused_sgd_follows_replaced_state_and_storage(dev):
    """the fast path caches raw pointers of parameters and momentum buffers.  load_state_dict() replaces the
    buffer tensors and `p.data = ...` the parameter storage: the next step must use the NEW tensors (as torch.optim.SGD
    does), and state_dict() must hold the momentum that was actually applied."""
    from cim_amd.optim import SGD
    g = torch.Generator().manual_seed(9)
    base = [torch.randn(300, 40, generator=g), torch.randn(1234, generator=g)]
    ref_p = [b.clone().requires_grad_(True) for b in base]
    hip_p = [b.clone().to(dev).requires_grad_(True) for b in base]
    ref, hip = torch.optim.SGD(ref_p, lr=0.1, momentum=0.9), SGD(hip_p, lr=0.1, momentum=0.9)

    def step():
        for rp, hp in zip(ref_p, hip_p):
            gr = torch.randn(rp.shape, generator=g)
            rp.grad, hp.grad = gr.clone(), gr.clone().to(dev)
        ref.step()
        hip.step()

    step()
    step()
    # checkpoint round trip with a DIFFERENT momentum (as resuming from another run would)
    sd_r, sd_h = ref.state_dict(), hip.state_dict()
    for sd in (sd_r, sd_h):
        for st in sd["state"].values():
            st["momentum_buffer"] = st["momentum_buffer"] * 0.5 + 1.0
    ref.load_state_dict(sd_r)
    hip.load_state_dict(sd_h)
    step()
    # the parameter's storage is swapped (what .to() / .half().float() round trips do)
    for rp, hp in zip(ref_p, hip_p):
        rp.data = rp.data.clone()
        hp.data = hp.data.clone()
    step()
    for rp, hp in zip(ref_p, hip_p):
        torch.testing.assert_close(hp.detach().cpu(), rp.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(hip.state[hp]["momentum_buffer"].cpu(), ref.state[rp]["momentum_buffer"], rtol=1e-5, atol=1e-6)
        got = hip.state_dict()["state"]
    assert all(torch.equal(st["momentum_buffer"], hip.state[p]["momentum_buffer"]) for st, p in zip(got.values(), hip_p))


def test_fused_sgd_matrix_mode_hands_scales_to_the_contractions(dev):
    """Weights of >= 2^20 elements take the SGD kernel's matrix mode: same update as torch.optim.SGD, and the row / column
    max |w_new| it registers are exactly what cim_amax_rowcol computes - valid for this version of the weight only."""
    from cim_amd.optim import SGD
    from cim_amd.ops import gemm as G
    from experiments import engines as X            # cim_amax_rowcol / the f16x2 engine's linear: the consumers these scales were made for
    g = torch.Generator().manual_seed(5)
    rows, cols = 1036, 1028                                  # ragged against the 64 x 1024 tiles (and the 4-row unroll)
    w0 = torch.randn(rows, cols, generator=g)
    small0 = torch.randn(77, generator=g)
    rp = [w0.clone().requires_grad_(True), small0.clone().requires_grad_(True)]
    hp = [w0.clone().to(dev).requires_grad_(True), small0.clone().to(dev).requires_grad_(True)]
    ref = torch.optim.SGD(rp, lr=0.1, momentum=0.9, weight_decay=0.01)
    hip = SGD(hp, lr=0.1, momentum=0.9, weight_decay=0.01)
    for _ in range(3):
        for a, b in zip(rp, hp):
            gr = torch.randn(a.shape, generator=g)
            a.grad, b.grad = gr.clone(), gr.clone().to(dev)
        ref.step()
        hip.step()
    torch.testing.assert_close(hp[0].detach().cpu(), rp[0].detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(hp[1].detach().cpu(), rp[1].detach(), rtol=1e-5, atol=1e-6)
    reg = G._registered_scales(hp[0], rows, cols)
    assert reg is not None
    ra, ca = X.amax(hp[0].detach(), rows, cols, cols, True, True)
    assert torch.equal(reg[0], ra) and torch.equal(reg[1], ca)
    x = torch.randn(50, cols, generator=g).to(dev)
    y_reg = X.linear(x, hp[0])                               # uses the registered scales
    with torch.no_grad():
        hp[0].mul_(1.0)                                      # any other in-place change invalidates them
    assert G._registered_scales(hp[0], rows, cols) is None
    y_own = X.linear(x, hp[0])
    assert torch.equal(y_reg, y_own)


def test_fused_sgd_under_reference_lr_schedule(golden_dir):
    """cim_amd.optim.make_optimizer (fused HIP SGD, tools/train.py:282-311 groups) stepped through the reference's
    warm-up / decay / momentum-correction schedule: parameters and momentum history against the golden trace of
    torch.optim.SGD under lib/utils/net.py (fp32 update rule, different FMA contraction: 1e-6 relative)."""
    from test_host_cpu import _run_lr_schedule
    from cim_amd.optim import SGD, make_optimizer
    g = np.load(os.path.join(golden_dir, "lr_schedule.npz"))
    made = []

    def make(m):
        made.append(make_optimizer(m))
        return made[-1]

    out = _run_lr_schedule(make, device="cuda")
    assert isinstance(made[0], SGD)
    assert np.array_equal(out["lrs"], g["lrs"]) and np.array_equal(out["decay_lrs"], g["decay_lrs"])
    for k in ("params", "history", "decay_history", "clip_small", "clip_large"):
        np.testing.assert_allclose(out[k], g[k], rtol=2e-6, atol=1e-7, err_msg=k)
