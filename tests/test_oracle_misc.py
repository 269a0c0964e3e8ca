"""Oracle restatements of the losses, head activations and mask-IoU maps against the golden
vectors captured from the reference; ROIAlign oracle against closed-form (analytic) answers."""
import os

import numpy as np
import pytest

from cases import MINING_CASES, case_inputs, procedural
from oracle import losses, mask_iou, roi_align

F32_RTOL = 2e-5     # fp32 evaluation order differs between torch and numpy reductions
F64_RTOL = 1e-12


@pytest.mark.parametrize("name", ["n300_c20_k2", "n1000_c80_k3"])
def test_losses_match_reference(name, golden_dir):
    g = np.load(os.path.join(golden_dir, "losses_%s.npz" % name))
    m = np.load(os.path.join(golden_dir, "mining_%s.npz" % name))
    inp = case_inputs(MINING_CASES[name])
    for dt, tag, rtol in ((np.float32, "f32", F32_RTOL), (np.float64, "f64", F64_RTOL)):
        for li in range(3):
            lmda = 3 if li == 0 else 1
            cls, _, iou = inp["layers"][li]
            got = losses.cls_iou_loss(cls, iou, m["l%d_pseudo_labels" % li], m["l%d_pseudo_iou_labels" % li],
                                      lmda * m["l%d_loss_weights" % li].astype(dt), inp["labels"], dtype=dt)
            np.testing.assert_allclose(np.array(got, dtype=np.float64), g["%s_l%d_cls_iou_bag" % (tag, li)], rtol=rtol)
        cls, det, _ = inp["layers"][0]
        np.testing.assert_allclose(losses.mil_bag_loss(cls, det, inp["labels"], dt), g[tag + "_mil_bag"], rtol=rtol)
        # the reference accumulates PCL_loss into a float32 tensor (heads.py:11) even for fp64 inputs
        np.testing.assert_allclose(losses.pcl_loss(cls, inp["mat"], dt), g[tag + "_pcl"], rtol=max(rtol, 2e-7))
        cls, _, iou = inp["layers"][1]
        bg = np.zeros_like(cls)
        bg[::3, 0] = 1
        got = losses.cls_iou_loss(cls, iou, bg, m["l1_pseudo_iou_labels"], m["l1_loss_weights"].astype(dt),
                                  inp["labels"], dtype=dt)
        np.testing.assert_allclose(np.array(got, dtype=np.float64), g[tag + "_bgonly_cls_iou_bag"], rtol=rtol)
        assert bool(g[tag + "_empty_raises_assertion"])
    # fp32 vs fp64 of the reference itself: sets the tolerance the GPU tests use
    assert abs(g["f32_pcl"] - g["f64_pcl"]) / abs(g["f64_pcl"]) < 1e-5


def test_head_activations_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "heads_small.npz"))
    dim_in, C1, n = 64, 21, 50
    names = ["classifier", "detector"] + ["refine_cls.%d" % i for i in range(3)] + ["refine_iou.%d" % i for i in range(3)]
    params = {}
    k = 0
    for nm in names:                       # named_parameters order: weight, bias per module
        params[nm] = (procedural((C1, dim_in), k + 1), procedural((C1,), k + 2))
        k += 2
    x = procedural((n, dim_in), 99) * 20
    lin = {nm: x @ w.T + b for nm, (w, b) in params.items()}
    act = losses.head_activations(dict(classifier=lin["classifier"], detector=lin["detector"],
                                       refine_cls=[lin["refine_cls.%d" % i] for i in range(3)],
                                       refine_iou=[lin["refine_iou.%d" % i] for i in range(3)]))
    np.testing.assert_allclose(act["predict_cls"], g["predict_cls"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(act["predict_det"], g["predict_det"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(np.stack(act["refine_cls"]), g["refine_cls"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(np.stack(act["refine_iou"]), g["refine_iou"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("n", [6, 64, "witness"])
def test_mask_iou_matches_reference(n, golden_dir):
    g = np.load(os.path.join(golden_dir, "mask_iou_%s.npz" % n))
    n = g["iou"].shape[0]
    h, w = int(g["h"]), int(g["w"])
    masks = np.unpackbits(g["masks_packed"], axis=1)[:, :h * w].reshape(n, h, w).astype(bool)
    iou, asy = mask_iou.mask_iou_maps(masks)
    assert iou.dtype == np.float16
    np.testing.assert_array_equal(iou, g["iou"])
    np.testing.assert_array_equal(asy, g["asy"])
    if n == 6:
        iou2, asy2 = mask_iou.mask_iou_maps_loops(masks)
        np.testing.assert_array_equal(iou2, g["iou"])
        np.testing.assert_array_equal(asy2, g["asy"])


# ---- ROIAlign oracle: analytic known answers (parity unpinned vs mmcv) ----

def test_roi_align_constant_map():
    feat = np.full((1, 3, 9, 11), 2.5, dtype=np.float32)
    rois = np.array([[0, 8, 8, 100, 90], [0, 0, 0, 176, 144], [0, 33.3, 17.2, 60.1, 70.9]], dtype=np.float32)
    out = roi_align.roi_align_fwd(feat, rois, P=7, scale=1 / 16.0, aligned=True)
    np.testing.assert_allclose(out, 2.5, rtol=1e-6)


def test_roi_align_linear_ramp_is_mean_of_sample_coords():
    H, W = 20, 24
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    feat = np.stack([xx, yy, 2 * xx + 3 * yy])[None].astype(np.float32)
    # strictly interior ROI: bilinear interpolation of a linear function is exact,
    # so each bin equals the function at the bin centre (mean of its sample grid).
    rois = np.array([[0, 40, 48, 200, 240]], dtype=np.float32)
    scale = 1 / 16.0
    out = roi_align.roi_align_fwd(feat, rois, P=7, scale=scale, aligned=True)
    x1, y1, x2, y2 = rois[0, 1:] * scale - 0.5
    bw, bh = (x2 - x1) / 7, (y2 - y1) / 7
    cx = x1 + (np.arange(7) + 0.5) * bw
    cy = y1 + (np.arange(7) + 0.5) * bh
    np.testing.assert_allclose(out[0, 0], np.broadcast_to(cx[None, :], (7, 7)), rtol=1e-5)
    np.testing.assert_allclose(out[0, 1], np.broadcast_to(cy[:, None], (7, 7)), rtol=1e-5)
    np.testing.assert_allclose(out[0, 2], 2 * cx[None, :] + 3 * cy[:, None], rtol=1e-5)


def test_roi_align_outside_and_degenerate():
    rng = np.random.RandomState(0)
    feat = rng.randn(1, 2, 8, 8).astype(np.float32)
    rois = np.array([[0, 1000, 1000, 1200, 1300],      # fully outside -> 0
                     [0, 40, 40, 40, 40],              # zero-size, aligned: grid 0x0, count clamps to 1 -> 0
                     [0, 40, 40, 41, 41]], dtype=np.float32)
    out = roi_align.roi_align_fwd(feat, rois, P=7, scale=1 / 16.0, aligned=True)
    assert np.all(out[0] == 0) and np.all(out[1] == 0)
    assert np.isfinite(out).all()
    # aligned=False clamps the ROI to 1x1 feature px (roi_align_kernel.cu:82-84): one sample per bin
    out0 = roi_align.roi_align_fwd(feat, rois[1:2], P=7, scale=1 / 16.0, aligned=False)
    assert np.abs(out0).sum() > 0


def test_roi_align_bwd_is_adjoint_of_fwd():
    rng = np.random.RandomState(1)
    feat = rng.randn(1, 3, 10, 13).astype(np.float32)
    rois = np.array([[0, 3, 5, 150, 120], [0, -20, -10, 90, 200], [0, 60, 60, 75, 66]], dtype=np.float32)
    go = rng.randn(3, 3, 7, 7).astype(np.float32)
    out = roi_align.roi_align_fwd(feat, rois, aligned=True)
    gin = roi_align.roi_align_bwd(go, rois, feat.shape, aligned=True)
    # <fwd(feat), go> == <feat, bwd(go)> (the op is linear in feat)
    lhs = float((out.astype(np.float64) * go).sum())
    rhs = float((feat.astype(np.float64) * gin).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(lhs))
    # total gradient mass: every in-range sample spreads weight 1/count
    ones = roi_align.roi_align_bwd(np.ones_like(go), rois[:1], feat.shape, aligned=True)
    np.testing.assert_allclose(ones.sum(), 3 * 49, rtol=1e-5)


# ------------------------------------------------------------------ f-3: network-input image chain (oracle closed forms)
def test_image_prep_oracle_closed_forms():
    from oracle import image_prep as ip
    rng = np.random.RandomState(0)
    im = rng.randint(0, 256, size=(37, 53, 3)).astype(np.uint8)
    # scale 1: the resize is the identity (f = 0 everywhere) -> plain BGR2RGB, /255, normalise
    ref = ((im[:, :, ::-1].astype(np.float32) / np.float32(255) - ip.MEAN) / ip.STD).transpose(2, 0, 1)
    np.testing.assert_array_equal(ip.prep_image(im, 1.0), ref)
    # horizontal flip commutes with the identity resize
    np.testing.assert_array_equal(ip.prep_image(im, 1.0, hflip=True), ref[:, :, ::-1])
    # output size = round-half-even(h * scale), (w * scale): the training scales on a 375 x 500 VOC image
    big = rng.randint(0, 256, size=(375, 500, 3)).astype(np.uint8)
    for target, hw in ((480, (360, 480)), (576, (432, 576)), (688, (516, 688)), (864, (648, 864)), (1200, (900, 1200))):
        assert ip.prep_image(big, target / 500.0).shape == (3,) + hw
    # a constant image stays constant per channel; values are (k/255 - mean) / std for an integer k
    c = np.full((20, 30, 3), 200, np.uint8)
    out = ip.prep_image(c, 1.376)
    for ch in range(3):
        vals = np.unique(out[ch])
        assert vals.size == 1
        k = np.round((vals[0] * ip.STD[ch] + ip.MEAN[ch]) * 255)
        assert k in (199, 200)                        # truncation after the float32 interpolation may lose one level
    # exact 2x down-scale of a 2-periodic pattern: every output pixel is the mean of a 2 x 2 block
    blk = np.zeros((8, 12, 3), np.uint8)
    blk[0::2, 0::2], blk[0::2, 1::2], blk[1::2, 0::2], blk[1::2, 1::2] = 10, 20, 30, 40
    out = ip.prep_image(blk, 0.5)
    want = (np.float32(25) / np.float32(255) - ip.MEAN[0]) / ip.STD[0]
    np.testing.assert_allclose(out[0], want, rtol=0, atol=1e-6)
