"""The NumPy oracle (oracle/mining.py) against the golden vectors captured from the reference
(tests/golden/mining_*.npz, made by tests/golden/make_golden.py).  Bit-exact, stage by stage."""
import os

import numpy as np
import pytest

from cases import MINING_CASES, THRESHOLDS, case_inputs
from oracle import mining


def _check_layer(g, prefix, cls, det, labels, iou, asy, cls_thr, iou_thr, seed, sampling=True, using_cim=True):
    trace = {}
    np.random.seed(seed)
    out = mining.cim_layer_forward(cls, det, labels, iou, asy, p_seed=0.1, cls_thr=cls_thr, iou_thr=iou_thr,
                                   anti_noise_sampling=sampling, using_cim=using_cim, trace=trace)
    probe = np.random.random_sample()
    assert bool(g[prefix + "is_none"]) == (out[0] is None)
    assert probe == float(g[prefix + "rng_probe"]), "NumPy RNG stream position differs"
    order = g[prefix + "class_order"]
    assert list(order) == list(np.nonzero(labels.reshape(-1))[0])
    for k, c in enumerate(order):
        for v in ("keep_sort_idx", "keep_nms_idx", "res_idx"):
            key = "%sc%d_%s" % (prefix, c, v)
            if key in g.files:
                np.testing.assert_array_equal(trace[v][k], g[key], err_msg=key)
            elif v == "res_idx" and using_cim:
                assert trace[v][k].size == 0
    np.testing.assert_array_equal(trace["gt_idxs"], g[prefix + "label_gt_idxs"])
    if using_cim:
        np.testing.assert_array_equal(trace["asy_iou_flag"], g[prefix + "asy_iou_flag"])
        np.testing.assert_array_equal(trace["gt_labels_full"][trace["gt_idxs"]], g[prefix + "label_gt_labels"])
        np.testing.assert_array_equal(trace["gt_weights_full"][trace["gt_idxs"]], g[prefix + "label_gt_weights"])
    if out[0] is None:
        return
    if sampling:
        np.testing.assert_array_equal(trace["sample_keep"], g[prefix + "sample_keep"])
    np.testing.assert_array_equal(trace["max_overlap_idx"], g[prefix + "max_overlap_idx"])
    np.testing.assert_array_equal(out[0], g[prefix + "pseudo_labels"])
    assert out[1].dtype == g[prefix + "pseudo_iou_labels"].dtype == np.float16
    np.testing.assert_array_equal(out[1], g[prefix + "pseudo_iou_labels"])
    np.testing.assert_array_equal(out[2], g[prefix + "loss_weights"])


@pytest.mark.parametrize("name", list(MINING_CASES))
def test_mining_matches_reference(name, golden_dir):
    case = MINING_CASES[name]
    g = np.load(os.path.join(golden_dir, "mining_%s.npz" % name))
    inp = case_inputs(case)
    if "in_iou" in g.files:      # stored inputs == procedurally regenerated inputs
        np.testing.assert_array_equal(inp["iou"], g["in_iou"])
        np.testing.assert_array_equal(inp["asy"], g["in_asy"])
        np.testing.assert_array_equal(inp["layers"][1][0], g["in_l1_cls"])
        np.testing.assert_array_equal(inp["layers"][2][1], g["in_l2_det"])
    np.testing.assert_array_equal(inp["labels"], g["labels"])
    for li, (cls_thr, iou_thr) in enumerate(THRESHOLDS):
        cls, det, _ = inp["layers"][li]
        _check_layer(g, "l%d_" % li, cls, det, inp["labels"], inp["iou"], inp["asy"], cls_thr, iou_thr, 100 + li)
    cls, det, _ = inp["layers"][0]
    _check_layer(g, "nosample_", cls, det, inp["labels"], inp["iou"], inp["asy"], 0.25, 0.5, 7, sampling=False)
    _check_layer(g, "mist_", cls, det, inp["labels"], inp["iou"], inp["asy"], 0.25, 0.5, 8, using_cim=False)


def test_mining_degenerate(golden_dir):
    g = np.load(os.path.join(golden_dir, "mining_degenerate.npz"))
    _check_layer(g, "huge_", g["in_cls"], g["in_det"], g["labels"], g["in_iou"], g["in_asy"], 0.25, 0.5, 9)
    assert bool(g["huge_is_none"]) and not g["huge_asy_iou_flag"].any()
    out = mining.cim_layer_forward(g["in_cls"], g["in_det"], np.zeros_like(g["labels"]), g["in_iou"], g["in_asy"])
    assert out[0] is None and bool(g["nolabel_is_none"])


def test_fp16_threshold_semantics():
    """the map-vs-threshold compares run in float16."""
    v = np.array([0.35009765625, 0.3499], dtype=np.float16)
    assert list(v < np.float16(0.35)) == [False, True]
    iou = np.array([[1.0, 0.35009765625], [0.35009765625, 1.0]], dtype=np.float16)
    # iou == thr (after rounding) is NOT < thr -> second instance suppressed
    assert list(mining.instance_nms(iou, 0.35)) == [0]
