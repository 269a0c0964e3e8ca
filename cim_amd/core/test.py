"""Inference with test-time augmentation on the device (f-2).

Same entry points and argument meaning as /root/reference/lib/core/test.py:48-281 (`im_detect_all`, `im_detect_bbox`,
`im_detect_bbox_aug`, `im_detect_bbox_hflip`, `im_detect_bbox_scale`): per image the reference runs 10 sequential
forward passes (TEST.SCALE flipped + 4 BBOX_AUG.SCALES x {plain, flipped} + TEST.SCALE plain, configs/*.yaml TEST
section), each preparing the image on the host (cv2), and averages the `refine_score` of the three refinement heads
over heads and passes (test.py:131-135, 219-221).

Here the image stays on the device as the BGR uint8 array (`cim_amd.utils.blob`), every pass's network input is one
HIP launch, and the plain + flipped passes of a scale run as ONE forward with batch 2: one backbone call on
[2,3,H,W], ONE ROIAlign launch over the 2N rois (batch index column), one MaskFuse / heads pass over 2N proposals
(`im_detect_bbox_aug`: 5 forwards instead of 10).  The eval branch of `Generalized_RCNN.forward` only uses the per-row
refinement scores (model_builder.py:60-68), so batching images changes nothing per proposal.
Results are device tensors (scores [N, C+... see below]); `.cpu().numpy()` gives the reference's arrays.
"""
import numpy as np
import torch

from ..utils import blob as blob_utils
from .config import cfg

_TEST_DEFAULTS = dict(SCALE=600, MAX_SIZE=1000)                      # lib/core/config.py:123-126
_AUG_DEFAULTS = dict(ENABLED=False, SCORE_HEUR="AVG", COORD_HEUR="ID", H_FLIP=False, SCALES=(), MAX_SIZE=4000,
                     SCALE_H_FLIP=False, SCALE_SIZE_DEP=False, ASPECT_RATIOS=(), ASPECT_RATIO_H_FLIP=False)   # :167-201


def _test_cfg(key):
    t = cfg.TEST if "TEST" in cfg else {}
    return t[key] if key in t else _TEST_DEFAULTS[key]


def _aug_cfg(key):
    t = cfg.TEST if "TEST" in cfg else {}
    a = t["BBOX_AUG"] if "BBOX_AUG" in t else {}
    return a[key] if key in a else _AUG_DEFAULTS[key]


def flip_boxes(boxes, im_width):
    """lib/utils/boxes.py:252-257."""
    out = boxes.clone() if torch.is_tensor(boxes) else boxes.copy()
    out[:, 0::4] = im_width - boxes[:, 2::4] - 1
    out[:, 2::4] = im_width - boxes[:, 0::4] - 1
    return out


def _scores(return_dict):
    """mean over the refinement heads (test.py:131-135)."""
    s = return_dict["refine_score"][0].clone()
    for r in return_dict["refine_score"][1:]:
        s += r
    return s / len(return_dict["refine_score"])


def _forward(model, im, target_scale, target_max_size, boxes, masks, flips, flag):
    """One forward over len(flips) views of `im` at one scale.  Returns ([scores per view], im_scale, blob_conv)."""
    dev = next(model.parameters()).device
    flag = flag or cfg.transform_mode
    im_w = int(im.shape[1])
    boxes_t = torch.as_tensor(np.asarray(boxes), dtype=torch.float32, device=dev) if not torch.is_tensor(boxes) else boxes.to(dev)
    masks_t = torch.as_tensor(np.asarray(masks), dtype=torch.float32, device=dev) if not torch.is_tensor(masks) else masks.to(dev, torch.float32)
    src = blob_utils._as_device_u8(im, dev)
    views, rois, mks = [], [], []
    im_scale = None
    for b, hf in enumerate(flips):
        ims, scales = blob_utils.prep_im_for_blob(src, None, [target_scale], target_max_size, flag, hflip=hf, device=dev)
        im_scale = scales[0]
        views.append(ims[0])
        bx = flip_boxes(boxes_t, im_w) if hf else boxes_t                    # test.py:252
        rois.append(blob_utils.project_im_rois(bx, im_scale, batch_index=b, device=dev))
        mks.append(torch.flip(masks_t, dims=(2,)) if hf else masks_t)        # test.py:256
    data = torch.stack(views, 0)                                             # same size: same image, same scale
    out = model(data=data, rois=torch.cat(rois, 0), masks=torch.cat(mks, 0), labels=None, gtrois=None, mat=None)
    n = boxes_t.shape[0]
    s = _scores(out)
    return [s[i * n:(i + 1) * n] for i in range(len(flips))], im_scale, out["blob_conv"]


def im_detect_bbox(model, im, target_scale, target_max_size, boxes=None, masks=None, mat=None, path=None, flag=None,
                   labels=None):
    """test.py:83-146: one pass.  Returns (scores [N,C], pred_boxes, im_scale, blob_conv)."""
    if cfg.DEDUP_BOXES > 0:
        raise NotImplementedError("DEDUP_BOXES > 0 (no shipped config uses it: configs/*.yaml set 0)")
    scores, im_scale, blob_conv = _forward(model, im, target_scale, target_max_size, boxes, masks, [False], flag)
    return scores[0], boxes, im_scale, blob_conv


def im_detect_bbox_hflip(model, im, target_scale, target_max_size, box_proposals=None, masks=None, mat=None, path=None,
                         flag=None, labels=None):
    """test.py:244-264: the horizontally flipped image; boxes come back in the original frame."""
    scores, im_scale, _ = _forward(model, im, target_scale, target_max_size, box_proposals, masks, [True], flag)
    return scores[0], box_proposals, im_scale


def im_detect_bbox_scale(model, im, target_scale, target_max_size, box_proposals=None, masks=None, mat=None, hflip=False,
                         path=None, flag=None, labels=None):
    """test.py:267-281."""
    scores, _, _ = _forward(model, im, target_scale, target_max_size, box_proposals, masks, [bool(hflip)], flag)
    return scores[0], box_proposals


def im_detect_bbox_aug(model, im, box_proposals=None, masks=None, mat=None, path=None, flag=None, labels=None, batched=True):
    """test.py:149-241 with the plain and flipped pass of every scale in one forward (batched=False: one by one, in
    the reference's order - same result up to fp32 summation order of the average)."""
    assert not _aug_cfg("SCALE_SIZE_DEP"), "Size dependent scaling not implemented"
    if len(_aug_cfg("ASPECT_RATIOS")):
        raise NotImplementedError("aspect-ratio augmentation (no shipped config uses it)")
    heur, coord = _aug_cfg("SCORE_HEUR"), _aug_cfg("COORD_HEUR")
    # test.py:154-160: the two heuristics are UNION together or not at all
    assert not heur == "UNION" or coord == "UNION", "Coord heuristic must be union whenever score heuristic is union"
    assert not coord == "UNION" or heur == "UNION", "Score heuristic must be union whenever coord heuristic is union"
    if heur not in ("ID", "AVG", "UNION"):
        raise NotImplementedError("Score heur {} not supported".format(heur))
    if coord not in ("ID", "AVG", "UNION"):
        raise NotImplementedError("Coord heur {} not supported".format(coord))
    scale0, max0 = _test_cfg("SCALE"), _test_cfg("MAX_SIZE")
    # views in the REFERENCE's order (test.py:171-216): flipped @ SCALE, then per scale (plain, flipped), identity LAST;
    # `order` = position of a (scale slot, flip) view in that order
    order, views = {}, []
    if _aug_cfg("H_FLIP"):
        order[(-1, True)] = len(order)
    for si, _ in enumerate(_aug_cfg("SCALES")):
        order[(si, False)] = len(order)
        if _aug_cfg("SCALE_H_FLIP"):
            order[(si, True)] = len(order)
    order[(-1, False)] = len(order)
    plan = []                                            # (scale slot, scale, max_size, [flips]): one forward each
    if batched:
        plan.append((-1, scale0, max0, [True, False] if _aug_cfg("H_FLIP") else [False]))
        for si, scale in enumerate(_aug_cfg("SCALES")):
            plan.append((si, scale, _aug_cfg("MAX_SIZE"), [False, True] if _aug_cfg("SCALE_H_FLIP") else [False]))
    else:
        if _aug_cfg("H_FLIP"):
            plan.append((-1, scale0, max0, [True]))
        for si, scale in enumerate(_aug_cfg("SCALES")):
            plan.append((si, scale, _aug_cfg("MAX_SIZE"), [False]))
            if _aug_cfg("SCALE_H_FLIP"):
                plan.append((si, scale, _aug_cfg("MAX_SIZE"), [True]))
        plan.append((-1, scale0, max0, [False]))
    scores_i = im_scale_i = blob_conv_i = None
    for slot, scale, max_size, flips in plan:
        scores, im_scale, blob_conv = _forward(model, im, scale, max_size, box_proposals, masks, flips, flag)
        for k, f in enumerate(flips):
            views.append((order[(slot, f)], scores[k]))
            if slot == -1 and not f:                     # the identity view (test.py:211-216)
                scores_i, im_scale_i = scores[k], im_scale
                blob_conv_i = blob_conv[k:k + 1]
    scores_ts = [sc for _, sc in sorted(views, key=lambda v: v[0])]
    if heur == "ID":
        scores_c = scores_i
    elif heur == "AVG":
        scores_c = torch.stack(scores_ts, 0).mean(0)
    else:
        scores_c = torch.cat(scores_ts, 0)
    # every view scores the SAME proposals (no box regression on this path: im_detect_bbox returns the proposals), so
    # ID and AVG give the proposals themselves and UNION repeats them once per view (np.vstack(boxes_ts), test.py:236)
    boxes_c = box_proposals
    if coord == "UNION" and box_proposals is not None:
        rep = len(scores_ts)
        boxes_c = (box_proposals.repeat(rep, 1) if torch.is_tensor(box_proposals)
                   else np.vstack([box_proposals] * rep))
    return scores_c, boxes_c, im_scale_i, blob_conv_i


def im_detect_all(model, im, box_proposals=None, masks=None, mat=None, timers=None, path=None, flag=None, labels=None):
    """test.py:48-80."""
    if _aug_cfg("ENABLED"):
        scores, boxes, _, _ = im_detect_bbox_aug(model, im, box_proposals, masks, mat, path=path, flag=flag, labels=labels)
    else:
        scores, boxes, _, _ = im_detect_bbox(model, im, _test_cfg("SCALE"), _test_cfg("MAX_SIZE"), box_proposals, masks, mat,
                                             path=path, flag=flag, labels=labels)
    return {"scores": scores, "boxes": boxes}
