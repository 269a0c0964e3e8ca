"""ORACLE (test infrastructure, not product code).

NumPy restatement of the reference's Complete-Instances-Mining pseudo-label
assignment, `CIM_layer` in /root/reference/lib/modeling/heads.py:222-502.
Every function cites the lines it follows.  Parity is PINNED: tests/test_oracle_mining.py
checks this file stage by stage against tests/golden/mining_*.npz, which
tests/golden/make_golden.py captured by importing and running the reference itself.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Bit-exactness notes:
  * iou_map / asy_iou_map are float16; every map-vs-threshold compare is done in float16
    with the Python-float threshold rounded to float16 (heads.py:251,338,387,489,500-501).
  * argsort(descending) is restated as a STABLE descending sort (ties -> lower index).
  * argmax / max ties -> first index.
  * the "big proposal" relabel (heads.py:493-498) raises inside a bare try/except in the
    reference and is therefore a no-op; it is deliberately not restated.
"""
import numpy as np

F16 = np.float16


def instance_nms(temp_iou, nms_thr):
    """heads.py:237-258.  temp_iou: [K,K] float16 in score-descending order.
    Returns the selected positions (into the K list) in selection order."""
    thr = F16(nms_thr)
    alive = list(range(temp_iou.shape[0]))
    selected = []
    while alive:
        src = alive.pop(0)
        selected.append(src)
        alive = [d for d in alive if temp_iou[src, d] < thr]
    return np.asarray(selected, dtype=np.int64)


def stable_argsort_desc(x):
    """heads.py:279,354 `argsort(descending=True)`; stable: ties keep ascending index."""
    return np.argsort(-x, kind="stable")


def asy_flag(asy_iou_map, con_thr):
    """heads.py:338.  [N,1] bool: proposal i does NOT contain >= 90% of all proposals."""
    n = asy_iou_map.shape[-1]
    cnt = (asy_iou_map > F16(con_thr)).sum(axis=-1, keepdims=True)
    return cnt < 0.9 * n


def cim_label(predict_cls, predict_det, label, iou_map, asy_iou_map, p_seed, nms_thr, con_thr, trace=None):
    """heads.py:319-407 (CIM_label).  Returns gt_labels[G,C+1], gt_weights[G], gt_idxs[N] bool,
    asy_iou_flag[N,1] bool.  `trace` (dict) receives per-class intermediates."""
    label = label.reshape(-1)
    C = label.shape[-1]
    cls = (predict_cls[:, 1:] if predict_cls.shape[-1] - 1 == C else predict_cls).astype(np.float32)
    det = (predict_det[:, 1:] if predict_det.shape[-1] - 1 == C else predict_det).astype(np.float32)
    preds = cls * det
    N = cls.shape[0]
    K = int(np.ceil(p_seed * N))                                         # :332
    klasses = np.nonzero(label)[0]
    gt_labels = np.zeros((N, C + 1), dtype=np.float32)                   # :335
    gt_weights = -np.ones((N,), dtype=np.float32)                        # :336
    flag_all = asy_flag(asy_iou_map, con_thr)                            # :338
    thr = F16(con_thr)
    for c in klasses:
        cls_c = cls[:, c]
        det_c = det[:, c] if det.shape[-1] == C else det[:, 0]           # :343-349
        preds_c = preds[:, c]
        keep_sort_idx = stable_argsort_desc(cls_c)[:K]                   # :354
        temp = iou_map[keep_sort_idx][:, keep_sort_idx]                  # :361
        keep_nms = instance_nms(temp, nms_thr)                           # :363-372
        keep_nms_idx = keep_sort_idx[keep_nms]                           # :380
        flag = (asy_iou_map[:, keep_nms_idx] > thr) & flag_all           # :386-390
        res_idx = np.zeros((0,), dtype=np.int64)
        if flag.sum() != 0:                                              # :391
            flag = flag[:, flag.sum(axis=0) > 0]                         # :392
            res_det = flag.astype(np.float32) * det_c[:, None]           # :393
            res_idx = np.unique(np.argmax(res_det, axis=0))              # :394-395
            higher = preds_c[res_idx] > gt_weights[res_idx]              # :397
            if higher.sum() > 0:
                keep_idxs = res_idx[higher]
                gt_labels[keep_idxs, :] = 0                              # :400
                gt_labels[keep_idxs, c + 1] = 1
                gt_weights[keep_idxs] = preds_c[keep_idxs]
        if trace is not None:
            trace.setdefault("keep_sort_idx", []).append(keep_sort_idx.copy())
            trace.setdefault("keep_nms_idx", []).append(keep_nms_idx.copy())
            trace.setdefault("res_idx", []).append(res_idx.copy())
    gt_idxs = gt_labels.sum(axis=-1) > 0                                 # :404
    if trace is not None:
        trace["asy_iou_flag"] = flag_all.copy()
        trace["gt_idxs"] = gt_idxs.copy()
        trace["gt_labels_full"] = gt_labels.copy()
        trace["gt_weights_full"] = gt_weights.copy()
    return gt_labels[gt_idxs], gt_weights[gt_idxs], gt_idxs, flag_all


def mist_label(preds, label, iou_map, p_seed, nms_thr, trace=None):
    """heads.py:261-316 (MIST_label) with the mask-IoU NMS branch (iou_map given)."""
    label = label.reshape(-1)
    C = label.shape[-1]
    preds = (preds if preds.shape[-1] == C else preds[:, 1:]).astype(np.float32)   # :269
    N = preds.shape[0]
    K = int(np.ceil(p_seed * N))
    klasses = np.nonzero(label)[0]
    gt_labels = np.zeros((N, C + 1), dtype=np.float32)
    gt_weights = -np.ones((N,), dtype=np.float32)
    for c in klasses:
        p = preds[:, c]
        keep_sort_idx = stable_argsort_desc(p)[:K]                       # :279
        temp = iou_map[keep_sort_idx][:, keep_sort_idx]
        keep_nms_idx = keep_sort_idx[instance_nms(temp, nms_thr)]        # :296,304
        higher = p[keep_nms_idx] > gt_weights[keep_nms_idx]              # :306
        keep_idxs = keep_nms_idx[higher]
        gt_labels[keep_idxs, :] = 0
        gt_labels[keep_idxs, c + 1] = 1
        gt_weights[keep_idxs] = p[keep_idxs]
        if trace is not None:
            trace.setdefault("keep_sort_idx", []).append(keep_sort_idx.copy())
            trace.setdefault("keep_nms_idx", []).append(keep_nms_idx.copy())
    gt_idxs = gt_labels.sum(axis=-1) > 0
    if trace is not None:
        trace["gt_idxs"] = gt_idxs.copy()
    return gt_labels[gt_idxs], gt_weights[gt_idxs], gt_idxs


def anti_noise_sample(gt_labels, gt_weights, label, rng=None):
    """heads.py:447-469.  Returns the boolean keep-mask over the G pseudo-GT rows.
    rng=None draws from the global legacy NumPy RNG exactly like the reference."""
    rng = np.random if rng is None else rng
    label = label.reshape(-1)
    klasses = np.nonzero(label)[0]
    inds = np.ones((gt_labels.shape[0],), dtype=np.float32)              # :449
    for c in klasses:
        class_idx = np.nonzero(gt_labels[:, c + 1] == 1)[0]              # :453
        if len(class_idx) == 0:
            continue
        prob = gt_weights[class_idx].astype(np.float32)                  # :457
        sampled = rng.choice(class_idx, size=len(class_idx), replace=True, p=prob / prob.sum())  # :459
        sampled = np.unique(sampled)
        inds[class_idx] = 0
        inds[sampled] = 1
    return inds == 1


def assign(overlaps, gt_labels, gt_weights, cls_thr, iou_thr):
    """heads.py:476-503.  overlaps: [N,G'] float16 (columns = surviving pseudo-GTs in
    ascending proposal order).  Returns pseudo_labels[N,C+1] f32, pseudo_iou_labels[N] f16,
    loss_weights[N] f32, max_overlap_idx[N] int64."""
    max_i = np.argmax(overlaps, axis=-1)                                  # :477 ties -> first
    max_v = overlaps[np.arange(overlaps.shape[0]), max_i]
    pseudo_labels = gt_labels[max_i].copy()
    loss_weights = gt_weights[max_i].copy()
    pseudo_iou = max_v.copy()
    ignore = max_v == 0                                                   # :484
    pseudo_labels[ignore, :] = 0
    loss_weights[ignore] = 0
    bg = (max_v < F16(cls_thr)) & ~ignore                                 # :489
    pseudo_labels[bg, :] = 0
    pseudo_labels[bg, 0] = 1
    # :493-498 raises IndexError inside try/except in the reference -> no-op.
    pseudo_iou[pseudo_iou > F16(iou_thr)] = 1                             # :500
    pseudo_iou[pseudo_iou <= F16(iou_thr)] = 0                            # :501
    return pseudo_labels, pseudo_iou, loss_weights, max_i


def cim_layer_forward(predict_cls, predict_det, labels, iou_map, asy_iou_map,
                      p_seed=0.1, cls_thr=0.25, iou_thr=0.5, con_thr=0.85,
                      anti_noise_sampling=True, using_cim=True, rng=None, trace=None):
    """heads.py:410-503 (CIM_layer.forward) on NumPy arrays.  rois are not needed when
    iou_map is given (the box-IoU fallback, heads.py:299-302,374-377,432-433, is dead)."""
    nms_thr = cls_thr                                                     # :227
    if using_cim:
        gt_labels, gt_weights, gt_idxs, _ = cim_label(predict_cls, predict_det, labels, iou_map, asy_iou_map,
                                                      p_seed, nms_thr, con_thr, trace)
    else:
        preds = predict_cls * predict_det if predict_det is not None else predict_cls
        gt_labels, gt_weights, gt_idxs = mist_label(preds, labels, iou_map, p_seed, nms_thr, trace)
    if gt_idxs.sum() == 0:                                                # :429
        return None, None, None
    overlaps = iou_map[:, gt_idxs]                                        # :435
    if anti_noise_sampling:
        keep = anti_noise_sample(gt_labels, gt_weights, labels, rng)
        if trace is not None:
            trace["sample_keep"] = keep.copy()
        gt_weights = gt_weights[keep]
        gt_labels = gt_labels[keep, :]
        overlaps = overlaps[:, keep]
    pseudo_labels, pseudo_iou, loss_weights, max_i = assign(overlaps, gt_labels, gt_weights, cls_thr, iou_thr)
    if trace is not None:
        trace["max_overlap_idx"] = max_i.copy()
    return pseudo_labels, pseudo_iou, loss_weights
