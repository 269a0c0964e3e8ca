"""ORACLE (test infrastructure, not product code).

NumPy restatement of the reference's head activations and four training losses,
/root/reference/lib/modeling/heads.py:10-166 and :194-219 (formulas: SURVEY.md App. E).
`dtype` selects float32 (same arithmetic type as the reference) or float64 (used to set
tolerances).  Parity is PINNED by tests/golden/losses_*.npz (captured from the reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np


def softmax(x, axis, dtype=np.float32):
    x = x.astype(dtype)
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def sigmoid(x, dtype=np.float32):
    x = x.astype(dtype)
    return 1.0 / (1.0 + np.exp(-x))


def head_activations(logits8, dtype=np.float32):
    """heads.py:194-219.  logits8: dict classifier/detector/refine_cls[i]/refine_iou[i] -> [N,C+1]."""
    return dict(
        predict_cls=softmax(logits8["classifier"], -1, dtype),
        predict_det=softmax(logits8["detector"], 0, dtype),
        refine_cls=[softmax(l, -1, dtype) for l in logits8["refine_cls"]],
        refine_iou=[sigmoid(l, dtype) for l in logits8["refine_iou"]],
    )


def _clamp(x, dtype):
    return np.clip(x, dtype(1e-6), dtype(1 - 1e-6))


def mil_loss(cls_score, labels, dtype=np.float32):
    """heads.py:140-147."""
    s = _clamp(cls_score.astype(dtype), dtype)
    l = np.clip(labels.astype(dtype), 0, 1)
    return (-l * np.log(s) - (1 - l) * np.log(1 - s)).mean(dtype=dtype)


def mil_bag_loss(predict_cls, predict_det, labels, dtype=np.float32):
    """heads.py:149-166.  labels: [1,C]."""
    pred = (predict_cls.astype(dtype) * predict_det.astype(dtype)).sum(axis=0, keepdims=True, dtype=dtype)
    pred = _clamp(pred, dtype)
    labels = labels.reshape(1, -1).astype(dtype)
    if pred.shape[-1] - 1 == labels.shape[-1]:
        lt = np.ones((1, labels.shape[1] + 1), dtype=dtype)
        lt[:, 1:] = labels
    else:
        lt = labels
    return (-(lt * np.log(pred) + (1 - lt) * np.log(1 - pred))).mean(dtype=dtype)


def loss_weight_bag_loss(predict, pseudo_labels, labels, loss_weight, dtype=np.float32):
    """heads.py:43-74.  labels here is the bg-padded [1,C+1] vector."""
    predict = predict.astype(dtype)
    labels = labels.reshape(-1).astype(dtype)
    ind = (pseudo_labels != 0).sum(-1) != 0
    tmp = (pseudo_labels != 0).astype(dtype)
    masked = ind[:, None].astype(dtype) * predict * tmp
    fg_i = masked.argmax(axis=0)
    fg_v = masked[fg_i, np.arange(masked.shape[1])]
    un_i = predict.argmax(axis=0)
    un_v = predict[un_i, np.arange(predict.shape[1])]
    agg = _clamp(fg_v * labels + un_v * (1 - labels), dtype)
    flag = labels == 1
    agg_idx = np.where(flag, fg_i, un_i)
    w = loss_weight.astype(dtype)[agg_idx]
    w[~flag] = 1
    loss = -(labels * np.log(agg) + (1 - labels) * np.log(1 - agg)) * w
    return loss.mean(dtype=dtype)


def smooth_l1(x, t, dtype):
    d = np.abs(x.astype(dtype) - t.astype(dtype))
    return np.where(d < 1, dtype(0.5) * d * d, d - dtype(0.5))


def cls_iou_loss(cls_score, iou_score, pseudo_labels, pseudo_iou_labels, loss_weights, labels, dtype=np.float32):
    """heads.py:78-138 (class-specific IoU branch, del_iou_branch=False).
    Returns (cls_loss, iou_loss, bag_loss)."""
    pseudo_iou_labels = pseudo_iou_labels.reshape(-1).astype(dtype)
    cls_score = _clamp(cls_score.astype(dtype), dtype)
    iou_score = _clamp(iou_score.astype(dtype), dtype)
    labels = labels.reshape(1, -1).astype(dtype)
    lt = np.ones((1, labels.shape[1] + 1), dtype=dtype)
    lt[:, 1:] = labels
    ind = (pseudo_labels != 0).sum(-1) != 0
    loss_weights = loss_weights.astype(dtype)
    bag = loss_weight_bag_loss(cls_score * iou_score, pseudo_labels, lt, loss_weights, dtype)
    cls_loss = dtype(0)
    iou_loss = dtype(0)
    if ind.sum() != 0:
        pl = (pseudo_labels[ind] != 0).astype(dtype)
        pil = pseudo_iou_labels[ind]
        cs = cls_score[ind]
        io = iou_score[ind]
        lw = loss_weights[ind]
        cls_loss = (-pl * np.log(cs) * lw[:, None]).sum(dtype=dtype) / pl.sum(dtype=dtype)
        fg = (pl[:, 1:] != 0).sum(-1) != 0
        if fg.sum() != 0:
            fpl = pl[fg]
            s = (fpl * io[fg]).sum(-1, dtype=dtype)
            iou_loss = (smooth_l1(s, pil[fg], dtype) * lw[fg]).sum(dtype=dtype) / fpl.sum(dtype=dtype)
    return cls_loss, iou_loss, bag


def pcl_loss(predict_cls, mat, dtype=np.float32):
    """heads.py:10-41."""
    predict_cls = predict_cls.astype(dtype)
    bg_ind = np.setdiff1d(mat[:, 0], [0])
    if len(bg_ind) == 0:
        bg_ind = 10000
    else:
        assert len(bg_ind) == 1
        bg_ind = bg_ind[0]
    num = dtype(1e-6)
    loss = dtype(0)
    for k in np.unique(mat):
        if k != 0 and k != bg_ind:
            tf = mat == k
            rows = tf.sum(1) != 0
            r = predict_cls[rows, :]
            col = (tf.sum(0) != 0).astype(dtype)
            v = r.mean(axis=0, dtype=dtype)
            num += r.shape[0]
            loss += r.shape[0] * mil_loss(v, col, dtype)
        elif k == bg_ind:
            tf = mat == k
            rows = tf.sum(1) != 0
            r = predict_cls[rows, :]
            gt = (mat[rows, :] != 0).astype(dtype)
            num += r.shape[0]
            loss += r.shape[0] * mil_loss(r, gt, dtype)
    return dtype(12) * (loss / num)
