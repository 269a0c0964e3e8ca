"""Scoring heads, Complete-Instances-Mining layer and the four training losses on MI355X.

Mirrors /root/reference/lib/modeling/heads.py (same public names, argument meaning and error
behaviour): `cls_iou_model` (:168-219), `CIM_layer` (:222-503), `cls_iou_loss` (:78-138),
`loss_weight_bag_loss` (:43-74), `mil_loss` (:140-147), `mil_bag_loss` (:149-166),
`PCL_loss` (:10-41).

The mining (`CIM_layer`) runs on hand-written HIP kernels through the C ABI of
include/cim_hip.h (cim_amd/csrc/mining.hip) - the anti-noise sampling included: the reference draws it from the
process-global legacy NumPy RNG (heads.py:459), so the host pre-draws the doubles `np.random.choice` would consume,
the device restates choice() on them, and the generator is rewound to the consumed count afterwards (`_RngLedger`):
same pseudo labels, same stream position, no host wait inside the step.  There is no CPU fallback.
"""
import ctypes
import os
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from ..utils import engine

_EPS = 1e-6


# --------------------------------------------------------------------------- losses
def _clamp(x):
    return x.clamp(_EPS, 1 - _EPS)


def mil_loss(cls_score, labels, loss_weight=None):
    """BCE with clamping (reference heads.py:140-147)."""
    s = _clamp(cls_score)
    t = labels.clamp(0, 1)
    loss = -(t * torch.log(s)) - (1 - t) * torch.log(1 - s)
    if loss_weight is not None:
        loss = loss * loss_weight
    return loss.mean()


def _pad_bg(labels):
    """[1,C] image labels -> [1,C+1] with the background column forced to 1 (heads.py:83-84)."""
    return torch.cat((labels.new_ones(labels.shape[0], 1), labels), dim=1)


def mil_bag_loss(predict_cls, predict_det, labels):
    """Image-level MIL loss of the anti-noise branch (reference heads.py:149-166)."""
    pred = _clamp((predict_cls * predict_det).sum(dim=0, keepdim=True))
    target = _pad_bg(labels) if pred.shape[-1] - 1 == labels.shape[-1] else labels
    return -(target * torch.log(pred) + (1 - target) * torch.log(1 - pred)).mean()


def loss_weight_bag_loss(predict, pseudo_labels, labels, loss_weight):
    """Weighted image-level BCE on max-aggregated scores (reference heads.py:43-74)."""
    assert predict.ndim == 2
    labels = labels.squeeze()
    assert labels.ndim == 1
    member = (pseudo_labels != 0).to(predict.dtype)
    _assert_max_is_one(member)                                     # heads.py:51
    fg_val, fg_idx = (predict * member).max(dim=0)                 # rows without a label contribute 0
    un_val, un_idx = predict.max(dim=0)
    seen = labels == 1
    agg = _clamp(fg_val * labels + un_val * (1 - labels))
    idx = torch.where(seen, fg_idx, un_idx)
    weight = torch.where(seen, loss_weight[idx], torch.ones_like(agg))
    loss = -(labels * torch.log(agg) + (1 - labels) * torch.log(1 - agg)) * weight
    return loss.mean()


def _assert_max_is_one(member):
    if member.is_cuda:
        torch._assert_async(member.max() == 1)
    else:
        assert member.max() == 1


def cls_iou_loss(cls_score, iou_score, pseudo_labels, pseudo_iou_labels, loss_weights, labels, del_iou_branch=False):
    """Refinement-branch losses (reference heads.py:78-138): returns (cls_loss, iou_loss, bag_loss).
    Written without host synchronisation: empty selections are handled by masked sums."""
    target_iou = pseudo_iou_labels.flatten().to(cls_score.dtype)
    cls = _clamp(cls_score)
    iou = _clamp(iou_score)
    padded = _pad_bg(labels)
    if del_iou_branch:
        fused = cls
    elif iou.shape[-1] == 1:
        fused = torch.cat((cls[:, 0:1], cls[:, 1:] * iou), dim=1)
    else:
        fused = cls * iou
    bag_loss = loss_weight_bag_loss(fused, pseudo_labels, padded, loss_weights)

    member = (pseudo_labels != 0).to(cls.dtype)                     # one-hot rows or all-zero rows
    n_lab = member.sum()
    w = loss_weights.view(-1, 1)
    cls_num = -(member * torch.log(cls) * w).sum()
    cls_loss = torch.where(n_lab > 0, cls_num / n_lab.clamp(min=1), torch.zeros_like(cls_num))

    fg_member = member.clone()
    fg_member[:, 0] = 0
    is_fg = fg_member.sum(-1)                                       # 1 for fg rows
    n_fg = is_fg.sum()
    if iou.shape[-1] == member.shape[-1]:
        sel = (member * iou).sum(-1)
    elif iou.shape[-1] == 1:
        sel = iou.squeeze(-1)
    else:
        raise NotImplementedError("Please check shape of fg_iou_score")
    sl1 = F.smooth_l1_loss(sel, target_iou, reduction="none")
    iou_num = (sl1 * loss_weights * is_fg).sum()
    iou_loss = torch.where(n_fg > 0, iou_num / n_fg.clamp(min=1), torch.zeros_like(iou_num))
    return cls_loss, iou_loss, bag_loss


def PCL_loss(predict_cls, mat, labels):
    """Proposal-cluster loss (reference heads.py:10-41, https://arxiv.org/abs/1807.03342).
    `mat` [N,C+1]: non-zero entries are cluster ids; the id found in column 0 is the background
    cluster.  One host read of the (tiny) id list; everything else is batched on the device."""
    ids = torch.unique(mat)
    ids = ids[ids != 0]
    if ids.numel() == 0:
        return 12 * predict_cls.new_zeros(())
    col0 = torch.unique(mat[:, 0])
    col0 = col0[col0 != 0]
    assert col0.numel() <= 1                                        # heads.py:20
    eq = mat.unsqueeze(0) == ids.view(-1, 1, 1)                     # [Kc,N,C1]
    rows = eq.any(dim=2).to(predict_cls.dtype)                      # [Kc,N]
    cols = eq.any(dim=1).to(predict_cls.dtype)                      # [Kc,C1]
    n_k = rows.sum(dim=1)                                           # [Kc]
    is_bg = (ids == col0[0]) if col0.numel() == 1 else torch.zeros_like(ids, dtype=torch.bool)
    p = _clamp(predict_cls)
    # foreground clusters: BCE(mean row vector, column indicator)
    mean_vec = _clamp((rows @ predict_cls) / n_k.clamp(min=1).unsqueeze(1))
    fg_term = (-(cols * torch.log(mean_vec)) - (1 - cols) * torch.log(1 - mean_vec)).mean(dim=1)
    # background cluster: BCE of every member row against its own non-zero pattern
    gt = (mat != 0).to(predict_cls.dtype)
    bce_rows = (-(gt * torch.log(p)) - (1 - gt) * torch.log(1 - p)).mean(dim=1)          # [N]
    bg_term = (rows * bce_rows.unsqueeze(0)).sum(dim=1) / n_k.clamp(min=1)
    term = torch.where(is_bg, bg_term, fg_term) * n_k
    return 12 * (term.sum() / (1e-6 + n_k.sum()))


# --------------------------------------------------------------------------- fused losses (HIP)
class _LossArgs(ctypes.Structure):
    """Mirror of `cim_loss_args` in include/cim_hip.h."""
    _fields_ = [("pc", ctypes.c_void_p), ("pd", ctypes.c_void_p),
                ("rc", ctypes.c_void_p * 3), ("ri", ctypes.c_void_p * 3),
                ("pseudo_labels", ctypes.c_void_p * 3), ("pseudo_iou_f16", ctypes.c_void_p * 3),
                ("loss_weights", ctypes.c_void_p * 3), ("weight_scale", ctypes.c_float * 3),
                ("layer_valid", ctypes.c_void_p), ("labels", ctypes.c_void_p),
                ("mat", ctypes.c_void_p), ("status", ctypes.c_void_p),
                ("N", ctypes.c_int), ("C1", ctypes.c_int), ("R", ctypes.c_int),
                ("part", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("ld", ctypes.c_int)]


STATUS_BITS = {1: "a class produced more than K pseudo ground truths (internal)",
               2: "the pre-drawn uniforms were exhausted (internal)",
               4: "`mat` has rows with several non-zero entries: the fused PCL loss handles the reference's "
                  "one-cluster-per-row format (tools/pre/AGPL_label_assign.py:137-185) - set CIM_PCL_GENERAL=1 for the "
                  "general ATen formulation",
               8: "several distinct cluster ids in column 0 of `mat` (heads.py:20 asserts at most one)",
               16: "more PRM clusters in `mat` than the fused PCL loss supports - set CIM_PCL_GENERAL=1"}


class FusedLossFunction(torch.autograd.Function):
    """All four losses of the training step in one HIP launch (cim_amd/csrc/losses.hip).
    Returns six 0-dim tensors (bag_loss, pcl_loss, cls_loss, iou_loss, 3 iou_loss, total) with the per-layer lmda weights
    applied: iou as the layers produce it, 3 iou as model_builder.py:199 reports it, total = bag + pcl + cls + 3 iou - what the
    driver sums up for the backward pass (lib/utils/training_stats.py:72-83).  Nothing is read back: which layers
    count (`valid`, from the mining launches) and the PRM cluster structure of `mat` are resolved on the device."""

    @staticmethod
    def forward(ctx, meta, pc, pd, *scores):
        R = meta["R"]
        fused = meta.get("fused", False)        # pc IS the heads' score matrix [N, (2 + 2R) C1]; its column blocks are the scores
        dev = pc.device
        if fused:
            all_scores = pc.contiguous()
            N, C1 = all_scores.shape[0], all_scores.shape[1] // (2 + 2 * R)
            base, blk = all_scores.data_ptr(), 4 * C1
            ptrs = [base + h * blk for h in range(2 + 2 * R)]
            ld = (2 + 2 * R) * C1
        else:
            rc, ri = scores[:R], scores[R:2 * R]
            N, C1 = pc.shape
            pc, pd = pc.contiguous(), pd.contiguous()
            rc = [t.contiguous() for t in rc]
            ri = [t.contiguous() for t in ri]
            ptrs = [pc.data_ptr(), pd.data_ptr()] + [t.data_ptr() for t in rc] + [t.data_ptr() for t in ri]
            ld = 0
        part = torch.empty((R + 2, 4), dtype=torch.float32, device=dev)
        grad = torch.empty((3 + 4 * R, N, C1), dtype=torch.float32, device=dev)
        a = _LossArgs()
        a.pc, a.pd, a.ld = ptrs[0], ptrs[1], ld
        keep = []
        for i in range(R):
            a.rc[i], a.ri[i] = ptrs[2 + i], ptrs[2 + R + i]
            y, t16, w = (x.contiguous() for x in meta["pseudo"][i])
            assert t16.dtype == torch.float16 and y.dtype == torch.float32 and w.dtype == torch.float32
            keep += [y, t16, w]
            a.pseudo_labels[i], a.pseudo_iou_f16[i], a.loss_weights[i] = y.data_ptr(), t16.data_ptr(), w.data_ptr()
            a.weight_scale[i] = float(meta["scales"][i])
        labels = meta["labels"].reshape(-1).to(torch.float32).contiguous()
        mat = meta["mat"].to(torch.float32).contiguous()
        assert tuple(mat.shape) == (N, C1), "mat must be [N, C+1]"
        valid, status = meta["valid"], meta["status"]
        a.labels, a.mat = labels.data_ptr(), mat.data_ptr()
        a.layer_valid, a.status = _lib.ptr(valid), status.data_ptr()
        a.N, a.C1, a.R = N, C1, R
        a.part, a.grad = part.data_ptr(), grad.data_ptr()
        _lib.call("cim_losses_fwd", ctypes.byref(a), _lib.stream_ptr())
        ctx.R, ctx.fused, ctx.dims = R, fused, (N, C1)
        ctx.save_for_backward(grad)
        ctx.set_materialize_grads(False)        # (unused outputs arrive as None, not as zero tensors made by a launch each)
        out = torch.empty(6, dtype=torch.float32, device=dev)
        _lib.call("cim_loss_finish", part.data_ptr(), R + 2, out.data_ptr(), _lib.stream_ptr())
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *g):
        (G,) = ctx.saved_tensors
        R = ctx.R
        # when the whole backward pass has been queued: bring the host's NumPy generator to the position the
        # mining consumed (no stall: the GPU finished the mining long before the host gets here)
        if _rng.pending is not None and not LAZY_SETTLE:
            engine.queue_callback(settle_rng)
        _order_callers_stream_after_backward(G.device)
        if ctx.fused:       # ONE launch: the gradient of the fused score matrix, ready for the head activations' backward
            N, C1 = ctx.dims
            out = torch.empty((N, (2 + 2 * R) * C1), dtype=torch.float32, device=G.device)
            g = [None if t is None else t.contiguous() for t in g]
            _lib.call("cim_loss_grad_combine", G.data_ptr(), *[_lib.ptr(t) for t in g], out.data_ptr(), N, C1, R, _lib.stream_ptr())
            return None, out, None
        zero = G.new_zeros(())
        v = [zero if t is None else t for t in g]
        g_bag, g_pcl, g_cls, g_iou = v[0] + v[5], v[1] + v[5], v[2] + v[5], v[3] + 3 * (v[4] + v[5])
        d_pc = g_bag * G[0] + g_pcl * G[1]
        d_pd = g_bag * G[2]
        d_rc = [g_cls * G[3 + 4 * i] + g_bag * G[4 + 4 * i] for i in range(R)]
        d_ri = [g_iou * G[5 + 4 * i] + g_bag * G[6 + 4 * i] for i in range(R)]
        return (None, d_pc, d_pd, *d_rc, *d_ri)


def _order_callers_stream_after_backward(dev):
    """The training step runs on a high-priority stream of its own (model_builder.Generalized_RCNN.forward) and leaves weight
    gradients on a side stream.  Autograd orders the caller's stream behind the streams of the leaf accumulations; this makes the
    ordering explicit for EVERYTHING the step enqueued: at the end of the backward pass the stream `backward()` was called on
    waits for both (an optimizer that is not ours then needs no knowledge of them)."""
    from ..ops import gemm as _g
    if not (engine.HAS_ENGINE_CALLBACK and _g.HIGH_PRIO and dev.type == "cuda"):
        return

    def order():
        cur = torch.cuda.current_stream(dev)
        hp = _g.main_stream_high_priority(dev)
        if cur != hp:
            cur.wait_stream(hp)
        cur.wait_stream(_g._side_stream(dev))

    engine.queue_callback(order)


def _fused_base(predict_cls, predict_det, ref_cls_score, ref_iou_score):
    """The heads' score matrix [N, (2 + 2R) C1] when the eight score tensors are exactly its column blocks in order (what
    cls_iou_model.forward returns on the GPU), else None."""
    ts = [predict_cls, predict_det] + list(ref_cls_score) + list(ref_iou_score)
    base = getattr(predict_cls, "_base", None)
    if base is None or base.dim() != 2 or not base.is_contiguous() or not base.is_cuda:
        return None
    n, c1 = predict_cls.shape
    if base.shape != (n, len(ts) * c1):
        return None
    for h, t in enumerate(ts):
        if getattr(t, "_base", None) is not base or t.shape != (n, c1) or t.stride() != (base.shape[1], 1) \
                or t.storage_offset() != base.storage_offset() + h * c1:
            return None
    return base


def fused_losses(predict_cls, predict_det, ref_cls_score, ref_iou_score, labels, pseudo, scales, mat,
                 valid=None, status=None, with_total=False):
    """pseudo[i] = (pseudo_labels, pseudo_iou_labels, loss_weights) of CIM_layer i (or None: layer skipped on the
    host's say-so); scales[i] = lmda; mat = the PRM cluster matrix [N,C+1]; valid = device int32 [R] from
    `mine_step` (1 = the layer found pseudo ground truths), status = device int32 [1] error word.
    Returns (bag_loss, pcl_loss, cls_loss, iou_loss) as 0-dim tensors; with_total: also (3 iou_loss, bag + pcl + cls + 3 iou)
    from the same launch."""
    R = len(ref_cls_score)
    dev = predict_cls.device
    pseudo = list(pseudo)
    if valid is None or any(ps is None for ps in pseudo):
        host_valid = torch.tensor([0 if ps is None else 1 for ps in pseudo], dtype=torch.int32)
        valid = host_valid.to(dev) if valid is None else valid * host_valid.to(dev)
        some = next((ps for ps in pseudo if ps is not None), None)
        if some is None:
            n, c1 = predict_cls.shape
            some = (predict_cls.new_zeros((n, c1)), torch.zeros(n, dtype=torch.float16, device=dev), predict_cls.new_zeros(n))
        pseudo = [some if ps is None else ps for ps in pseudo]
    own_status = status is None
    if own_status:
        status = torch.zeros(1, dtype=torch.int32, device=dev)
    meta = dict(R=R, pseudo=pseudo, scales=list(scales), labels=labels, mat=mat, valid=valid, status=status)
    base = _fused_base(predict_cls, predict_det, ref_cls_score, ref_iou_score)
    if base is not None:            # the eight scores are column blocks of ONE matrix: no copies in, one gradient tensor out
        meta["fused"] = True
        out = FusedLossFunction.apply(meta, base, None)
    else:
        out = FusedLossFunction.apply(meta, predict_cls, predict_det, *ref_cls_score, *ref_iou_score)
    if own_status:                     # stand-alone use (tests): report format errors right away
        check_status(int(status.item()))
    return tuple(out) if with_total else (out[0], out[1], out[2], out[3])


def check_status(word):
    if word:
        raise _lib.CimHipError("; ".join(msg for bit, msg in STATUS_BITS.items() if word & bit))


# --------------------------------------------------------------------------- scoring heads
class cls_iou_model(nn.Module):
    """Eight linear heads on the MaskFuse feature (reference heads.py:168-219).  Parameter names
    (`classifier`, `detector`, `refine_cls.i`, `refine_iou.i`) are the checkpoint surface."""

    def __init__(self, dim_in, dim_out, refine_times, class_agnostic=False):
        super().__init__()
        self.classifier = nn.Linear(dim_in, dim_out)
        self.detector = nn.Linear(dim_in, dim_out)
        self.refine_cls = nn.ModuleList([nn.Linear(dim_in, dim_out) for _ in range(refine_times)])
        self.refine_iou = nn.ModuleList([nn.Linear(dim_in, dim_out) for _ in range(refine_times)])

    def detectron_weight_mapping(self):
        return {name: name for name, _ in self.named_parameters()}, []

    def forward(self, seg_feature):
        if seg_feature.dim() == 4:
            seg_feature = seg_feature.squeeze(3).squeeze(2)
        # one [N,dim_in] x [dim_in, 8*dim_out] contraction instead of eight small ones
        layers = [self.classifier, self.detector] + list(self.refine_cls) + list(self.refine_iou)
        r = len(self.refine_cls)
        c1 = self.classifier.out_features
        if seg_feature.is_cuda and seg_feature.dtype == torch.float32:
            # linear (own small-tile fp32-MFMA GEMM against the concatenated weights) + activations: one autograd node, whose
            # backward hands each head's weight / bias gradient out as a slice of one [8 C1, dim_in] product
            scores = HeadsFunction.apply(seg_feature, c1, r, *[l.weight for l in layers], *[l.bias for l in layers]).split(c1, dim=1)
        else:   # CPU tensors (host-side tests of the module): the same maths in ATen
            w = torch.cat([l.weight for l in layers], dim=0)
            b = torch.cat([l.bias for l in layers], dim=0)
            logits = F.linear(seg_feature, w, b)
            lg = logits.split(c1, dim=1)
            scores = ([F.softmax(lg[0], dim=-1), F.softmax(lg[1], dim=0)] + [F.softmax(l, dim=-1) for l in lg[2:2 + r]]
                      + [torch.sigmoid(l) for l in lg[2 + r:2 + 2 * r]])
        return scores[0], scores[1], list(scores[2:2 + r]), list(scores[2 + r:2 + 2 * r])


class HeadActFunction(torch.autograd.Function):
    """softmax(classes) / softmax(proposals) / sigmoid epilogue of the 8 heads in two HIP launches
    (cim_amd/csrc/losses.hip: head_colstat_kernel + head_act_fwd_kernel), analytic backward."""

    @staticmethod
    def forward(ctx, logits, c1, r):
        logits = logits.contiguous()
        n = logits.shape[0]
        scores = torch.empty_like(logits)
        stat = torch.empty(2 * c1, dtype=torch.float32, device=logits.device)
        _lib.call("cim_head_act_fwd", logits.data_ptr(), scores.data_ptr(), stat.data_ptr(), n, c1, r, _lib.stream_ptr())
        ctx.save_for_backward(scores)
        ctx.dims = (n, c1, r)
        return scores

    @staticmethod
    def backward(ctx, g):
        (scores,) = ctx.saved_tensors
        n, c1, r = ctx.dims
        g = g.contiguous()
        dx = torch.empty_like(scores)
        dot = torch.empty(c1, dtype=torch.float32, device=scores.device)
        _lib.call("cim_head_act_bwd", scores.data_ptr(), g.data_ptr(), dx.data_ptr(), dot.data_ptr(), n, c1, r,
                  _lib.stream_ptr())
        return dx, None, None


def _small_gemm(a, b, c, m, n, k, lda, ldb, ldc, a_mcontig, b_kcontig):
    """C[m,n] = A . B on the small-tile fp32-MFMA GEMM (cim_amd/csrc/conv1x1.hip), split-K through a workspace."""
    splits = _lib.call("cim_gemm_small_splits", m, n, k)
    ws = torch.empty(splits * m * n, dtype=torch.float32, device=c.device) if splits > 1 else None
    _lib.call("cim_gemm_small_f32", a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, lda, ldb, ldc, int(a_mcontig), int(b_kcontig),
              None, None, None, None, None, 0.0, None, 0, splits, _lib.ptr(ws), _lib.stream_ptr())


class HeadsFunction(torch.autograd.Function):
    """The eight heads of heads.py:194-219 as ONE node: scores [N, 8 C1] = act(x . Wcat^T + bcat) with Wcat the eight
    weights stacked.  Forward: concatenate, one GEMM with the bias in its epilogue, the fused activation launch.  Backward:
    activation backward, dx = dlogits . Wcat, dWcat = dlogits^T . x, dbcat = column sums - the per-head gradients are
    row slices of dWcat / dbcat (no 8-way scatter)."""

    @staticmethod
    def forward(ctx, x, c1, r, *wb):
        nh = 2 + 2 * r
        ws_, bs_ = wb[:nh], wb[nh:]
        x = x.contiguous()
        n, k = x.shape
        wcat = torch.cat([w.detach() for w in ws_], dim=0)
        bcat = torch.cat([b.detach() for b in bs_], dim=0)
        m = nh * c1
        logits = torch.empty((n, m), dtype=torch.float32, device=x.device)
        splits = _lib.call("cim_gemm_small_splits", n, m, k)
        wsp = torch.empty(splits * n * m, dtype=torch.float32, device=x.device) if splits > 1 else None
        _lib.call("cim_linear_bias_f32", x.data_ptr(), wcat.data_ptr(), bcat.data_ptr(), logits.data_ptr(), n, m, k, splits,
                  _lib.ptr(wsp), _lib.stream_ptr())
        scores = torch.empty_like(logits)
        stat = torch.empty(2 * c1, dtype=torch.float32, device=x.device)
        _lib.call("cim_head_act_fwd", logits.data_ptr(), scores.data_ptr(), stat.data_ptr(), n, c1, r, _lib.stream_ptr())
        ctx.save_for_backward(x, wcat, scores)
        ctx.dims = (n, k, c1, r)
        return scores

    @staticmethod
    def backward(ctx, g):
        x, wcat, scores = ctx.saved_tensors
        n, k, c1, r = ctx.dims
        nh = 2 + 2 * r
        m = nh * c1
        g = g.contiguous()
        dl = torch.empty_like(scores)
        dot = torch.empty(c1, dtype=torch.float32, device=scores.device)
        _lib.call("cim_head_act_bwd", scores.data_ptr(), g.data_ptr(), dl.data_ptr(), dot.data_ptr(), n, c1, r, _lib.stream_ptr())
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((n, k), dtype=torch.float32, device=x.device)
            _small_gemm(dl, wcat, dx, n, k, m, m, k, k, False, False)              # dlogits [n,m] . Wcat [m,k]
        if any(ctx.needs_input_grad[3:3 + nh]):
            dw = torch.empty((m, k), dtype=torch.float32, device=x.device)
            _small_gemm(dl, x, dw, m, k, n, m, k, k, True, False)                  # dlogits^T [m,n] . x [n,k]
        if any(ctx.needs_input_grad[3 + nh:]):
            db = dl.sum(dim=0)
        gw = [dw[h * c1:(h + 1) * c1] if dw is not None and ctx.needs_input_grad[3 + h] else None for h in range(nh)]
        gb = [db[h * c1:(h + 1) * c1] if db is not None and ctx.needs_input_grad[3 + nh + h] else None for h in range(nh)]
        return (dx, None, None, *gw, *gb)


# --------------------------------------------------------------------------- mining
class _MiningLayer(ctypes.Structure):
    """Mirror of `cim_mining_layer` in include/cim_hip.h."""
    _P, _I, _F = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
    _fields_ = [("seed_score", _P), ("seed_ld", _I), ("seed_off", _I),
                ("det", _P), ("det_ld", _I), ("det_off", _I), ("det_cs", _I),
                ("wa", _P), ("wa_ld", _I), ("wa_off", _I),
                ("wb", _P), ("wb_ld", _I), ("wb_off", _I), ("wb_cs", _I),
                ("nms_thr", _F), ("cls_thr", _F), ("iou_thr", _F), ("con_thr", _F),
                ("using_cim", _I), ("anti_noise", _I), ("flag_slot", _I), ("reserved_", _I),
                ("topk", _P), ("seeds", _P), ("n_seeds", _P), ("res", _P),
                ("gt_class", _P), ("gt_weight", _P), ("pre_idx", _P), ("pre_keep", _P),
                ("gt_idx", _P), ("gt_cls", _P), ("gt_w", _P), ("counts", _P),
                ("pseudo_labels", _P), ("pseudo_iou", _P), ("loss_weights", _P), ("max_idx", _P)]


MAX_LAYERS = 4      # CIM_MAX_LAYERS


class _MiningArgs(ctypes.Structure):
    """Mirror of `cim_mining_args` in include/cim_hip.h."""
    _P, _I = ctypes.c_void_p, ctypes.c_int32
    _fields_ = [("N", _I), ("C", _I), ("K", _I), ("R", _I),
                ("labels", _P), ("iou", _P), ("asy", _P), ("asy_t", _P), ("flags", _P), ("uniforms", _P),
                ("max_uniforms", _I), ("reserved_", _I),
                ("used", _P), ("status", _P), ("layer_valid", _P),
                ("layer", _MiningLayer * MAX_LAYERS)]


class _RngLedger:
    """Keeps the process-global legacy NumPy generator bit-compatible with the reference while the anti-noise
    sampling (heads.py:451-466: np.random.choice per class, on the host, from the generator the data loader's
    epoch sampler shares - lib/roi_data/loader.py:94) runs on the device:

      draw()    snapshots the MT19937 state and draws the MAXIMUM number of doubles a step can consume
                (np.random.random_sample: the very call RandomState.choice makes), which go to the device;
      commit()  after the step's launches: asynchronous D2H of {used, status, counts} + an event;
      settle()  waits for that event (long complete when called from the end of backward), restores the snapshot
                and re-draws exactly `used` doubles: the generator ends where the reference's would.

    settle() runs automatically at the end of the backward pass (FusedLossFunction.backward queues it as an
    autograd-engine callback) and before the next draw(); `settle_rng()` does it on demand; MINING_SYNC = True
    settles inside the forward (one stall per step, the round-1 behaviour)."""

    def __init__(self):
        self.pending = None
        self.snapshot = None

    def draw(self, n):
        self.settle()
        before = np.random.get_state()
        u = np.random.random_sample(n)
        self.snapshot = (before, np.random.get_state())
        return u

    def commit(self, meta_dev):
        host = torch.empty(meta_dev.numel(), dtype=torch.int32, pin_memory=True)
        host.copy_(meta_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending = (self.snapshot, host, ev)
        self.snapshot = None
        return host, ev

    def settle(self):
        if self.pending is None:
            return
        snapshot, host, ev = self.pending
        self.pending = None
        ev.synchronize()
        used, status = int(host[0]), int(host[1])
        if snapshot is not None:
            before, after = snapshot
            now = np.random.get_state()
            if now[2] == after[2] and np.array_equal(now[1], after[1]):
                np.random.set_state(before)
                if used > 0:
                    np.random.random_sample(used)
            else:       # the caller re-seeded / used the generator in between: its state wins, nothing to rewind
                if _strict():
                    raise _lib.CimHipError("np.random was re-seeded or used between a training forward and the end of its "
                                           "backward pass: the run has left the reference's NumPy stream (CIM_STRICT=1; call "
                                           "cim_amd.modeling.heads.settle_rng() right after the forward)")
                warnings.warn("np.random was re-seeded or used between a training forward and the end of its backward "
                              "pass: the anti-noise sampling's draws are not rewound (call "
                              "cim_amd.modeling.heads.settle_rng() right after the forward to avoid this)")
        check_status(status)


_rng = _RngLedger()


def _strict():
    from ..ops import fallback
    return fallback.strict()


MINING_SYNC = not engine.HAS_ENGINE_CALLBACK        # (True: settle inside the forward, one stall per step - the round-1 behaviour)
# Opt-in (heads.LAZY_SETTLE = True): do NOT settle at the end of the backward pass - the generator is settled
# right before the NEXT step's draw (always) or by settle_rng().  The settle waits for the step's mining launches to have run on
# the GPU: at the end of backward that caps the host's lead over the GPU at about half a step; one step later the wait is never
# felt.  Valid only when nothing in this process draws from np.random between a backward pass and the next forward without
# calling settle_rng() first (the reference's epoch sampler does draw there, lib/roi_data/loader.py:94: a training loop calls
# settle_rng() before it asks the sampler for a new epoch, or leaves this off).
LAZY_SETTLE = False
# (CIM_STRICT=1 - the test suite sets it - makes leaving the reference's NumPy stream an error, not a warning: ops/fallback.py)


def settle_rng():
    """Bring np.random to the stream position the last training forward consumed (see _RngLedger).  Call it before
    using the global NumPy generator between `model(**inputs)` and the end of `loss.backward()`."""
    _rng.settle()


def _f16_map(t, name, n):
    if t is None:
        raise NotImplementedError("Please generate or download " + name)   # model_builder.py:152,159
    if not t.is_cuda:
        raise _lib.CimHipError("CIM_layer: %s must be a CUDA/HIP tensor (no CPU fallback)" % name)
    if t.dtype != torch.float16:
        raise TypeError("CIM_layer: %s must be float16 like the reference's pickled maps, got %s" % (name, t.dtype))
    if tuple(t.shape) != (n, n):
        raise ValueError("CIM_layer: %s must be [N,N]" % name)
    return t.contiguous()


class MiningResult:
    """Device-side outputs of `mine_step`.  pseudo[i] = (pseudo_labels [N,C+1] f32, pseudo_iou_labels [N] f16,
    loss_weights [N] f32) of layer i - meaningful only where valid[i] != 0 (the reference returns (None, None, None)
    for such a layer, heads.py:429-430, and the model skips it: here the losses kernel reads valid[i])."""

    def __init__(self):
        self.pseudo = []
        self.valid = None            # device int32 [R]
        self.status = None           # device int32 [1]
        self.meta = None             # device int32 [2 + 2R]: used, status, (G, G') per layer
        self.debug = []              # per layer: dict of device tensors (tests / CIM_layer.last)
        self.host = None
        self.keep_alive = None

    def commit(self):
        """Queue the D2H of the step's bookkeeping words; the NumPy generator is settled later (see _RngLedger)."""
        self.host, _ = _rng.commit(self.meta)
        if MINING_SYNC:
            _rng.settle()
        return self


def _rows_in_place(t):
    """(tensor to keep alive, row stride in elements) of a 2-D fp32 score tensor whose rows are contiguous - a column block of a
    wider matrix is read where it lies; anything else is copied."""
    t = t.detach()
    if t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]:
        return t, t.stride(0)
    t = t.to(torch.float32).contiguous()
    return t, t.shape[-1]


class ContainmentPrep:
    """What the mining needs from the containment map alone (cim_asy_prep): flags [slots, N] (heads.py:338, one row per distinct
    con_thr) and the transposed map [N, N]; `event` (or None) orders them before their first use on another stream."""
    __slots__ = ("thr_slots", "flags", "asy_t", "event", "src", "n")

    def matches(self, asy_iou_map, n, thr_slots):
        return (self.n == n and self.src is not None and asy_iou_map is not None and self.src.data_ptr() == asy_iou_map.data_ptr()
                and self.src._version == asy_iou_map._version and self.thr_slots == thr_slots)


def _thr_slots(layers, using_CIM):
    slots = []
    for l, u in zip(layers, using_CIM):
        if u and float(l.con_thr) not in slots:
            slots.append(float(l.con_thr))
    return slots


@torch.no_grad()
def prepare_containment(layers, asy_iou_map, using_CIM=None, ahead=True):
    """Launch the input-only part of a step's mining NOW: the "not a huge proposal" flags (one N x N scan per DISTINCT con_thr; the
    reference rescans in every layer, heads.py:338) and the transposed containment map (the containment step reads columns).
    ahead=True: on the side stream - Generalized_RCNN.forward calls this before the backbone, whose small kernels leave most of
    the chip idle; mine_step(prep=...) then waits for the event.  -> ContainmentPrep or None (no CIM layer)."""
    using_CIM = [True] * len(layers) if using_CIM is None else list(using_CIM)
    slots = _thr_slots(layers, using_CIM)
    if not slots or asy_iou_map is None:
        return None
    n = asy_iou_map.shape[0]
    asy = _f16_map(asy_iou_map, "asy_iou_map", n)
    dev = asy.device
    from ..ops import gemm as _G
    cur = torch.cuda.current_stream(dev)
    side = _G._side_stream(dev) if (ahead and _G.OVERLAP and not torch.cuda.is_current_stream_capturing()) else None
    prep = ContainmentPrep()
    prep.thr_slots, prep.src, prep.n, prep.event = slots, asy, n, None
    thr = (ctypes.c_float * len(slots))(*slots)

    def launch():
        prep.flags = torch.empty((len(slots), n), dtype=torch.uint8, device=dev)
        prep.asy_t = torch.empty((n, (n + 7) & ~7), dtype=torch.float16, device=dev)      # rows padded to 16 bytes
        _lib.call("cim_asy_prep", asy.data_ptr(), n, thr, len(slots), prep.flags.data_ptr(), prep.asy_t.data_ptr(), _lib.stream_ptr())

    if side is None:
        launch()
    else:
        side.wait_stream(cur)                    # (whoever made the map enqueued it on `cur`)
        with torch.cuda.stream(side):
            launch()
            prep.event = torch.cuda.Event()
            prep.event.record(side)
        asy.record_stream(side)
    return prep


_SYNC = {}       # (device, raw stream) -> zeroed int64 scratch of cim_mining_sync_bytes(): the mining launch's meeting words


def _sync_scratch(dev, st):
    """The caller-owned scratch of cim_mining_step: one per (device, stream) - calls that can be in flight together must not share
    it; zero-filled once here, every call leaves it zeroed (include/cim_hip.h)."""
    key = (dev.index, st)
    t = _SYNC.get(key)
    if t is None:
        t = _SYNC[key] = torch.zeros((int(_lib.call("cim_mining_sync_bytes")) + 7) // 8, dtype=torch.int64, device=dev)
    return t


@torch.no_grad()
def mine_step(layers, scores, labels, iou_map, asy_iou_map, using_CIM=None, prep=None):
    """The mining + assignment of all CIM layers of one training step (reference: the three CIM_layer.forward calls
    of model_builder.py:170-187) in 2 launches, nothing read back.  layers: CIM_layer modules (thresholds);
    scores[i] = (predict_cls, predict_det) fed to layer i; labels [1,C] / [C] device tensor.  prep: prepare_containment()'s
    result for this step's asy_iou_map when the caller launched it ahead (else made here, on the current stream)."""
    R = len(layers)
    assert 1 <= R <= MAX_LAYERS
    using_CIM = [True] * R if using_CIM is None else list(using_CIM)
    cls0 = scores[0][0]
    if not cls0.is_cuda:
        raise _lib.CimHipError("CIM_layer: the HIP path needs CUDA/HIP tensors (no CPU fallback)")
    dev = cls0.device
    N = cls0.shape[0]
    labels = labels.detach().reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
    C = labels.shape[0]
    assert C == 20 or C == 80                                            # heads.py:266,324
    C1 = C + 1
    K = int(np.ceil(layers[0].p_seed * N))                               # heads.py:332
    assert all(int(np.ceil(l.p_seed * N)) == K for l in layers), "the layers of one step share p_seed"
    iou_map = _f16_map(iou_map, "iou_map", N)
    any_cim = any(using_CIM)
    if any_cim:
        asy_iou_map = _f16_map(asy_iou_map, "asy_iou_map", N)
    st = _lib.stream_ptr()

    # containment flags + transposed map: input-only work, launched ahead by the model (else here)
    thr_slots = _thr_slots(layers, using_CIM)
    if any_cim:
        if prep is None or not prep.matches(asy_iou_map, N, thr_slots):
            prep = prepare_containment(layers, asy_iou_map, using_CIM, ahead=False)
        elif prep.event is not None:
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(prep.event)
            prep.flags.record_stream(cur)
            prep.asy_t.record_stream(cur)
        flags, asy_t = prep.flags, prep.asy_t
    else:
        flags, asy_t = torch.empty((1, N), dtype=torch.uint8, device=dev), None
    pre_keep = torch.empty((R, N), dtype=torch.uint8, device=dev)

    per_i = 3 * C * K + C + 5 * N
    ints = torch.empty(R * per_i + 2 + 3 * R, dtype=torch.int32, device=dev)
    meta = ints[R * per_i:R * per_i + 2 + 2 * R]                          # used, status, (G, G') x R
    valid = ints[R * per_i + 2 + 2 * R:]
    f32 = torch.empty((R, 2, N), dtype=torch.float32, device=dev)
    n_anti = sum(1 for l in layers if l.Anti_noise_sampling)
    max_u = n_anti * min(N, C * K)
    if max_u:
        u_host = torch.empty(max_u, dtype=torch.float64, pin_memory=True)
        u_host.numpy()[:] = _rng.draw(max_u)
        uniforms = u_host.to(dev, non_blocking=True)
    else:
        _rng.settle()
        uniforms = torch.empty(1, dtype=torch.float64, device=dev)

    a = _MiningArgs()
    a.N, a.C, a.K, a.R = N, C, K, R
    a.labels, a.iou, a.asy = labels.data_ptr(), iou_map.data_ptr(), _lib.ptr(asy_iou_map if any_cim else None)
    a.asy_t = _lib.ptr(asy_t)
    a.flags, a.uniforms, a.max_uniforms = flags.data_ptr(), uniforms.data_ptr(), max_u
    a.used, a.status, a.layer_valid = meta[0:1].data_ptr(), meta[1:2].data_ptr(), valid.data_ptr()
    out = MiningResult()
    keep = [labels, iou_map, asy_iou_map, uniforms, flags, asy_t, pre_keep, ints, f32]
    for i, (layer, (pcls, pdet)) in enumerate(zip(layers, scores)):
        L = a.layer[i]
        # (the scores are column blocks of the heads' ONE score matrix: read in place through their row stride, no copies)
        cls, cls_ld = _rows_in_place(pcls)
        assert cls.shape[0] == N
        cls_off = 1 if cls.shape[-1] - 1 == C else 0
        if using_CIM[i]:
            det, det_ld = _rows_in_place(pdet)
            if det.shape[-1] - 1 == C:
                det_off, det_cs = 1, 1
            elif det.shape[-1] == C:
                det_off, det_cs = 0, 1
            elif det.shape[-1] == 1:
                det_off, det_cs = 0, 0
            else:
                raise NotImplementedError("Detector only supports class-specific and class-agnostic methods")
            L.seed_score, L.seed_ld, L.seed_off = cls.data_ptr(), cls_ld, cls_off
            L.det, L.det_ld, L.det_off, L.det_cs = det.data_ptr(), det_ld, det_off, det_cs
            L.wa, L.wa_ld, L.wa_off = cls.data_ptr(), cls_ld, cls_off
            L.wb, L.wb_ld, L.wb_off, L.wb_cs = det.data_ptr(), det_ld, det_off, det_cs
            L.flag_slot = thr_slots.index(float(layer.con_thr))
            keep += [cls, det]
        else:                                                            # MIST_label, heads.py:421-427,261-316
            preds = (cls * pdet.detach().to(torch.float32) if pdet is not None else cls).contiguous()      # (a fresh [N, C1] tensor)
            L.seed_score, L.seed_ld, L.seed_off = preds.data_ptr(), preds.shape[-1], cls_off
            L.wa, L.wa_ld, L.wa_off = preds.data_ptr(), preds.shape[-1], cls_off
            L.det = L.wb = None
            keep.append(preds)
        L.nms_thr, L.cls_thr, L.iou_thr, L.con_thr = float(layer.nms_thr), float(layer.cls_thr), float(layer.iou_thr), float(layer.con_thr)
        L.using_cim, L.anti_noise = int(bool(using_CIM[i])), int(bool(layer.Anti_noise_sampling))
        w = ints[i * per_i:(i + 1) * per_i]
        o = 0
        views = {}
        for name, n in (("topk", C * K), ("seeds", C * K), ("res", C * K), ("n_seeds", C), ("gt_class", N),
                        ("pre_idx", N), ("gt_idx", N), ("gt_cls", N), ("max_idx", N)):
            views[name] = w[o:o + n]
            o += n
            setattr(L, name, views[name].data_ptr())
        L.counts = meta[2 + 2 * i:4 + 2 * i].data_ptr()
        L.gt_weight, L.gt_w = f32[i, 0].data_ptr(), f32[i, 1].data_ptr()
        L.pre_keep = pre_keep[i].data_ptr()
        pseudo_labels = torch.empty((N, C1), dtype=torch.float32, device=dev)
        pseudo_iou = torch.empty((N,), dtype=torch.float16, device=dev)
        loss_weights = torch.empty((N,), dtype=torch.float32, device=dev)
        L.pseudo_labels, L.pseudo_iou, L.loss_weights = pseudo_labels.data_ptr(), pseudo_iou.data_ptr(), loss_weights.data_ptr()
        out.pseudo.append((pseudo_labels, pseudo_iou, loss_weights))
        views.update(gt_weight=f32[i, 0], gt_w=f32[i, 1], pre_keep=pre_keep[i], counts=meta[2 + 2 * i:4 + 2 * i],
                     asy_flag=flags[L.flag_slot] if using_CIM[i] else None, C=C, K=K)
        out.debug.append(views)
    try:
        _lib.call("cim_mining_step", ctypes.byref(a), _sync_scratch(dev, st).data_ptr(), st)
    except Exception:
        if _rng.snapshot is not None:                   # nothing was launched: give the drawn uniforms back
            np.random.set_state(_rng.snapshot[0])
            _rng.snapshot = None
        raise
    out.valid, out.status, out.meta, out.keep_alive = valid, meta[1:2], meta, keep
    return out


class CIM_layer(nn.Module):
    """Complete Instances Mining (reference heads.py:222-503): top-p seeds -> mask-IoU NMS ->
    containment mining -> cross-class arbitration -> anti-noise sampling -> IoU assignment.

    `forward` keeps the reference's one-call interface, including `(None, None, None)` when no pseudo ground truth
    is found - which needs the host to look at the result, so it waits for the device once.  The model does not call
    it: `Generalized_RCNN.forward` runs all layers of a step through `mine_step` without any read-back."""

    def __init__(self, p_seed=0.1, cls_thr=0.25, iou_thr=0.5, con_thr=0.85, Anti_noise_sampling=True):
        super().__init__()
        self.p_seed = p_seed
        self.cls_thr = cls_thr
        self.nms_thr = cls_thr          # nms_thr uses the same value as cls_thr (heads.py:227)
        self.iou_thr = iou_thr
        self.con_thr = con_thr
        self.Anti_noise_sampling = Anti_noise_sampling
        self.last = {}                  # device-side intermediates of the last call (tests / debugging)

    @torch.no_grad()
    def forward(self, predict_cls, predict_det, rois, labels, iou_map=None, asy_iou_map=None, using_CIM=True):
        res = mine_step([self], [(predict_cls, predict_det)], labels, iou_map, asy_iou_map, [using_CIM]).commit()
        _rng.settle()                                                    # waits for the device: G decides the return value
        d = res.debug[0]
        G, Gk = int(res.host[2]), int(res.host[3])
        C, K = d["C"], d["K"]
        classes = torch.nonzero(labels.detach().reshape(-1)).reshape(-1).to(d["topk"].device)
        pick = lambda name: d[name].view(C, K).index_select(0, classes)
        self.last = dict(topk=pick("topk"), seeds=pick("seeds"), res=pick("res"),
                         n_seeds=d["n_seeds"].index_select(0, classes), gt_class=d["gt_class"], gt_weight=d["gt_weight"],
                         G=G, asy_flag=d["asy_flag"] if classes.numel() else None)
        if G == 0:                                                       # heads.py:429-430
            return None, None, None
        if self.Anti_noise_sampling:
            self.last["sample_keep"] = d["pre_keep"][:G].cpu().numpy().astype(bool)
        self.last["max_overlap_idx"] = d["max_idx"]
        self.last["gt_idx"] = d["gt_idx"][:Gk]
        return res.pseudo[0]
