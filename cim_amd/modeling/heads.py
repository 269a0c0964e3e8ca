"""Scoring heads, Complete-Instances-Mining layer and the four training losses on MI355X.

Mirrors /root/reference/lib/modeling/heads.py (same public names, argument meaning and error
behaviour): `cls_iou_model` (:168-219), `CIM_layer` (:222-503), `cls_iou_loss` (:78-138),
`loss_weight_bag_loss` (:43-74), `mil_loss` (:140-147), `mil_bag_loss` (:149-166),
`PCL_loss` (:10-41).

The mining (`CIM_layer`) runs on hand-written HIP kernels through the C ABI of
include/cim_hip.h (cim_amd/csrc/mining.hip); only the anti-noise sampling stays on the host,
because the reference draws it from the process-global legacy NumPy RNG (heads.py:459) and
bit-identical pseudo labels require the identical stream.  There is no CPU fallback.
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

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
                ("layer_valid", ctypes.c_int * 3), ("labels", ctypes.c_void_p),
                ("row_cluster", ctypes.c_void_p), ("row_col", ctypes.c_void_p), ("cluster_size", ctypes.c_void_p),
                ("K", ctypes.c_int), ("bg_cluster", ctypes.c_int),
                ("N", ctypes.c_int), ("C1", ctypes.c_int), ("R", ctypes.c_int),
                ("part", ctypes.c_void_p), ("grad", ctypes.c_void_p)]


class PCLPlan:
    """Cluster structure of the PRM matrix `mat` (reference heads.py:14-36), built once per image
    on the host: per-row cluster index and column, cluster sizes, background cluster.  `None` is
    returned by `build` for matrices with more than one non-zero per row (general torch path)."""

    def __init__(self, row_cluster, row_col, sizes, bg, device):
        self.K = int(len(sizes))
        self.bg = int(bg)
        self.row_cluster = torch.from_numpy(row_cluster).to(device)
        self.row_col = torch.from_numpy(row_col).to(device)
        self.sizes = torch.from_numpy(sizes).to(device)

    @staticmethod
    def build(mat, device):
        m = mat.detach().cpu().numpy() if torch.is_tensor(mat) else np.asarray(mat)
        nz = m != 0
        if (nz.sum(1) > 1).any():
            return None
        row_col = nz.argmax(1).astype(np.int32)
        vals = m[np.arange(m.shape[0]), row_col]
        ids = np.unique(vals[vals != 0])
        col0 = np.unique(m[:, 0][m[:, 0] != 0])
        assert len(col0) <= 1                                      # heads.py:20
        row_cluster = np.full(m.shape[0], -1, dtype=np.int32)
        sizes = np.zeros(len(ids), dtype=np.int32)
        for k, v in enumerate(ids):
            sel = vals == v
            row_cluster[sel] = k
            sizes[k] = int(sel.sum())
        bg = int(np.nonzero(ids == col0[0])[0][0]) if len(col0) == 1 else -1
        return PCLPlan(row_cluster, row_col, sizes, bg, device)


class FusedLossFunction(torch.autograd.Function):
    """All four losses of the training step in one HIP launch (cim_amd/csrc/losses.hip).
    Returns a tensor [4] = (bag_loss, pcl_loss, cls_loss, iou_loss) with the per-layer lmda weights
    applied and iou NOT yet multiplied by 3 (model_builder.py:199 does that)."""

    @staticmethod
    def forward(ctx, meta, pc, pd, *scores):
        R = meta["R"]
        rc, ri = scores[:R], scores[R:2 * R]
        N, C1 = pc.shape
        dev = pc.device
        pc, pd = pc.contiguous(), pd.contiguous()
        rc = [t.contiguous() for t in rc]
        ri = [t.contiguous() for t in ri]
        part = torch.empty((R + 2, 4), dtype=torch.float32, device=dev)
        grad = torch.empty((3 + 4 * R, N, C1), dtype=torch.float32, device=dev)
        a = _LossArgs()
        a.pc, a.pd = pc.data_ptr(), pd.data_ptr()
        keep = []
        for i in range(R):
            a.rc[i], a.ri[i] = rc[i].data_ptr(), ri[i].data_ptr()
            ps = meta["pseudo"][i]
            a.layer_valid[i] = 0 if ps is None else 1
            a.weight_scale[i] = float(meta["scales"][i])
            if ps is not None:
                y, t16, w = (ps[0].contiguous(), ps[1].contiguous(), ps[2].contiguous())
                assert t16.dtype == torch.float16 and y.dtype == torch.float32 and w.dtype == torch.float32
                keep += [y, t16, w]
                a.pseudo_labels[i], a.pseudo_iou_f16[i], a.loss_weights[i] = y.data_ptr(), t16.data_ptr(), w.data_ptr()
        labels = meta["labels"].reshape(-1).to(torch.float32).contiguous()
        plan = meta["plan"]
        a.labels = labels.data_ptr()
        a.K, a.bg_cluster = plan.K, plan.bg
        if plan.K:
            a.row_cluster, a.row_col, a.cluster_size = plan.row_cluster.data_ptr(), plan.row_col.data_ptr(), plan.sizes.data_ptr()
        a.N, a.C1, a.R = N, C1, R
        a.part, a.grad = part.data_ptr(), grad.data_ptr()
        _lib.call("cim_losses_fwd", ctypes.byref(a), _lib.stream_ptr())
        ctx.R = R
        ctx.save_for_backward(grad)
        return part.sum(dim=0)

    @staticmethod
    def backward(ctx, g):
        (G,) = ctx.saved_tensors
        R = ctx.R
        g_bag, g_pcl, g_cls, g_iou = g[0], g[1], g[2], g[3]
        d_pc = g_bag * G[0] + g_pcl * G[1]
        d_pd = g_bag * G[2]
        d_rc = [g_cls * G[3 + 4 * i] + g_bag * G[4 + 4 * i] for i in range(R)]
        d_ri = [g_iou * G[5 + 4 * i] + g_bag * G[6 + 4 * i] for i in range(R)]
        return (None, d_pc, d_pd, *d_rc, *d_ri)


def fused_losses(predict_cls, predict_det, ref_cls_score, ref_iou_score, labels, pseudo, scales, plan):
    """pseudo[i] = (pseudo_labels, pseudo_iou_labels, loss_weights) of CIM_layer i, or None;
    scales[i] = lmda.  Returns (bag_loss, pcl_loss, cls_loss, iou_loss) as 0-dim tensors."""
    R = len(ref_cls_score)
    meta = dict(R=R, pseudo=list(pseudo), scales=list(scales), labels=labels, plan=plan)
    out = FusedLossFunction.apply(meta, predict_cls, predict_det, *ref_cls_score, *ref_iou_score)
    return out[0], out[1], out[2], out[3]


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
        w = torch.cat([l.weight for l in layers], dim=0)
        b = torch.cat([l.bias for l in layers], dim=0)
        logits = F.linear(seg_feature, w, b)
        r = len(self.refine_cls)
        c1 = self.classifier.out_features
        if logits.is_cuda and logits.dtype == torch.float32:
            scores = HeadActFunction.apply(logits, c1, r).split(c1, dim=1)     # fused HIP epilogue
        else:   # CPU tensors (host-side tests of the module): the same maths in ATen
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


# --------------------------------------------------------------------------- mining
class MiningContext:
    """Per-image state shared by the REFINE_TIMES CIM_layer calls of one training step:
    image classes (host + device), the containment flag (the three layers share con_thr, so
    the N x N scan of heads.py:338 is done once instead of three times) and workspaces."""

    def __init__(self, labels, n, device, labels_host=None):
        if labels_host is None:
            lab = labels.detach().reshape(-1)
            labels_host = lab.cpu().numpy() if lab.is_cuda else lab.numpy()
        lab_host = np.asarray(labels_host).reshape(-1)
        self.labels_host = lab_host
        self.classes_host = np.nonzero(lab_host)[0].astype(np.int32)
        self.num_classes = lab_host.shape[0]
        self.n = n
        self.device = device
        self.classes_dev = torch.from_numpy(self.classes_host).to(device)
        self.flags = {}
        self._pinned = {}
        self.slots = 0          # CIM_layer calls enqueued on this context (one pinned staging pair each)

    def asy_flag(self, asy_iou_map, con_thr):
        key = (asy_iou_map.data_ptr(), float(con_thr))
        f = self.flags.get(key)
        if f is None:
            f = torch.empty(self.n, dtype=torch.uint8, device=self.device)
            _lib.call("cim_asy_flag", asy_iou_map.data_ptr(), self.n, float(con_thr), f.data_ptr(), _lib.stream_ptr())
            self.flags[key] = f
        return f

    def pinned(self, which, nwords):
        """Page-locked staging buffers, one pair per CIM_layer call of the step: 'down<i>' (D2H pseudo-GT list)
        and 'up<i>' (H2D survivors).  Served by PyTorch's caching host allocator."""
        buf = self._pinned.get(which)
        if buf is None or buf.numel() < nwords:
            buf = torch.empty(nwords, dtype=torch.int32).pin_memory()
            self._pinned[which] = buf
        return buf


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


class CIM_layer(nn.Module):
    """Complete Instances Mining (reference heads.py:222-503): top-p seeds -> mask-IoU NMS ->
    containment mining -> cross-class arbitration -> anti-noise sampling -> IoU assignment."""

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
    def forward(self, predict_cls, predict_det, rois, labels, iou_map=None, asy_iou_map=None, using_CIM=True,
                _ctx=None):
        return self.finish(self.enqueue(predict_cls, predict_det, rois, labels, iou_map, asy_iou_map, using_CIM, _ctx))

    @torch.no_grad()
    def enqueue(self, predict_cls, predict_det, rois, labels, iou_map=None, asy_iou_map=None, using_CIM=True,
                _ctx=None):
        """Device half of forward(): seed selection, containment mining, arbitration and the asynchronous D2H copy
        of the pseudo-GT list.  The layers of one step read only the heads' outputs, so the model enqueues all of
        them before it waits for the first (`finish`, in layer order: the NumPy RNG stream is consumed exactly as
        by the reference's sequential calls) - one host stall per step instead of one per layer."""
        if not predict_cls.is_cuda:
            raise _lib.CimHipError("CIM_layer: the HIP path needs CUDA/HIP tensors (no CPU fallback)")
        dev = predict_cls.device
        N = predict_cls.shape[0]
        iou_map = _f16_map(iou_map, "iou_map", N)
        ctx = _ctx if _ctx is not None else MiningContext(labels, N, dev)
        C = ctx.num_classes
        assert C == 20 or C == 80                                        # heads.py:266,324
        C1 = C + 1
        n_cls = int(ctx.classes_host.shape[0])
        K = int(np.ceil(self.p_seed * N))                                # heads.py:332
        st = _lib.stream_ptr()

        cls = predict_cls.detach().to(torch.float32).contiguous()
        cls_off = 1 if cls.shape[-1] - 1 == C else 0
        if using_CIM:
            asy_iou_map = _f16_map(asy_iou_map, "asy_iou_map", N)
            det = predict_det.detach().to(torch.float32).contiguous()
            if det.shape[-1] - 1 == C:
                det_off, det_cs = 1, 1
            elif det.shape[-1] == C:
                det_off, det_cs = 0, 1
            elif det.shape[-1] == 1:
                det_off, det_cs = 0, 0
            else:
                raise NotImplementedError("Detector only supports class-specific and class-agnostic methods")
            seed_score, wa, wb = cls, cls, det
        else:
            preds = cls * predict_det.detach().to(torch.float32) if predict_det is not None else cls
            preds = preds.contiguous()
            seed_score, wa, wb = preds, preds, None
            det_off = det_cs = 0

        ws = torch.empty((3 * max(n_cls, 1) * K + max(n_cls, 1),), dtype=torch.int32, device=dev)
        topk = ws[0:n_cls * K]
        seeds = ws[n_cls * K:2 * n_cls * K]
        res = ws[2 * n_cls * K:3 * n_cls * K]
        n_seeds = ws[3 * max(n_cls, 1) * K:3 * max(n_cls, 1) * K + max(n_cls, 1)]
        gt_class = torch.empty(N, dtype=torch.int32, device=dev)
        gt_weight = torch.empty(N, dtype=torch.float32, device=dev)
        gt_pack = torch.empty(1 + 3 * N, dtype=torch.int32, device=dev)

        if n_cls > 0:
            _lib.call("cim_seed_select", seed_score.data_ptr(), seed_score.shape[-1], cls_off, iou_map.data_ptr(), N,
                      ctx.classes_dev.data_ptr(), n_cls, K, float(self.nms_thr), topk.data_ptr(), seeds.data_ptr(),
                      n_seeds.data_ptr(), st)
            if using_CIM:
                flag = ctx.asy_flag(asy_iou_map, self.con_thr)
                _lib.call("cim_contain_argmax", asy_iou_map.data_ptr(), flag.data_ptr(), det.data_ptr(),
                          det.shape[-1], det_off, det_cs, N, ctx.classes_dev.data_ptr(), n_cls, K,
                          float(self.con_thr), seeds.data_ptr(), n_seeds.data_ptr(), res.data_ptr(), st)
                cand = res
            else:
                cand = seeds
        else:
            cand = ws
        _lib.call("cim_arbitrate", _lib.ptr(cand), ctx.classes_dev.data_ptr(), n_cls, K, N,
                  wa.data_ptr(), wa.shape[-1], cls_off,
                  _lib.ptr(wb), (wb.shape[-1] if wb is not None else 0), det_off, det_cs,
                  gt_class.data_ptr(), gt_weight.data_ptr(), gt_pack.data_ptr(), st)

        # ---- the one host round trip of the layer: pseudo-GT list for the NumPy sampling
        slot = ctx.slots
        ctx.slots += 1
        host = ctx.pinned("down%d" % slot, 1 + 3 * N)
        host[:1 + 3 * N].copy_(gt_pack, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return dict(ctx=ctx, slot=slot, host=host, done=done, N=N, C1=C1, n_cls=n_cls, K=K, topk=topk, seeds=seeds,
                    n_seeds=n_seeds, res=res, gt_class=gt_class, gt_weight=gt_weight, iou_map=iou_map,
                    asy_iou_map=asy_iou_map, using_CIM=using_CIM, dev=dev, keep_alive=(gt_pack, ws, cls))

    @torch.no_grad()
    def finish(self, s):
        """Host half of forward(): wait for this layer's D2H copy, anti-noise sampling on the global NumPy RNG,
        H2D of the survivors, IoU assignment."""
        ctx, N, C1, n_cls, K, dev = s["ctx"], s["N"], s["C1"], s["n_cls"], s["K"], s["dev"]
        topk, seeds, n_seeds, res, gt_class, gt_weight = (s[k] for k in ("topk", "seeds", "n_seeds", "res", "gt_class", "gt_weight"))
        iou_map, asy_iou_map, using_CIM = s["iou_map"], s["asy_iou_map"], s["using_CIM"]
        st = _lib.stream_ptr()
        s["done"].synchronize()
        hp = s["host"].numpy()
        G = int(hp[0])
        self.last = dict(topk=topk.view(n_cls, K) if n_cls else topk, seeds=seeds.view(n_cls, K) if n_cls else seeds,
                         n_seeds=n_seeds[:n_cls], res=res.view(n_cls, K) if n_cls else res,
                         gt_class=gt_class, gt_weight=gt_weight, G=G,
                         asy_flag=ctx.asy_flag(asy_iou_map, self.con_thr) if (using_CIM and n_cls > 0) else None)
        if G == 0:                                                       # heads.py:429-430
            return None, None, None
        gt_idx = hp[1:1 + G].copy()
        gt_cls = hp[1 + N:1 + N + G].copy()
        gt_w = hp[1 + 2 * N:1 + 2 * N + G].copy().view(np.float32)

        if self.Anti_noise_sampling:                                     # heads.py:438-473
            keep = np.ones(G, dtype=bool)
            for c in ctx.classes_host:
                class_idx = np.nonzero(gt_cls == c + 1)[0]
                if len(class_idx) == 0:
                    continue
                prob = gt_w[class_idx]
                sampled = np.random.choice(class_idx, size=len(class_idx), replace=True, p=prob / prob.sum())
                keep[class_idx] = False
                keep[np.unique(sampled)] = True
            self.last["sample_keep"] = keep
            gt_idx, gt_cls, gt_w = gt_idx[keep], gt_cls[keep], gt_w[keep]
        Gk = int(gt_idx.shape[0])

        up = ctx.pinned("up%d" % s["slot"], 3 * N)      # per layer: the previous layer's H2D may still be in flight
        upn = up.numpy()
        upn[0:Gk] = gt_idx
        upn[Gk:2 * Gk] = gt_cls
        upn[2 * Gk:3 * Gk] = gt_w.view(np.int32)
        dev_gt = torch.empty(3 * Gk, dtype=torch.int32, device=dev)
        dev_gt.copy_(up[:3 * Gk], non_blocking=True)

        pseudo_labels = torch.empty((N, C1), dtype=torch.float32, device=dev)
        pseudo_iou = torch.empty((N,), dtype=torch.float16, device=dev)
        loss_weights = torch.empty((N,), dtype=torch.float32, device=dev)
        max_idx = torch.empty((N,), dtype=torch.int32, device=dev)
        _lib.call("cim_assign", iou_map.data_ptr(), N, dev_gt[0:Gk].data_ptr(), dev_gt[Gk:2 * Gk].data_ptr(),
                  dev_gt[2 * Gk:3 * Gk].data_ptr(), Gk, C1, float(self.cls_thr), float(self.iou_thr),
                  pseudo_labels.data_ptr(), pseudo_iou.data_ptr(), loss_weights.data_ptr(), max_idx.data_ptr(), st)
        self.last["max_overlap_idx"] = max_idx
        return pseudo_labels, pseudo_iou, loss_weights
