"""ORACLE (test infrastructure, not product code): the CIM per-image training step on the HOST.

CPU restatement of `Generalized_RCNN.forward` + backward
(/root/reference/lib/modeling/model_builder.py:117-213): the same ATen CPU conv / linear ops the
reference would run, oracle/roi_align_ref.c for ROIAlign (forward and backward), the unfused
MaskFuse prologue of resnet50.py:131-134, oracle/mining.py for the three CIM_layer calls and the
reference's loss formulas (oracle/losses.py documents them; here they are evaluated with torch
CPU autograd so that a backward pass exists).  Used by tests/ for end-to-end parity and by
bench.py's `cpu_baseline` leg.  Never imported by the product path.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import mining, roi_align


class _RoIAlignCPU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, rois, P, scale, sr):
        ctx.rois, ctx.shape, ctx.args = rois, tuple(feat.shape), (P, scale, sr)
        return torch.from_numpy(roi_align.roi_align_fwd(feat.detach().numpy(), rois, P, scale, sr, True))

    @staticmethod
    def backward(ctx, go):
        P, scale, sr = ctx.args
        g = roi_align.roi_align_bwd(go.contiguous().numpy(), ctx.rois, ctx.shape, P, scale, sr, True)
        return torch.from_numpy(g.astype(np.float32)), None, None, None, None


def _clamp(x):
    return x.clamp(1e-6, 1 - 1e-6)


def _bce_mean(p, t):
    p = _clamp(p)
    return (-t * torch.log(p) - (1 - t) * torch.log(1 - p)).mean()


def _losses(pc, pd, rc, ri, labels, mat, pseudo):
    """Loss formulas of heads.py:10-166, torch CPU."""
    lt = torch.cat((torch.ones(1, 1), labels), 1)
    out = dict(bag_loss=torch.zeros(()), pcl_loss=torch.zeros(()), cls_loss=torch.zeros(()), iou_loss=torch.zeros(()))
    for i, ps in enumerate(pseudo):
        if ps is None:
            continue
        Y = torch.from_numpy((ps[0] != 0).astype(np.float32))
        t = torch.from_numpy(ps[1].astype(np.float32))
        w = torch.from_numpy(ps[2]) * (3 if i == 0 else 1)
        cls, iou = _clamp(rc[i]), _clamp(ri[i])
        u = cls * iou
        fg_v, fg_i = (u * Y).max(0)
        un_v, un_i = u.max(0)
        L = lt.reshape(-1)
        agg = _clamp(fg_v * L + un_v * (1 - L))
        idx = torch.where(L == 1, fg_i, un_i)
        ww = torch.where(L == 1, w[idx], torch.ones_like(agg))
        out["bag_loss"] = out["bag_loss"] + (-(L * torch.log(agg) + (1 - L) * torch.log(1 - agg)) * ww).mean()
        rows = Y.sum(1) != 0
        if rows.any():
            out["cls_loss"] = out["cls_loss"] + (-(Y * torch.log(cls)) * w[:, None]).sum() / Y.sum()
            fg = Y[:, 1:].sum(1) != 0
            if fg.any():
                s = (Y * iou).sum(1)
                out["iou_loss"] = out["iou_loss"] + 3 * (F.smooth_l1_loss(s, t, reduction="none") * w)[fg].sum() / Y[fg].sum()
    out["bag_loss"] = out["bag_loss"] + _bce_mean((pc * pd).sum(0, keepdim=True), lt)
    ids = [k for k in np.unique(mat.numpy()) if k != 0]
    bg = [k for k in np.unique(mat[:, 0].numpy()) if k != 0]
    num, acc = 1e-6, torch.zeros(())
    for k in ids:
        tf = mat == float(k)
        rws = tf.any(1)
        r = pc[rws]
        num += r.shape[0]
        if bg and k == bg[0]:
            acc = acc + r.shape[0] * _bce_mean(r, (mat[rws] != 0).float())
        else:
            acc = acc + r.shape[0] * _bce_mean(r.mean(0), tf.any(0).float())
    out["pcl_loss"] = 12 * acc / num
    return out


def step(model, inp, iou, asy, n_sub=None, seed=0, timings=None):
    """One fwd+bwd of the training step on the CPU with the parameters of `model` (a
    cim_amd Generalized_RCNN living on the CPU).  `n_sub` evaluates the proposal-linear part
    (ROIAlign, MaskFuse, heads, losses) on the first n_sub proposals only (bounded sample).
    Returns the dict of 4 losses; parameter .grad are populated."""
    tm = {} if timings is None else timings

    def tick(name, t0):
        tm[name] = tm.get(name, 0.0) + time.perf_counter() - t0

    n = inp["rois"].shape[0] if n_sub is None else n_sub
    data = torch.from_numpy(inp["data"])
    rois = inp["rois"][:n]
    masks = torch.from_numpy(inp["masks"][:n])
    labels = torch.from_numpy(inp["labels"])
    mat = torch.from_numpy(inp["mat"][:n])
    bh = model.Box_Head
    P = 7
    t0 = time.perf_counter()
    feat_b = model.Conv_Body(data)
    tick("backbone_fwd", t0)
    feat = feat_b.detach().requires_grad_(True)        # cut here so backbone backward is timed on its own
    t0 = time.perf_counter()
    box = _RoIAlignCPU.apply(feat, rois, P, float(bh.spatial_scale), 0)
    tick("roialign_fwd", t0)
    t0 = time.perf_counter()
    cat = torch.cat((box, box * masks.unsqueeze(1)), dim=1)
    y = bh.mask_branch(cat)
    seg = bh.seg_fc(y.reshape(n, -1))
    pc, pd, rc, ri = model.cls_iou_model(seg)
    tick("head_fwd", t0)
    t0 = time.perf_counter()
    np.random.seed(seed)
    pseudo = []
    iou_s, asy_s = iou[:n, :n], asy[:n, :n]
    for i, layer in enumerate(model.CIM_layer_list):
        a, b = (pc, pd) if i == 0 else (rc[i - 1], ri[i - 1])
        r = mining.cim_layer_forward(a.detach().numpy(), b.detach().numpy(), inp["labels"], iou_s, asy_s,
                                     p_seed=layer.p_seed, cls_thr=layer.cls_thr, iou_thr=layer.iou_thr,
                                     con_thr=layer.con_thr, anti_noise_sampling=layer.Anti_noise_sampling)
        pseudo.append(None if r[0] is None else r)
    tick("mining", t0)
    t0 = time.perf_counter()
    losses = _losses(pc, pd, rc, ri, labels, mat, pseudo)
    total = sum(losses.values())
    tick("losses_fwd", t0)
    t0 = time.perf_counter()
    total.backward()
    tick("head_bwd", t0)
    t0 = time.perf_counter()
    feat_b.backward(feat.grad)
    tick("backbone_bwd", t0)
    return {k: float(v.detach()) for k, v in losses.items()}
