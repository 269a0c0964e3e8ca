"""ROIAlign on MI355X behind the reference's operator API.

Mirrors `mmcv.ops.RoIAlign` / `roi_align` as re-exported by /root/reference/lib/ops/__init__.py:6
and called at /root/reference/lib/modeling/model_builder.py:229-231 (same constructor
arguments, same NCHW tensor semantics, autograd-enabled).  Tensors are kept in
torch.channels_last memory format internally: the HIP kernels (cim_amd/csrc/roi_align.hip)
run with lanes along C.  No CPU path: CPU tensors raise.
"""
import os

import torch
from torch import nn
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from .. import _lib
from . import gemm as _gemm


# Forward kernel: the aggregated-weight form by default (each bin reads every pixel it touches once; a few ulp from
# the sample-order sum); EXACT (CIM_ROI_FWD_EXACT=1) keeps the reference's sample order, bit-identical to the oracle.
EXACT = os.environ.get("CIM_ROI_FWD_EXACT", "0") == "1"


def _check(feat, rois):
    if not feat.is_cuda:
        raise _lib.CimHipError("cim_amd.ops.roi_align: the HIP path needs CUDA/HIP tensors (no CPU fallback)")
    if feat.dtype != torch.float32:
        raise TypeError("roi_align: float32 features expected, got %s" % feat.dtype)
    if rois.dim() != 2 or rois.size(1) != 5:
        raise ValueError("roi_align: rois must be [K,5] (batch_idx, x1, y1, x2, y2)")


def _nhwc(x):
    return x.contiguous(memory_format=torch.channels_last)


def _workspace(k, p, h, w, device):
    """Scratch for the per-ROI interpolation tables (aggregated-weight forward, pixel-owner backward)."""
    nbytes = _lib.call("cim_roi_align_bwd_workspace", max(k, 1), p, h, w)
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)


def _scratch(k, b, c, h, w, device):
    """Per-ROI-group partial maps of the region-form backward (the caller frees them after the launch: the caching allocator hands
    the block out again in stream order)."""
    nbytes = _lib.call("cim_roi_align_bwd_scratch", k, b, c, h, w)
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device) if nbytes else None


def _empty_nhwc(k, c, h, w, like):
    return torch.empty((k, h, w, c), dtype=like.dtype, device=like.device).permute(0, 3, 1, 2)


class RoIAlignFunction(Function):
    @staticmethod
    def forward(ctx, feat, rois, output_size, spatial_scale, sampling_ratio, aligned):
        _check(feat, rois)
        P, P2 = _pair(output_size)
        if P != P2:
            raise ValueError("roi_align: square output_size expected")
        feat = _nhwc(feat)
        rois = rois.to(torch.float32).contiguous()
        B, C, H, W = feat.shape
        K = rois.size(0)
        out = _empty_nhwc(K, C, P, P, feat)
        ctx.tables = None if EXACT else _workspace(K, P, H, W, feat.device)      # built by the forward, reused by the backward
        _lib.call("cim_roi_align_fwd_ws", feat.data_ptr(), rois.data_ptr(), out.data_ptr(), B, C, H, W, K, P,
                  float(spatial_scale), int(sampling_ratio), int(bool(aligned)), _lib.ptr(ctx.tables), _lib.stream_ptr())
        ctx.save_for_backward(rois)
        ctx.geom = (B, C, H, W, K, P, float(spatial_scale), int(sampling_ratio), int(bool(aligned)))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (rois,) = ctx.saved_tensors
        B, C, H, W, K, P, scale, sr, aligned = ctx.geom
        grad_out = _nhwc(grad_out)
        grad_in = _empty_nhwc(B, C, H, W, grad_out)
        ws = ctx.tables if ctx.tables is not None else _workspace(K, P, H, W, grad_out.device)
        scratch = _scratch(K, B, C, H, W, grad_out.device)      # (a name, not a temporary inside the call: it must outlive the LAUNCH)
        _lib.call("cim_roi_align_bwd_ws", grad_out.data_ptr(), rois.data_ptr(), grad_in.data_ptr(), B, C, H, W, K, P,
                  scale, sr, aligned, ws.data_ptr(), int(ctx.tables is not None), _lib.ptr(scratch), _lib.stream_ptr())
        return grad_in, None, None, None, None, None


class RoIAlignMaskCatFunction(Function):
    """Fused roi_align -> (box_x, box_x * mask) channel concat: the input of
    MaskFuse.mask_branch (/root/reference/lib/modeling/resnet50.py:121-134)."""

    @staticmethod
    def forward(ctx, feat, rois, masks, output_size, spatial_scale, sampling_ratio, aligned):
        _check(feat, rois)
        P = int(output_size)
        feat = _nhwc(feat)
        rois = rois.to(torch.float32).contiguous()
        masks = masks.to(torch.float32).contiguous()
        B, C, H, W = feat.shape
        K = rois.size(0)
        if tuple(masks.shape) != (K, P, P):
            raise ValueError("roi_align_maskcat: masks must be [K,P,P]")
        cat = _empty_nhwc(K, 2 * C, P, P, feat)
        ctx.tables = None if EXACT else _workspace(K, P, H, W, feat.device)
        _lib.call("cim_roi_align_maskcat_fwd_ws", feat.data_ptr(), rois.data_ptr(), masks.data_ptr(), cat.data_ptr(),
                  B, C, H, W, K, P, float(spatial_scale), int(sampling_ratio), int(bool(aligned)), _lib.ptr(ctx.tables),
                  _lib.stream_ptr())
        ctx.save_for_backward(rois, masks)
        ctx.geom = (B, C, H, W, K, P, float(spatial_scale), int(sampling_ratio), int(bool(aligned)))
        return cat

    @staticmethod
    def backward(ctx, grad_cat):
        rois, masks = ctx.saved_tensors
        B, C, H, W, K, P, scale, sr, aligned = ctx.geom
        grad_cat = _nhwc(grad_cat)
        grad_in = _empty_nhwc(B, C, H, W, grad_cat)
        ws = ctx.tables if ctx.tables is not None else _workspace(K, P, H, W, grad_cat.device)
        scratch = _scratch(K, B, C, H, W, grad_cat.device)
        _lib.call("cim_roi_align_maskcat_bwd_ws", grad_cat.data_ptr(), rois.data_ptr(), masks.data_ptr(),
                  grad_in.data_ptr(), B, C, H, W, K, P, scale, sr, aligned, ws.data_ptr(), int(ctx.tables is not None),
                  _lib.ptr(scratch), _lib.stream_ptr())
        _gemm.run_postponed(grad_cat.device)        # MaskFuse's late weight gradients start behind this launch (ops/gemm.py)
        return grad_in, None, None, None, None, None, None


def roi_align(input, rois, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode="avg", aligned=True):
    if pool_mode != "avg":
        raise NotImplementedError("roi_align: only pool_mode='avg' is on the CIM path")
    return RoIAlignFunction.apply(input, rois, output_size, spatial_scale, sampling_ratio, aligned)


def roi_align_maskcat(input, rois, masks, output_size, spatial_scale=1.0, sampling_ratio=0, aligned=True):
    return RoIAlignMaskCatFunction.apply(input, rois, masks, output_size, spatial_scale, sampling_ratio, aligned)


class RoIAlign(nn.Module):
    """Same constructor as mmcv.ops.RoIAlign (positional: output_size, spatial_scale,
    sampling_ratio), so `RoIAlign(resolution, spatial_scale, sampling_ratio)(feat, rois)` at
    model_builder.py:230-231 works unchanged."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode="avg", aligned=True,
                 use_torchvision=False):
        super().__init__()
        self.output_size = _pair(output_size)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = int(sampling_ratio)
        self.pool_mode = pool_mode
        self.aligned = aligned
        self.use_torchvision = use_torchvision

    def forward(self, input, rois):
        return roi_align(input, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.pool_mode,
                         self.aligned)

    def __repr__(self):
        return "%s(output_size=%s, spatial_scale=%s, sampling_ratio=%s, pool_mode=%s, aligned=%s)" % (
            self.__class__.__name__, self.output_size, self.spatial_scale, self.sampling_ratio, self.pool_mode,
            self.aligned)
