"""Exact-fp32 MFMA contractions of the MaskFuse head (cim_amd/csrc/gemm_f32.hip) with autograd.

`linear(x, weight, bias, relu)` == F.relu(F.linear(x, weight, bias)) and
`conv3x3(x, weight, bias, relu)` == F.relu(F.conv2d(x, weight, bias, padding=1)) for the
N x 7 x 7 ROI maps of /root/reference/lib/modeling/resnet50.py:104-110,135-136; parameters keep
the reference's layouts ([out,in] and [out,in,3,3]) so checkpoints and optimizers are unchanged.
Forward and both backward contractions run on hand-written HIP; there is no CPU fallback.
"""
import torch
from torch.autograd import Function

from .. import _lib


def _ws(m, n, splits, like):
    return torch.empty(splits * m * n, dtype=torch.float32, device=like.device) if splits > 1 else None


def gemm(a, b, m, n, k, lda, ldb, a_mcontig=False, b_kcontig=False, bias=None, relu=False, out=None):
    """C[m,n] = A.B (+bias)(ReLU).  a/b are dense device tensors interpreted by the layout flags."""
    if not a.is_cuda:
        raise _lib.CimHipError("cim_amd.ops.gemm: CUDA/HIP tensors required (no CPU fallback)")
    c = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=a.device)
    splits = _lib.call("cim_gemm_f32_splits", m, n, k)
    ws = _ws(m, n, splits, a)
    _lib.call("cim_gemm_f32", a.data_ptr(), b.data_ptr(), c.data_ptr(), _lib.ptr(bias), m, n, k, lda, ldb, n,
              int(a_mcontig), int(b_kcontig), int(relu), splits, _lib.ptr(ws), _lib.stream_ptr())
    return c


class LinearFunction(Function):
    """y = relu?(x @ w.T + b); x [M,K], w [N,K] (nn.Linear layout)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x = x.contiguous()
        w = w.contiguous()
        m, k = x.shape
        n = w.shape[0]
        y = gemm(x, w, m, n, k, k, k, b_kcontig=True, bias=b, relu=relu)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.relu = relu
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        m, k = x.shape
        n = w.shape[0]
        dy = dy.contiguous()
        if ctx.relu:
            dy = dy * (y > 0)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm(dy, w, m, k, n, n, k)                                    # dY[M,N] . W[N,K]
        if ctx.needs_input_grad[1]:
            dw = gemm(dy, x, n, k, m, n, k, a_mcontig=True)                    # dY^T[N,M] . X[M,K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=0)
        return dx, dw, db, None


class Conv3x3Function(Function):
    """y = relu?(conv2d(x, w, b, padding=1)) on channels-last ROI maps.
    x: logical [R,Cin,P,P] in torch.channels_last (physical [R,P,P,Cin]); w [Cout,Cin,3,3]."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x = x.contiguous(memory_format=torch.channels_last)
        r, cin, p, _ = x.shape
        cout = w.shape[0]
        whwio = w.permute(2, 3, 1, 0).contiguous()
        y = torch.empty((r, p, p, cout), dtype=torch.float32, device=x.device)
        _lib.call("cim_conv3x3_f32", x.data_ptr(), whwio.data_ptr(), _lib.ptr(b), y.data_ptr(), r, p, cin, cout,
                  int(relu), _lib.stream_ptr())
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.relu = relu
        ctx.has_bias = b is not None
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        r, cin, p, _ = x.shape
        cout = w.shape[0]
        dy = dy.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)      # physical [R,P,P,Cout]
        if ctx.relu:
            dy = dy * (y > 0)
        dy = dy.contiguous()
        dx = dw = db = None
        st = _lib.stream_ptr()
        if ctx.needs_input_grad[0]:
            # data gradient = the same implicit GEMM on dY with flipped, in/out-swapped weights
            w2 = w.flip(2, 3).permute(2, 3, 0, 1).contiguous()                         # [3,3,Cout,Cin]
            dxp = torch.empty((r, p, p, cin), dtype=torch.float32, device=x.device)
            _lib.call("cim_conv3x3_f32", dy.data_ptr(), w2.data_ptr(), None, dxp.data_ptr(), r, p, cout, cin, 0, st)
            dx = dxp.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            m, n, k = 9 * cin, cout, r * p * p
            splits = _lib.call("cim_gemm_f32_splits", m, n, k)
            ws = _ws(m, n, splits, x)
            dwh = torch.empty((3, 3, cin, cout), dtype=torch.float32, device=x.device)
            _lib.call("cim_conv3x3_wgrad_f32", x.data_ptr(), dy.data_ptr(), dwh.data_ptr(), r, p, cin, cout, splits,
                      _lib.ptr(ws), st)
            dw = dwh.permute(3, 2, 0, 1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(dim=(0, 1, 2))
        return dx, dw, db, None


def linear(x, weight, bias=None, relu=False):
    return LinearFunction.apply(x, weight, bias, relu)


def conv3x3(x, weight, bias=None, relu=False):
    return Conv3x3Function.apply(x, weight, bias, relu)
