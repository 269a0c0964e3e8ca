"""1 x 1 convolution + frozen-statistics BatchNorm (+ residual) (+ ReLU) of the ResNet bottlenecks as ONE HIP launch
(cim_amd/csrc/conv1x1.hip), with autograd.

`conv1x1_bn_act(x, conv, bn, residual=None, relu=True)` == F.relu(bn(conv(x)) + residual) for an nn.Conv2d with a
1 x 1 kernel (no bias, no padding; a stride is applied by sub-sampling the input first) and an nn.BatchNorm2d in eval()
mode (fuse_input_bn=True: see ops/chain.py) - /root/reference/lib/modeling/resnet50.py:17-44 (torchvision Bottleneck conv1 / conv3 / downsample) with every
BatchNorm frozen as :53-77 does.  The parameters stay the modules' own tensors (checkpoint surface unchanged).
In NCHW the convolution of one image is W[Cout,Cin] . X[Cin,HW]: forward, data gradient (W^T . dY) and weight gradient
(dY . X^T, split-K) are three layouts of the same small-tile fp32-MFMA GEMM; the BatchNorm backward stays the fused
`bn_act` kernel.  CPU tensors, a BatchNorm in training mode or other convolution shapes take the ATen ops."""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib
from . import chain
from . import fallback
from . import gemm as _gemm_mod


def _gemm(a, b, c, m, n, k, lda, ldb, ldc, a_mcontig, b_kcontig, x_raw=None, bn=None, eps=0.0, res=None, relu=False):
    splits = _lib.call("cim_gemm_small_splits", m, n, k)
    ws = torch.empty(splits * m * n, dtype=torch.float32, device=c.device) if splits > 1 else None
    g, be, mu, var = bn if bn is not None else (None, None, None, None)
    _lib.call("cim_gemm_small_f32", a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, lda, ldb, ldc, int(a_mcontig), int(b_kcontig),
              _lib.ptr(x_raw), _lib.ptr(g), _lib.ptr(be), _lib.ptr(mu), _lib.ptr(var), float(eps), _lib.ptr(res), int(relu), splits,
              _lib.ptr(ws), _lib.stream_ptr())


class Conv1x1BnActFunction(Function):
    @staticmethod
    def forward(ctx, x, w, res, gamma, beta, mean, var, eps, relu, in_bn=None, state=None):
        x = x.contiguous()
        B, cin, H, W = x.shape
        cout, hw = w.shape[0], H * W
        w2 = w.reshape(cout, cin)
        need_grad = any(ctx.needs_input_grad[:6])
        y = torch.empty((B, cout, H, W), dtype=torch.float32, device=x.device)
        xr = torch.empty_like(y) if need_grad else None            # convolution output: the BatchNorm backward's x
        if res is not None:
            res = res.contiguous()
        for b in range(B):
            _gemm(w2, x[b], y[b], cout, hw, cin, cin, hw, hw, False, False, x_raw=(xr[b] if xr is not None else None),
                  bn=(gamma, beta, mean, var), eps=eps, res=(res[b] if res is not None else None), relu=relu)
        if need_grad:
            ctx.save_for_backward(x, w2, xr, y if relu else None, gamma, mean, var)
        ctx.param = w if isinstance(w, torch.nn.Parameter) else None      # (its .grad tells the backward whether it may defer the join)
        ctx.cfg = (B, cin, cout, H, W, float(eps), bool(relu), res is not None)
        ctx.in_bn, ctx.state = in_bn, state          # ops/chain.py: the producer's BatchNorm (fused into dx) / this layer's hand-over state
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2, xr, y, gamma, mean, var = ctx.saved_tensors
        B, cin, cout, H, W, eps, relu, has_res = ctx.cfg
        hw = H * W
        is_dconv = chain.take(dy, ctx.state)         # the consumer's data gradient already applied this layer's BatchNorm + ReLU backward
        dy = dy.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_res = has_res and ctx.needs_input_grad[2]
        need_mean = ctx.needs_input_grad[5]          # a convolution bias folded into the mean: d/dmean = -a sum dz
        need_affine = ctx.needs_input_grad[3] or ctx.needs_input_grad[4] or need_mean
        # ONE host call: BatchNorm / ReLU backward, data gradient, weight gradient (+ split-K reduces)  (csrc/conv1x1.hip)
        dev = dy.device
        dres = torch.empty_like(dy) if need_res else None
        dgamma = dbeta = None
        if need_affine:
            alloc = torch.zeros if _lib.call("cim_bn_act_bwd_chunks", B, cout, hw) > 1 else torch.empty
            dgamma, dbeta = alloc(2, cout, dtype=torch.float32, device=dev).unbind(0)
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=dev) if need_w else None
        ws = torch.empty(_lib.call("cim_conv1x1_bwd_workspace", B, cin, cout, hw) // 4, dtype=torch.float32, device=dev)
        side, join = _gemm_mod.side_stream_for_backward(dev, ctx.param if (need_x and need_w) else None)
        _lib.call("cim_conv1x1_bn_act_bwd", dy.data_ptr(), _lib.ptr(y), xr.data_ptr(), x.data_ptr(), w2.data_ptr(),
                  gamma.data_ptr(), mean.data_ptr(), var.data_ptr(), eps, int(relu), _lib.ptr(dres), _lib.ptr(dgamma),
                  _lib.ptr(dbeta), _lib.ptr(dx), _lib.ptr(dw), B, cin, cout, hw, ws.data_ptr(), _lib.stream_ptr(), side, join,
                  int(is_dconv), *_in_bn_args(ctx.in_bn if need_x else None))
        if ctx.in_bn is not None and need_x:
            chain.hand_over(dx)
        if not join:                       # the weight gradient is still running on the side stream: installed as .grad at the join
            _gemm_mod.defer_side_join(dev, ctx.param, dw, ws, x)
            dw = None
        dmean = -(gamma * torch.rsqrt(var + eps)) * dbeta if need_mean else None
        return dx, dw, dres, (dgamma if ctx.needs_input_grad[3] else None), (dbeta if ctx.needs_input_grad[4] else None), \
            dmean, None, None, None, None, None


def _in_bn_args(in_bn):
    if in_bn is None:
        return None, None, 0.0
    return in_bn[0].data_ptr(), in_bn[1].data_ptr(), float(in_bn[2])


def conv1x1_bn_act(x, conv, bn, residual=None, relu=True, fuse_input_bn=False):
    """relu?(bn(conv(x)) + residual) for a 1 x 1 nn.Conv2d `conv` and an nn.BatchNorm2d `bn`.  A convolution bias (HRNet's
    final_layer, HRNet.py:298-312) is folded into the BatchNorm's mean: bn(conv + bias) = a conv + (beta - (mean - bias) a)."""
    stride = conv.stride[0]
    fused = (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.kernel_size == (1, 1)
             and conv.padding == (0, 0) and conv.stride[0] == conv.stride[1] and conv.groups == 1
             and (not bn.training) and bn.affine and bn.track_running_stats)
    if not fused:
        if x.is_cuda:
            fallback.note("conv1x1_bn_act", "BatchNorm in training mode" if bn.training else "unsupported geometry %s" % (conv,))
        out = bn(conv(x))
        if residual is not None:
            out = out + residual
        return F.relu(out) if relu else out
    in_bn = chain.input_bn(x, fuse_input_bn and stride == 1 and torch.is_grad_enabled() and x.requires_grad)
    if stride != 1:                                   # a strided 1 x 1 convolution only sees every stride-th pixel
        x = x[:, :, ::stride, ::stride]
    mean = bn.running_mean if conv.bias is None else bn.running_mean - conv.bias
    args = (x, conv.weight, residual, bn.weight, bn.bias, mean, bn.running_var, bn.eps, relu)
    if not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in args[:6])):
        with torch.no_grad():
            return Conv1x1BnActFunction.apply(*args)
    state = {"taken": False}
    out = Conv1x1BnActFunction.apply(*args, in_bn, state)
    if chain.tag(out, bn.weight, bn.bias, mean, bn.running_var, bn.eps, relu, residual is not None) is not None:
        out._cim_bn = out._cim_bn[:3] + (state,)      # (the node's own state object: its backward checks it)
    return out
