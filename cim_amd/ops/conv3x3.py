"""3 x 3 convolution + frozen-statistics BatchNorm (+ residual) (+ ReLU) of the ResNet bottlenecks as ONE HIP launch
(implicit GEMM, cim_amd/csrc/conv1x1.hip: conv3x3_small_kernel), with autograd.

`conv3x3_bn_act(x, conv, bn, residual=None, relu=True)` == F.relu(bn(conv(x)) + residual) for an nn.Conv2d with a
3 x 3 kernel, padding 1, stride 1 or 2, no bias, and an nn.BatchNorm2d in eval() mode -
/root/reference/lib/modeling/resnet50.py:17-44 (torchvision Bottleneck conv2 / bn2) with every BatchNorm frozen as
:53-77 does.  The parameters stay the modules' own tensors (checkpoint surface unchanged; the kernel reads the
[Cout, Cin, 3, 3] weight as it is).  Forward, data gradient and weight gradient are three loaders of the same kernel; the
BatchNorm backward stays the fused `bn_act` kernel.  CPU tensors, a BatchNorm in training mode or other convolution
shapes take the ATen ops."""
import ctypes
import weakref

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib
from . import chain
from . import fallback
from .conv1x1 import affine_outputs
from . import gemm as _gemm_mod


# ---- the data gradient's A operand: the weight transposed to [cout][3][3][cin].  It depends on the weight only, so a body makes
# all of its layers' transposes in ONE launch on the side stream at the start of the forward pass (prefetch_transposed_weights:
# 10 launches per step less on the backward's main chain at cfg2); a layer that was not prefetched transposes in its own backward.
_WT = {}      # id(weight) -> (version, data_ptr, transposed tensor, event, weak reference: an id can be reused by another tensor)


class _WtDesc(ctypes.Structure):              # cim_wt_desc of include/cim_hip.h
    _fields_ = [("w", ctypes.c_void_p), ("wt", ctypes.c_void_p), ("cin", ctypes.c_int), ("cout", ctypes.c_int)]


def _transposed(w):
    # Under HIP-graph capture the prefetched buffer must NOT be used: its address would be baked into the graph, replays never
    # re-run the prefetch, and after the first optimizer step the captured data gradient would read the capture-time weights
    # (later: freed memory).  The captured call then makes the transpose itself, in its own workspace, on every replay.
    if torch.cuda.is_current_stream_capturing():
        return None
    e = _WT.get(id(w))
    if e is not None and e[4]() is w and e[0] == w._version and e[1] == w.data_ptr():
        torch.cuda.current_stream(w.device).wait_event(e[3])
        e[2].record_stream(torch.cuda.current_stream(w.device))
        return e[2]
    return None


def prefetch_transposed_weights(convs):
    """`convs`: the 3 x 3 nn.Conv2d modules of a body whose data gradient this step will need."""
    ws = [c.weight for c in convs if c.weight.is_cuda and c.weight.requires_grad and c.weight.is_contiguous()
          and c.kernel_size == (3, 3) and c.in_channels * 36 <= 64 * 1024]
    ws = [w for w in ws if not (id(w) in _WT and _WT[id(w)][4]() is w and _WT[id(w)][0] == w._version and _WT[id(w)][1] == w.data_ptr())]
    if not ws or not _gemm_mod.OVERLAP or torch.cuda.is_current_stream_capturing():
        return
    dev = ws[0].device
    cur, side = torch.cuda.current_stream(dev), _gemm_mod._side_stream(dev)
    side.wait_stream(cur)                       # (the optimizer's update of the weights was enqueued on `cur`)
    with torch.cuda.stream(side), torch.no_grad():
        buf = torch.empty(sum(w.numel() for w in ws), dtype=torch.float32, device=dev)
        descs, outs, o = (_WtDesc * len(ws))(), [], 0
        for i, w in enumerate(ws):
            t = buf[o:o + w.numel()]
            o += w.numel()
            outs.append(t)
            descs[i] = _WtDesc(w.data_ptr(), t.data_ptr(), w.shape[1], w.shape[0])
        _lib.call("cim_conv3x3_wt_multi", descs, len(ws), _lib.stream_ptr())
        ev = torch.cuda.Event()
        ev.record(side)
    for w, t in zip(ws, outs):
        key = id(w)
        _WT[key] = (w._version, w.data_ptr(), t, ev, weakref.ref(w, lambda _r, key=key: _WT.pop(key, None)))


class Conv3x3BnActFunction(Function):
    @staticmethod
    def forward(ctx, x, w, res, gamma, beta, mean, var, eps, relu, stride, dilation=1, in_bn=None, state=None):
        x = x.contiguous()
        w = w.contiguous()
        B, cin, H, W = x.shape
        cout = w.shape[0]
        ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        need_grad = any(ctx.needs_input_grad[:6])
        y = torch.empty((B, cout, ho, wo), dtype=torch.float32, device=x.device)
        xr = torch.empty_like(y) if need_grad else None            # convolution output: the BatchNorm backward's x
        if res is not None:
            res = res.contiguous()
        splits = _lib.call("cim_conv3x3_nchw_splits", cin, cout, H, W, stride)
        ws = torch.empty(splits * cout * ho * wo, dtype=torch.float32, device=x.device) if splits > 1 else None
        st = _lib.stream_ptr()
        for b in range(B):
            _lib.call("cim_conv3x3_nchw_f32", x[b].data_ptr(), w.data_ptr(), y[b].data_ptr(), cin, cout, H, W, stride, dilation,
                      _lib.ptr(xr[b] if xr is not None else None), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), var.data_ptr(),
                      float(eps), _lib.ptr(res[b] if res is not None else None), int(relu), splits, _lib.ptr(ws), st)
        if need_grad:
            ctx.save_for_backward(x, w, xr, y if relu else None, gamma, mean, var)
        ctx.param = w if isinstance(w, torch.nn.Parameter) else None      # (its .grad tells the backward whether it may defer the join)
        ctx.cfg = (B, cin, cout, H, W, float(eps), bool(relu), res is not None, stride, dilation)
        ctx.in_bn, ctx.state = in_bn, state          # ops/chain.py
        if state is not None:
            state["xr"] = xr                         # (a chaining consumer's epilogue needs it for this layer's dgamma)
        ctx.affine = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, xr, y, gamma, mean, var = ctx.saved_tensors
        B, cin, cout, H, W, eps, relu, has_res, stride, dilation = ctx.cfg
        is_dconv, dy_part = chain.take(dy, ctx.state)
        dy = dy.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_res = has_res and ctx.needs_input_grad[2]
        need_mean = ctx.needs_input_grad[5]          # a convolution bias folded into the mean (conv3x3_bn_act): d/dmean = -a sum dz
        need_affine = ctx.needs_input_grad[3] or ctx.needs_input_grad[4] or need_mean
        dev = dy.device
        hwo = dy.shape[2] * dy.shape[3]
        dres = torch.empty_like(dy) if need_res else None
        dgamma, dbeta, chained = affine_outputs(ctx, is_dconv, dy_part, need_affine, B, cout, hwo, dev)
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty_like(w) if need_w else None
        ws = torch.empty(_lib.call("cim_conv3x3_nchw_bwd_workspace", B, cin, cout, H, W, stride) // 4, dtype=torch.float32, device=dev)
        in_bn = ctx.in_bn if need_x else None
        if in_bn is not None and not chain.still_private(x, in_bn, ctx):
            in_bn = None                         # somebody looks at x's gradient / the producer's backward is not in this pass
        parts = _lib.call("cim_conv3x3_dx_parts", H, W, stride if dilation == 1 else 1)    # (stride 2: groups of the data gradient's pixel classes)
        in_part = torch.empty((B, 2, parts, cin), dtype=torch.float32, device=dev) if in_bn is not None and in_bn.affine else None
        side, ev_fork, ev_join, join = _gemm_mod.side_stream_for_backward(dev, ctx.param if (need_x and need_w) else None, ctx)
        wt = _transposed(ctx.param) if (need_x and ctx.param is not None) else None
        _lib.call("cim_conv3x3_nchw_bn_act_bwd", dy.data_ptr(), _lib.ptr(y), xr.data_ptr(), x.data_ptr(), w.data_ptr(),
                  gamma.data_ptr(), mean.data_ptr(), var.data_ptr(), eps, int(relu), _lib.ptr(dres),
                  _lib.ptr(None if chained else dgamma), _lib.ptr(None if chained else dbeta), _lib.ptr(dx), _lib.ptr(dw),
                  B, cin, cout, H, W, stride, dilation, ws.data_ptr(), _lib.stream_ptr(), side, ev_fork, ev_join, join,
                  int(is_dconv), *chain.c_args(in_bn, in_part), _lib.ptr(wt))
        if in_bn is not None:
            chain.hand_over(in_bn, dx, in_part)
        if not join:                       # the weight gradient is still running on the side stream: installed as .grad at the join
            # (everything the side stream reads stays referenced until then - a chained layer's handed-over `dy` included)
            _gemm_mod.defer_side_join(dev, ctx.param, dw, ws, x, *((dy,) if is_dconv else ()))
            dw = None
        dmean = -(gamma * torch.rsqrt(var + eps)) * dbeta if need_mean else None
        return dx, dw, dres, (dgamma if ctx.needs_input_grad[3] else None), (dbeta if ctx.needs_input_grad[4] else None), \
            dmean, None, None, None, None, None, None, None


def _apply(fn, args, n_diff, in_bn=None, tag=None):
    if not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in args[:n_diff])):
        with torch.no_grad():
            return fn.apply(*args)
    state = {"taken": False}
    out = fn.apply(*args, in_bn, state)
    if tag is not None:
        chain.tag(out, *tag, state)
    return out


def _same_padded(conv):
    d = conv.dilation[0]
    return (conv.kernel_size == (3, 3) and conv.dilation == (d, d) and conv.padding == (d, d) and conv.groups == 1
            and conv.stride in ((1, 1), (2, 2)) and (d == 1 or conv.stride == (1, 1)) and 1 <= d <= 8)


def _geometry_ok(x, conv, needs_backward):
    return (x.dtype == torch.float32 and x.dim() == 4 and _same_padded(conv) and conv.out_channels % 4 == 0
            and (conv.in_channels % 4 == 0 or not needs_backward)          # the RGB stems are frozen: forward only
            and x.shape[3] <= 4096 and x.shape[2] * x.shape[3] < (1 << 20))


def conv3x3_bn_act(x, conv, bn, residual=None, relu=True, fuse_input_bn=False):
    """relu?(bn(conv(x)) + residual) for a 3 x 3 nn.Conv2d `conv` ("same" padding, stride 1 / 2, dilation d with padding d)
    and an nn.BatchNorm2d `bn` in eval() mode.  A convolution bias (HRNet's downsamp_modules, HRNet.py:283-296) is folded
    into the BatchNorm's mean: bn(conv + bias) = a conv + (beta - (mean - bias) a)."""
    grad_on = torch.is_grad_enabled()
    needs_bwd = grad_on and (x.requires_grad or conv.weight.requires_grad)
    fused = (x.is_cuda and _geometry_ok(x, conv, needs_bwd)
             and (not bn.training) and bn.affine and bn.track_running_stats)
    if not fused:
        if x.is_cuda:
            fallback.note("conv3x3_bn_act", "BatchNorm in training mode" if bn.training else "unsupported geometry %s" % (conv,))
        out = bn(conv(x))
        if residual is not None:
            out = out + residual
        return F.relu(out) if relu else out
    mean = bn.running_mean if conv.bias is None else bn.running_mean - conv.bias
    args = (x, conv.weight, residual, bn.weight, bn.bias, mean, bn.running_var, bn.eps, relu, conv.stride[0], conv.dilation[0])
    in_bn = chain.input_bn(x, fuse_input_bn and torch.is_grad_enabled() and x.requires_grad)       # (ops/chain.py)
    return _apply(Conv3x3BnActFunction, args, 6, in_bn, (bn.weight, bn.bias, mean, bn.running_var, bn.eps, relu, residual is not None))


_IDENTITY_BN = {}      # (device, channels) -> (ones, zeros): BatchNorm statistics that make the epilogue y = x + beta exactly


def conv3x3_bias_act(x, conv, relu=True):
    """relu?(conv(x) + bias) for a 3 x 3 nn.Conv2d with a bias and no BatchNorm - the 13 convolutions of the dilated VGG16 body,
    /root/reference/lib/modeling/vgg16.py:34-78 (conv5: dilation 2) - on the implicit-GEMM kernel: the BatchNorm epilogue with
    gamma = 1, mean = 0, var = 1, eps = 0 (a = 1 exactly) and beta = the bias, whose gradient is the epilogue's dbeta."""
    needs_bwd = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)
    if not (x.is_cuda and conv.bias is not None and _geometry_ok(x, conv, needs_bwd)):
        if x.is_cuda:
            fallback.note("conv3x3_bias_act", "unsupported geometry %s" % (conv,))
        out = conv(x)
        return F.relu(out) if relu else out
    key = (x.device, conv.out_channels)
    if key not in _IDENTITY_BN:
        _IDENTITY_BN[key] = (torch.ones(conv.out_channels, device=x.device), torch.zeros(conv.out_channels, device=x.device))
    ones, zeros = _IDENTITY_BN[key]
    args = (x, conv.weight, None, ones, conv.bias, zeros, ones, 0.0, relu, conv.stride[0], conv.dilation[0])
    return _apply(Conv3x3BnActFunction, args, 6)


def conv7x7_bn_act(x, conv, bn, relu=True):
    """relu?(bn(conv(x))) for the FROZEN 7 x 7 stem convolution (padding 3, stride 1 / 2, no bias) and its eval-mode BatchNorm:
    one launch of the implicit-GEMM kernel with 49 taps.  A stem that still trains takes the ATen ops."""
    frozen = not (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad))
    fused = (frozen and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.kernel_size == (7, 7) and conv.bias is None
             and conv.padding == (3, 3) and conv.stride in ((1, 1), (2, 2)) and conv.dilation == (1, 1) and conv.groups == 1
             and x.shape[3] <= 4096 and x.shape[2] * x.shape[3] < (1 << 20)
             and (not bn.training) and bn.affine and bn.track_running_stats)
    if not fused:
        if x.is_cuda:
            fallback.note("conv7x7_bn_act", "stem not frozen / unsupported geometry")
        out = bn(conv(x))
        return F.relu(out) if relu else out
    with torch.no_grad():
        x = x.contiguous()
        w = conv.weight.contiguous()
        B, cin, H, W = x.shape
        stride, cout = conv.stride[0], w.shape[0]
        y = torch.empty((B, cout, (H - 1) // stride + 1, (W - 1) // stride + 1), dtype=torch.float32, device=x.device)
        for b in range(B):
            _lib.call("cim_conv7x7_nchw_f32", x[b].data_ptr(), w.data_ptr(), y[b].data_ptr(), cin, cout, H, W, stride,
                      bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps),
                      int(relu), _lib.stream_ptr())
    return y
