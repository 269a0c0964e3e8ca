"""1 x 1 convolution + frozen-statistics BatchNorm (+ residual) (+ ReLU) of the ResNet bottlenecks as ONE HIP launch
(cim_amd/csrc/conv1x1.hip), with autograd.

`conv1x1_bn_act(x, conv, bn, residual=None, relu=True)` == F.relu(bn(conv(x)) + residual) for an nn.Conv2d with a
1 x 1 kernel (no bias, no padding; a stride is applied by sub-sampling the input first) and an nn.BatchNorm2d in eval()
mode (fuse_input_bn=True: see ops/chain.py) - /root/reference/lib/modeling/resnet50.py:17-44 (torchvision Bottleneck conv1 / conv3 / downsample) with every
BatchNorm frozen as :53-77 does.  The parameters stay the modules' own tensors (checkpoint surface unchanged).
In NCHW the convolution of one image is W[Cout,Cin] . X[Cin,HW]: forward, data gradient (W^T . dY) and weight gradient
(dY . X^T, split-K) are three layouts of the same small-tile fp32-MFMA GEMM; the BatchNorm backward stays the fused
`bn_act` kernel.  CPU tensors, a BatchNorm in training mode or other convolution shapes take the ATen ops."""
import ctypes
import weakref

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
    def forward(ctx, x, w, res, gamma, beta, mean, var, eps, relu, in_bn=None, state=None, recv=None, send=None, send_dx=None):
        ctx.send_dx = send_dx                    # (token, full-resolution width or 0): this layer's OWN data gradient goes to the token
        x = x.contiguous()
        B, cin, H, W = x.shape
        cout, hw = w.shape[0], H * W
        w2 = w.reshape(cout, cin)
        ctx.recv, ctx.send = recv, send          # branch token (see conv1x1_bn_act): receive / send the identity path's gradient
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
        if state is not None:
            state["xr"] = xr                         # (a chaining consumer's epilogue needs it for this layer's dgamma)
        ctx.affine = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2, xr, y, gamma, mean, var = ctx.saved_tensors
        B, cin, cout, H, W, eps, relu, has_res = ctx.cfg
        hw = H * W
        is_dconv, dy_part = chain.take(dy, ctx.state)       # the consumer's data gradient already applied this layer's BatchNorm + ReLU backward
        dy = dy.contiguous()
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_res = has_res and ctx.needs_input_grad[2]
        need_mean = ctx.needs_input_grad[5]          # a convolution bias folded into the mean: d/dmean = -a sum dz
        need_affine = ctx.needs_input_grad[3] or ctx.needs_input_grad[4] or need_mean
        # ONE host call: BatchNorm / ReLU backward, data gradient, weight gradient (+ split-K reduces)  (csrc/conv1x1.hip)
        dev = dy.device
        dres = torch.empty_like(dy) if need_res else None
        dx_add, dx_add_w = None, 0
        if ctx.recv is not None and need_x:     # the other branch's gradient of x, stashed by a layer whose backward runs before this one:
            ctx.recv["received"] = ctx.recv.get("received", 0) + 1      # the block's last layer (identity path) or its downsample layer
            dx_add = ctx.recv.pop("dres", None)
            if dx_add is not None and dx_add.shape != x.shape:          # a stride-2 downsample layer's: every second pixel
                if tuple(dx_add.shape) != (B, cin, (H + 1) // 2, (W + 1) // 2):
                    raise RuntimeError("cim_amd: branch hand-over of a gradient of shape %s to an input of shape %s"
                                       % (tuple(dx_add.shape), tuple(x.shape)))
                dx_add_w = W
        dgamma, dbeta, chained = affine_outputs(ctx, is_dconv, dy_part, need_affine, B, cout, hw, dev)
        dx = torch.empty_like(x) if need_x else None
        dw = torch.empty((cout, cin, 1, 1), dtype=torch.float32, device=dev) if need_w else None
        ws = torch.empty(_lib.call("cim_conv1x1_bwd_workspace", B, cin, cout, hw) // 4, dtype=torch.float32, device=dev)
        in_bn = ctx.in_bn if need_x else None
        if in_bn is not None and not chain.still_private(x, in_bn, ctx):
            in_bn = None                         # somebody looks at x's gradient / the producer's backward is not in this pass
        in_part = torch.empty((B, 2, (hw + 31) // 32, cin), dtype=torch.float32, device=dev) if in_bn is not None and in_bn.affine else None
        side, ev_fork, ev_join, join = _gemm_mod.side_stream_for_backward(dev, ctx.param if (need_x and need_w) else None, ctx)
        _lib.call("cim_conv1x1_bn_act_bwd", dy.data_ptr(), _lib.ptr(y), xr.data_ptr(), x.data_ptr(), w2.data_ptr(),
                  gamma.data_ptr(), mean.data_ptr(), var.data_ptr(), eps, int(relu), _lib.ptr(dres),
                  _lib.ptr(None if chained else dgamma), _lib.ptr(None if chained else dbeta), _lib.ptr(dx), _lib.ptr(dw),
                  B, cin, cout, hw, ws.data_ptr(), _lib.stream_ptr(), side, ev_fork, ev_join, join, int(is_dconv), *chain.c_args(in_bn, in_part),
                  _lib.ptr(dx_add), dx_add_w)
        if ctx.send_dx is not None and dx is not None and _receiver_runs(ctx.send_dx, ctx):
            # a downsample layer: its data gradient is the SECOND gradient of the block's input - handed to the block's first layer
            # (whose backward runs later: checked) instead of autograd, which would scatter a stride-2 layer's into a zero-filled
            # tensor through two slice nodes (two fills + two copies) and add the two gradients with a launch of its own
            tok = ctx.send_dx
            if tok.get("sent", 0) != tok.get("received", 0):
                raise RuntimeError("cim_amd: the bottleneck's first layer ran its backward before its downsample layer - branch hand-over out of order")
            tok["sent"] = tok.get("sent", 0) + 1
            tok["dres"] = dx
            dx = None
        if ctx.send is not None and dres is not None and _receiver_runs(ctx.send, ctx):
            # hand the identity path's gradient to the block's FIRST layer instead of autograd (which would add it to that layer's
            # data gradient with a launch of its own); that layer's backward runs after this one (later nodes first) - checked
            if ctx.send.get("sent", 0) != ctx.send.get("received", 0):
                raise RuntimeError("cim_amd: the bottleneck's first layer ran its backward before its last one - branch hand-over out of order")
            ctx.send["sent"] = ctx.send.get("sent", 0) + 1
            ctx.send["dres"] = dres
            dres = None
        if in_bn is not None:
            chain.hand_over(in_bn, dx, in_part)
        if not join:                       # the weight gradient is still running on the side stream: installed as .grad at the join
            # (everything the side stream reads stays referenced until then - the handed-over gradient `dy` of a chained layer
            # included: autograd drops it when this node returns)
            _gemm_mod.defer_side_join(dev, ctx.param, dw, ws, x, *((dy,) if is_dconv else ()))
            dw = None
        dmean = -(gamma * torch.rsqrt(var + eps)) * dbeta if need_mean else None
        return dx, dw, dres, (dgamma if ctx.needs_input_grad[3] else None), (dbeta if ctx.needs_input_grad[4] else None), \
            dmean, None, None, None, None, None, None, None, None


def _receiver_runs(tok, node):
    """Branch hand-over: the block's first layer takes the second gradient of the block's input in its data-gradient epilogue - only
    in a complete .backward() pass in which its node runs (torch.autograd.grad towards this layer's weight alone never reaches
    it); else the gradient goes back to autograd as usual."""
    r = tok.get("recv_node")
    return chain.node_runs(r() if r is not None else None) and not chain.restricted_pass(node)


class _BnPartDesc(ctypes.Structure):          # cim_bn_part_desc of include/cim_hip.h
    _fields_ = [("part", ctypes.c_void_p), ("var", ctypes.c_void_p), ("eps", ctypes.c_float), ("dgamma", ctypes.c_void_p),
                ("dbeta", ctypes.c_void_p), ("images", ctypes.c_int), ("parts", ctypes.c_int), ("channels", ctypes.c_int)]


def finish_affine(records):
    """ONE launch (per 24 layers) for the affine gradients of chained layers: records = (partial sums [B][2][parts][C], var, eps,
    gamma or None, beta or None) -> [(parameter, gradient)].  Runs on the current stream."""
    descs, out = (_BnPartDesc * len(records))(), []
    for i, (part, var, eps, g, b) in enumerate(records):
        B, _, parts, C = part.shape
        dg, db = torch.empty(2, C, dtype=torch.float32, device=part.device).unbind(0)
        descs[i] = _BnPartDesc(part.data_ptr(), var.data_ptr(), eps, dg.data_ptr() if g is not None else None,
                               db.data_ptr() if b is not None else None, B, parts, C)
        out += [(q, t) for q, t in ((g, dg), (b, db)) if q is not None]
    _lib.call("cim_bn_part_finish", descs, len(records), _lib.stream_ptr())
    return out


def affine_outputs(ctx, is_dconv, dy_part, need_affine, B, cout, hw, dev):
    """-> (dgamma, dbeta buffers for the call's own BatchNorm backward or None, chained).  A CHAINED layer's affine gradients are
    finished from the partial sums its consumer left (`dy_part`): at the end of the backward pass, one launch for all layers
    (ops/gemm.py: defer_finisher; they are then installed as .grad like the deferred weight gradients and do not travel through
    autograd) - or at once, returned through autograd, when nothing can be deferred (no engine callback, graph capture, affine
    tensors that are not Parameters)."""
    if not need_affine:
        return None, None, False
    if is_dconv:
        if dy_part is None:
            raise RuntimeError("cim_amd: chained BatchNorm backward without the partial sums of its trainable affine parameters")
        g, b = ctx.affine
        g, b = (g if ctx.needs_input_grad[3] else None), (b if ctx.needs_input_grad[4] else None)
        _, _, _, _, gamma, mean, var = ctx.saved_tensors
        rec = (dy_part, var, ctx.cfg[5], g, b)
        if (_gemm_mod.DEFER_DW and not torch.cuda.is_current_stream_capturing()
                and all(q is None or isinstance(q, torch.nn.Parameter) for q in (g, b))
                and not chain.restricted_pass(ctx)):
            _gemm_mod.defer_finisher(dev, finish_affine, rec, [q for q in (g, b) if q is not None])
            return None, None, True
        got = dict((id(q), t) for q, t in finish_affine([rec]))
        return got.get(id(g)), got.get(id(b)), True
    alloc = torch.zeros if _lib.call("cim_bn_act_bwd_chunks", B, cout, hw) > 1 else torch.empty
    dgamma, dbeta = alloc(2, cout, dtype=torch.float32, device=dev).unbind(0)
    return dgamma, dbeta, False


def conv1x1_bn_act(x, conv, bn, residual=None, relu=True, fuse_input_bn=False, branch=None):
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
    if (fuse_input_bn and branch is not None and residual is None and getattr(x, "_cim_bn", None) is not None
            and torch.is_grad_enabled() and x.requires_grad):
        raise ValueError("conv1x1_bn_act: `branch` on a layer without a residual makes it the RECEIVER of a second gradient of its "
                         "input (added in its data-gradient epilogue); that cannot be combined with fuse_input_bn=True on an input "
                         "whose producer is chained (the same epilogue slot) - pass one of the two")
    in_bn = chain.input_bn(x, fuse_input_bn and stride == 1 and torch.is_grad_enabled() and x.requires_grad)
    send_dx = None
    same = lambda t: branch.get("x") is not None and branch["x"]() is t       # the very tensor object the first layer registered
    if (branch is not None and residual is None and stride in (1, 2) and x.requires_grad and x.is_contiguous() and same(x)):
        send_dx = branch            # the block's downsample layer, called after the first layer registered the same input
    if stride != 1:                                   # a strided 1 x 1 convolution only sees every stride-th pixel
        x = x[:, :, ::stride, ::stride]
    mean = bn.running_mean if conv.bias is None else bn.running_mean - conv.bias
    args = (x, conv.weight, residual, bn.weight, bn.bias, mean, bn.running_var, bn.eps, relu)
    if not (torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in args[:6])):
        with torch.no_grad():
            return Conv1x1BnActFunction.apply(*args)
    # `branch`: a dict shared by the first (no residual) and the last (residual = the block's input) 1 x 1 layer of a bottleneck whose
    # identity path is the input itself: the last layer's backward hands the identity gradient to the first layer's data-gradient
    # epilogue (dx_add) instead of returning it to autograd - one add launch per block less.  Only between calls that see the
    # SAME tensor, with stride 1, when both need its gradient.
    recv = send = None
    if branch is not None and send_dx is None and stride == 1 and x.requires_grad:
        if residual is None:
            branch["x"] = weakref.ref(x)          # identity, not address: a different autograd tensor over the same storage is not x
            recv = branch
        elif same(residual) and residual.requires_grad and residual.is_contiguous():
            send = branch
    state = {"taken": False}
    out = Conv1x1BnActFunction.apply(*args, in_bn, state, recv, send, send_dx)
    if recv is not None and out.grad_fn is not None:
        recv["recv_node"] = weakref.ref(out.grad_fn)      # the senders ask the engine whether this node is part of their pass
    chain.tag(out, bn.weight, bn.bias, mean, bn.running_var, bn.eps, relu, residual is not None, state)
    return out
