"""Frozen-statistics BatchNorm (+ residual) (+ ReLU) as one HIP launch each way (cim_amd/csrc/bn_act.hip).

`bn_act(x, bn, residual=None, relu=True)` == F.relu(bn(x) + residual) for an nn.BatchNorm2d in eval() mode - the state
every BN of the reference's bodies is kept in (/root/reference/lib/modeling/resnet50.py:53-77: running statistics,
trainable affine).  A BN in training mode, CPU tensors or non-fp32 data take the ATen formulation.
"""
import torch
import torch.nn.functional as F
from torch.autograd import Function

from .. import _lib
from . import fallback


class BnActFunction(Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, mean, var, eps, relu):
        x = x.contiguous()
        n, c = x.shape[0], x.shape[1]
        hw = x.numel() // (n * c)
        if res is not None:
            res = res.contiguous()
        y = torch.empty_like(x)
        _lib.call("cim_bn_act_fwd", x.data_ptr(), _lib.ptr(res), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                  var.data_ptr(), float(eps), y.data_ptr(), n, c, hw, int(relu), _lib.stream_ptr())
        ctx.save_for_backward(x, y if relu else None, gamma, mean, var)
        ctx.cfg = (n, c, hw, float(eps), bool(relu), res is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, var = ctx.saved_tensors
        n, c, hw, eps, relu, has_res = ctx.cfg
        dy = dy.contiguous()
        need_x, need_res = ctx.needs_input_grad[0], has_res and ctx.needs_input_grad[1]
        need_affine = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        dx = torch.empty_like(x) if need_x else None
        dres = torch.empty_like(x) if need_res else None
        dgamma = dbeta = None
        if need_affine:      # accumulated with atomics when the launch splits a channel over several workgroups
            alloc = torch.zeros if _lib.call("cim_bn_act_bwd_chunks", n, c, hw) > 1 else torch.empty
            dgamma, dbeta = alloc(2, c, dtype=torch.float32, device=x.device).unbind(0)
        _lib.call("cim_bn_act_bwd", dy.data_ptr(), _lib.ptr(y), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                  var.data_ptr(), eps, _lib.ptr(dx), _lib.ptr(dres), _lib.ptr(dgamma), _lib.ptr(dbeta), n, c, hw,
                  int(relu), _lib.stream_ptr())
        return dx, dres, (dgamma if ctx.needs_input_grad[2] else None), (dbeta if ctx.needs_input_grad[3] else None), \
            None, None, None, None


def bn_act(x, bn, residual=None, relu=True):
    """relu?(bn(x) + residual) for a BatchNorm2d module `bn`."""
    fused = (not bn.training) and x.is_cuda and x.dtype == torch.float32 and bn.affine and bn.track_running_stats \
        and x.dim() == 4 and x.is_contiguous()
    if not fused:
        if x.is_cuda:
            fallback.note("bn_act", "BatchNorm in training mode" if bn.training else "unsupported layout / dtype")
        out = bn(x)
        if residual is not None:
            out = out + residual
        return F.relu(out) if relu else out
    if not (torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad
                                        or (residual is not None and residual.requires_grad))):
        with torch.no_grad():
            return BnActFunction.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)
    return BnActFunction.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, relu)
