"""Max-pooling and nearest-neighbour up-sampling of the bodies on csrc/pool.hip.

`max_pool2d(x, module)` runs an `nn.MaxPool2d` (ResNet stem, /root/reference/lib/modeling/resnet50.py:29 through torchvision;
VGG16, /root/reference/lib/modeling/vgg16.py:43,50,60), `upsample_nearest(x, module)` an `nn.Upsample(mode='nearest')` with an
integer scale (HRNet fuse layers, /root/reference/lib/modeling/HRNet.py:201) - ATen's semantics (first maximum of a window wins,
NaN propagates; block sums in row order), NCHW fp32.  CPU tensors take the module itself (host-side tests of the model code);
a GPU tensor in a configuration the kernels do not take is a counted fallback (ops/fallback.py: an error unless allowed explicitly).
"""
import torch
from torch.autograd import Function

from .. import _lib
from . import fallback


class MaxPool2dFunction(Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        x = x.contiguous()
        n, c, h, w = x.shape
        ho = _lib.call("cim_maxpool2d_out_size", h, k, s, p)
        wo = _lib.call("cim_maxpool2d_out_size", w, k, s, p)
        y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
        idx = torch.empty((n, c, ho, wo), dtype=torch.int32, device=x.device) if ctx.needs_input_grad[0] else None
        _lib.call("cim_maxpool2d_fwd", x.data_ptr(), y.data_ptr(), _lib.ptr(idx), n * c, h, w, k, s, p, _lib.stream_ptr())
        ctx.geom = (n, c, h, w, k, s, p)
        if idx is not None:
            ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w, k, s, p = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
        _lib.call("cim_maxpool2d_bwd", dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), n * c, h, w, k, s, p, _lib.stream_ptr())
        return dx, None, None, None


class UpsampleNearestFunction(Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = x.contiguous()
        n, c, h, w = x.shape
        y = torch.empty((n, c, h * scale, w * scale), dtype=torch.float32, device=x.device)
        _lib.call("cim_upsample_nearest_fwd", x.data_ptr(), y.data_ptr(), n * c, h, w, scale, 0, _lib.stream_ptr())
        ctx.geom = (n, c, h, w, scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w, scale = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
        _lib.call("cim_upsample_nearest_bwd", dy.data_ptr(), dx.data_ptr(), n * c, h, w, scale, _lib.stream_ptr())
        return dx, None


def _one(v):
    if isinstance(v, (tuple, list)):
        return int(v[0]) if len(set(v)) == 1 else None
    return int(v)


def max_pool2d(x, m):
    """`m(x)` for an nn.MaxPool2d `m`."""
    if not x.is_cuda:
        return m(x)
    k, s, p, d = _one(m.kernel_size), _one(m.stride if m.stride is not None else m.kernel_size), _one(m.padding), _one(m.dilation)
    ok = (x.dim() == 4 and x.dtype == torch.float32 and None not in (k, s, p, d) and d == 1 and not m.ceil_mode
          and not m.return_indices and 1 <= k <= 7 and 2 * p <= k and x.size(2) + 2 * p >= k and x.size(3) + 2 * p >= k)
    if not ok:
        fallback.note("max_pool2d", "unsupported geometry %s" % (m,))
        return m(x)
    return MaxPool2dFunction.apply(x, k, s, p)


def upsample_nearest(x, m):
    """`m(x)` for an nn.Upsample(scale_factor = integer, mode = 'nearest') `m`."""
    if not x.is_cuda:
        return m(x)
    sf = m.scale_factor
    if isinstance(sf, (tuple, list)):
        sf = sf[0] if len(set(sf)) == 1 else None
    ok = (x.dim() == 4 and x.dtype == torch.float32 and m.mode == "nearest" and m.size is None and sf is not None
          and float(sf) == int(sf) and 1 <= int(sf) <= 64)
    if not ok:
        fallback.note("upsample_nearest", "unsupported module %s" % (m,))
        return m(x)
    return UpsampleNearestFunction.apply(x, int(sf))
