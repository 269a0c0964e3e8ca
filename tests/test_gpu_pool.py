"""csrc/pool.hip against ATen: the max-pools of the ResNet stem / VGG16 and HRNet's nearest up-sampling (SURVEY.md a-11;
/root/reference/lib/modeling/resnet50.py:29, vgg16.py:43,50,60, HRNet.py:201).  Bit-exact: a maximum is exact, and the backward
tests use integer-valued gradients (sums of small integers are exact in fp32 whatever the order - ATen's backward scatters with
atomics) next to a random-gradient case held to 1e-6."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("shape,k,s,p", [((1, 64, 258, 344), 3, 2, 1), ((2, 3, 7, 9), 3, 2, 1), ((1, 5, 33, 47), 2, 2, 0),
                                         ((1, 128, 130, 173), 2, 2, 0), ((1, 2, 8, 8), 3, 1, 1), ((1, 1, 5, 6), 5, 3, 2)])
def test_max_pool_forward_backward_equal_aten(shape, k, s, p):
    from cim_amd.ops import max_pool2d
    dev = _dev()
    g = torch.Generator().manual_seed(sum(shape) + k)
    for ties in (False, True):
        x = torch.randint(-3, 4, shape, generator=g).float() if ties else torch.randn(shape, generator=g)
        x = x.to(dev).requires_grad_(True)
        xr = x.detach().clone().requires_grad_(True)
        m = nn.MaxPool2d(k, s, p)
        y = max_pool2d(x, m)
        yr = m(xr)
        assert y.shape == yr.shape and torch.equal(y, yr)
        dy = torch.randint(-4, 5, y.shape, generator=g).float().to(dev)
        y.backward(dy)
        yr.backward(dy)
        assert torch.equal(x.grad, xr.grad), (ties, float((x.grad - xr.grad).abs().max()))
    # random gradients: equal up to the order of <= 4 additions per pixel
    x = torch.randn(shape, generator=g).to(dev).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    dy = torch.randn(max_pool2d(x, m).shape, generator=g).to(dev)
    max_pool2d(x, m).backward(dy)
    m(xr).backward(dy)
    assert float((x.grad - xr.grad).abs().max()) <= 1e-6 * float(dy.abs().max()) * 4


def test_max_pool_nan_and_no_grad():
    from cim_amd.ops import max_pool2d
    dev = _dev()
    x = torch.randn(1, 4, 9, 11, device=dev)
    x[0, 1, 4, 5] = float("nan")
    m = nn.MaxPool2d(3, 2, 1)
    y, yr = max_pool2d(x, m), m(x)
    assert torch.equal(torch.isnan(y), torch.isnan(yr)) and torch.equal(torch.nan_to_num(y), torch.nan_to_num(yr))
    assert not y.requires_grad


def test_max_pool_unsupported_module_is_a_counted_fallback():
    from cim_amd import _lib
    from cim_amd.ops import fallback, max_pool2d
    x = torch.randn(1, 2, 9, 9, device=_dev())
    m = nn.MaxPool2d(3, 2, 1, ceil_mode=True)
    with pytest.raises(_lib.CimHipError):             # a library branch is an error by default
        max_pool2d(x, m)
    with fallback.allowed("max_pool2d"):
        assert torch.equal(max_pool2d(x, m), m(x))


@pytest.mark.parametrize("shape,scale", [((1, 96, 17, 23), 2), ((1, 192, 9, 12), 4), ((2, 7, 5, 6), 8), ((1, 3, 4, 4), 1)])
def test_upsample_nearest_forward_backward_equal_aten(shape, scale):
    from cim_amd.ops import upsample_nearest
    dev = _dev()
    g = torch.Generator().manual_seed(sum(shape) + scale)
    x = torch.randn(shape, generator=g).to(dev).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    m = nn.Upsample(scale_factor=scale, mode="nearest")
    y, yr = upsample_nearest(x, m), m(xr)
    assert y.shape == yr.shape and torch.equal(y, yr)
    dy = torch.randint(-4, 5, y.shape, generator=g).float().to(dev)
    y.backward(dy)
    yr.backward(dy)
    assert torch.equal(x.grad, xr.grad)
    x.grad = None
    xr.grad = None
    dy = torch.randn(y.shape, generator=g).to(dev)
    upsample_nearest(x, m).backward(dy)
    m(xr).backward(dy)
    assert float((x.grad - xr.grad).abs().max()) <= 1e-6 * float(dy.abs().max()) * scale * scale


def test_bodies_take_no_library_pooling(monkeypatch):
    """The three bodies' forward passes (tiny inputs) must not reach F.max_pool2d / F.interpolate on a GPU tensor."""
    from cim_amd.core.presets import apply_preset
    from cim_amd.modeling.model_builder import get_func
    from cim_amd.core.config import cfg
    dev = _dev()

    def boom(*a, **k):
        raise AssertionError("library pooling / interpolation reached from a body's forward")
    for preset, size in (("resnet50_voc", (1, 3, 64, 96)), ("vgg16_voc", (1, 3, 64, 96)), ("hrnet48_voc", (1, 3, 64, 96))):
        try:
            apply_preset(preset)
        except KeyError:
            continue
        torch.manual_seed(0)
        body = get_func(cfg.MODEL.CONV_BODY)().to(dev).train()
        x = torch.randn(size, device=dev)
        with monkeypatch.context() as mp:
            mp.setattr(F, "max_pool2d", boom)
            mp.setattr(F, "interpolate", boom)
            y = body(x)
        assert torch.isfinite(y).all()
        if y.requires_grad:
            y.sum().backward()
    apply_preset("resnet50_voc")
