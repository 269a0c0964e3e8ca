"""Import shims that let /root/reference run on this CPU-only container.

Used ONLY by tests/golden/make_golden.py, in the build container, to capture golden vectors
by executing the reference itself.  Nothing here is reference code; every stub stands in for
a third-party package the image lacks (torchvision, mmcv, pynvml, chainer, torchsummary) or
for an API removed from current PyTorch / Python (torch._six, collections.Sequence).
"""
import collections
import collections.abc
import sys
import types

import numpy as np
import torch
import yaml

REF_ROOT = "/root/reference"


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install(roi_align_cls=None, resnet50_factory=None):
    collections.Sequence = collections.abc.Sequence
    collections.Mapping = collections.abc.Mapping
    collections.Iterable = collections.abc.Iterable          # lib/utils/misc.py:3
    _module("torch._six", string_classes=(str, bytes), int_classes=(int,))
    _module("pynvml", nvmlInit=lambda: None)

    def _unavailable(*a, **k):
        raise RuntimeError("box-IoU / box-NMS fallback is dead code on the CIM path")

    tv = _module("torchvision")
    tv.ops = _module("torchvision.ops", box_iou=_unavailable, nms=_unavailable)
    tv.models = _module("torchvision.models",
                        resnet50=resnet50_factory or _unavailable, vgg16=_unavailable)
    tv.transforms = _module("torchvision.transforms")
    tv.transforms.functional = _module("torchvision.transforms.functional")
    _module("torchsummary", summary=lambda *a, **k: None)

    _orig_load = yaml.load
    if not getattr(yaml.load, "_cim_shim", False):
        def _load(stream, Loader=None):
            return _orig_load(stream, Loader=Loader or yaml.FullLoader)
        _load._cim_shim = True
        yaml.load = _load

    torch.Tensor.cuda = lambda self, *a, **k: self           # heads.py:11 on a CPU-only host

    mm = _module("mmcv")
    mm.ops = _module("mmcv.ops", RoIAlign=roi_align_cls, RoIPool=None, roi_align=None,
                     roi_pool=None, nms=None, soft_nms=None)
    ch = _module("chainer")
    ch.backends = _module("chainer.backends")
    ch.backends.cuda = _module("chainer.backends.cuda", get_array_module=lambda *a: np)
    if REF_ROOT + "/lib" not in sys.path:
        sys.path.insert(0, REF_ROOT + "/lib")
