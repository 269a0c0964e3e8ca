"""Drop-in for /root/reference/lib/ops/__init__.py (which re-exports mmcv.ops).

Only RoIAlign / roi_align are on the CIM training path (model_builder.py:229-231); the other
names the reference re-exports are kept importable and raise if called (RoIPool is reachable
only through ROI_XFORM_METHOD=RoIPoolF, which no shipped config sets; nms / soft_nms are
eval-only box ops) - SURVEY.md section 2.2."""
from .bn_act import bn_act
from .conv1x1 import conv1x1_bn_act
from .conv3x3 import conv3x3_bias_act, conv3x3_bn_act, conv7x7_bn_act
from .pool import max_pool2d, upsample_nearest
from .roi_align import RoIAlign, roi_align, roi_align_maskcat


def _out_of_scope(name):
    def fn(*args, **kwargs):
        raise NotImplementedError("%s is not on the CIM training hot path and is not provided by cim_amd" % name)
    fn.__name__ = name
    return fn


class RoIPool:  # noqa: D401 - constructor-compatible placeholder
    def __init__(self, *args, **kwargs):
        raise NotImplementedError("RoIPool is not on the CIM training hot path (no shipped config uses RoIPoolF)")


roi_pool = _out_of_scope("roi_pool")
nms = _out_of_scope("nms")
soft_nms = _out_of_scope("soft_nms")

__all__ = ["RoIPool", "RoIAlign", "roi_pool", "roi_align", "nms", "soft_nms", "roi_align_maskcat", "bn_act", "conv1x1_bn_act", "conv3x3_bn_act", "conv3x3_bias_act", "conv7x7_bn_act", "max_pool2d", "upsample_nearest"]
