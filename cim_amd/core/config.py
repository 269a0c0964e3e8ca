"""Minimal global config for the CIM training step.

Mirrors the part of /root/reference/lib/core/config.py the hot path reads (`cfg` AttrDict,
`cfg_from_file`, `cfg_from_list`, `assert_and_infer_cfg`): the keys below are the ones
`Generalized_RCNN`, the backbones and `MaskFuse` consume (config.py:84,251,357-375,387,423,437,
459,469,483,539-551).  The reference's seven YAML files load unchanged; keys this build does
not model (TEST.*, SOLVER.*, dataset paths ...) are accepted and stored as-is.
"""
import copy

import yaml


class AttrDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def _defaults():
    c = AttrDict()
    c.TRAIN = AttrDict(FREEZE_CONV_BODY=False, SCALES=(600,), MAX_SIZE=1000, IMS_PER_BATCH=1,
                       BATCH_SIZE_PER_IM=4096)
    c.TEST = AttrDict(SCALE=600, MAX_SIZE=1000,           # lib/core/config.py:123-126,167-201
                      BBOX_AUG=AttrDict(ENABLED=False, SCORE_HEUR="AVG", COORD_HEUR="ID", H_FLIP=False, SCALES=(),
                                        MAX_SIZE=4000, SCALE_H_FLIP=False, SCALE_SIZE_DEP=False, ASPECT_RATIOS=(),
                                        ASPECT_RATIO_H_FLIP=False))
    c.MODEL = AttrDict(TYPE="generalized_rcnn", CONV_BODY="", NUM_CLASSES=-1,
                       LOAD_IMAGENET_PRETRAINED_WEIGHTS=False, EXTRA=AttrDict())
    c.SOLVER = AttrDict(BASE_LR=0.0005, WEIGHT_DECAY=0.0005, MOMENTUM=0.9, TYPE="SGD",
                        BIAS_DOUBLE_LR=True, BIAS_WEIGHT_DECAY=False,
                        # schedule keys read by cim_amd/optim/solver.py (defaults of lib/core/config.py:277-343)
                        LR_POLICY="step", GAMMA=0.1, STEPS=[], MAX_ITER=40000,
                        WARM_UP_ITERS=500, WARM_UP_FACTOR=1.0 / 3.0, WARM_UP_METHOD="linear",
                        SCALE_MOMENTUM=True, SCALE_MOMENTUM_THRESHOLD=1.1, LOG_LR_CHANGE_THRESHOLD=1.1)
    c.FAST_RCNN = AttrDict(ROI_BOX_HEAD="", MLP_HEAD_DIM=1024, ROI_XFORM_METHOD="RoIPoolF",
                           ROI_XFORM_SAMPLING_RATIO=0, ROI_XFORM_RESOLUTION=14, MASK_SIZE=7)
    c.VGG = AttrDict(IMAGENET_PRETRAINED_WEIGHTS="", FREEZE_AT=2)
    c.ResNet = AttrDict(IMAGENET_PRETRAINED_WEIGHTS="", FREEZE_AT=2)
    c.HRNET = AttrDict(IMAGENET_PRETRAINED_WEIGHTS="", FREEZE_AT=2)
    c.NUM_GPUS = 1
    c.REFINE_TIMES = 3
    c.DEDUP_BOXES = 1.0 / 8.0
    c.RNG_SEED = 3
    c.VGG_CLS_FEATURE = False
    c.ResNet_CLS_FEATURE = False
    c.HRNET_CLS_FEATURE = False
    c.Anti_noise_sampling = False
    c.p_seed = 0.1
    c.step_rate = 0.0
    c.transform_mode = "org"
    c.iou_dir = ""
    c.asy_iou_dir = ""
    c.PYTORCH_VERSION_LESS_THAN_040 = False
    return c


class _CfgProxy:
    """The `cfg` every cim_amd module imports.  It forwards every access to a TARGET AttrDict: this
    module's own defaults, or - after `bind(reference_cfg)`, which `cim_amd.install_as_lib()` calls when
    the reference's `core.config` gets imported - the reference's own global `cfg` object
    (lib/core/config.py:22-23).  So `tools/train.py` and this package read and write ONE config: nothing
    is copied, and modules that did `from ..core.config import cfg` before the binding see it too."""

    __slots__ = ("_target",)

    def __init__(self, target):
        object.__setattr__(self, "_target", target)

    def bind(self, target):
        object.__setattr__(self, "_target", target)

    def target(self):
        return self._target

    def __getattr__(self, name):
        return getattr(self._target, name)

    def __setattr__(self, name, value):
        setattr(self._target, name, value)

    def __getitem__(self, k):
        return self._target[k]

    def __setitem__(self, k, v):
        self._target[k] = v

    def __contains__(self, k):
        return k in self._target

    def __iter__(self):
        return iter(self._target)

    def __len__(self):
        return len(self._target)

    def __repr__(self):
        return "cfg -> " + repr(self._target)


_own = _defaults()
cfg = _CfgProxy(_own)
__C = cfg


def reset_cfg():
    """Back to this module's own defaults (also undoes a binding to the reference's cfg)."""
    _own.clear()
    _own.update(_defaults())
    cfg.bind(_own)


def _to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: _to_attr(v) for k, v in d.items()})
    return d


def _merge(a, b):
    for k, v in a.items():
        if isinstance(v, dict) and isinstance(b.get(k), dict) and (b[k] or k not in ("EXTRA",)):
            _merge(v, b[k])
        else:
            if isinstance(v, str) and isinstance(b.get(k), tuple):
                v = tuple(yaml.safe_load(v.replace("(", "[").replace(")", "]")))
            b[k] = copy.deepcopy(v)


def merge_cfg_from_file(cfg_filename):
    with open(cfg_filename, "r") as f:
        y = _to_attr(yaml.safe_load(f))
    _merge(y, cfg.target())


def merge_cfg_from_list(cfg_list):
    assert len(cfg_list) % 2 == 0
    for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
        d = cfg.target()
        keys = full_key.split(".")
        for sub in keys[:-1]:
            assert sub in d, "Non-existent key: {}".format(full_key)
            d = d[sub]
        assert keys[-1] in d, "Non-existent key: {}".format(full_key)
        if isinstance(v, str):
            try:
                v = yaml.safe_load(v)
            except yaml.YAMLError:
                pass
        d[keys[-1]] = v


cfg_from_file = merge_cfg_from_file
cfg_from_list = merge_cfg_from_list


def assert_and_infer_cfg(make_immutable=True):
    """The reference freezes the AttrDict here (config.py:652-672); this loader only validates."""
    assert cfg.MODEL.NUM_CLASSES > 0, "set cfg.MODEL.NUM_CLASSES (20 for VOC, 80 for COCO) as tools/train.py:185-190 does"
