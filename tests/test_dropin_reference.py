"""b-1: the reference's own driver resolves to cim_amd after `install_as_lib()`.

Runs only where /root/reference exists (the build container).  A child process (the shims are process-global)
executes the import block of tools/train.py:25-39 exactly as the driver does, loads
configs/resnet50_voc.yaml through the REFERENCE's cfg_from_file, builds `Generalized_RCNN()` and runs one
CPU-side check of the shared config.  Third-party packages the image lacks (cv2, pycocotools, the reference's
Cython extensions compiled for py3.6/3.7) are stubbed by the test; nothing of cim_amd is stubbed."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r'''
import sys, types
sys.path.insert(0, %(repo)r)
sys.path.insert(0, %(repo)r + "/tests/golden")
import _ref_shims
_ref_shims.install()                       # torchvision / mmcv / pynvml / removed-API stubs (third party only)

def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m
for name, attrs in (("cv2", dict(setNumThreads=lambda n: None)), ("utils.cython_bbox", dict(bbox_overlaps=None)),
                    ("utils.cython_nms", dict(nms=None, soft_nms=None)),
                    ("pycocotools", {}), ("pycocotools.mask", {}), ("pycocotools.coco", dict(COCO=object)),
                    ("pycocotools.cocoeval", dict(COCOeval=object)), ("tqdm", dict(tqdm=lambda x, **k: x))):
    try:
        __import__(name)
    except Exception:
        stub(name, **attrs)

import cim_amd
cim_amd.install_as_lib()                   # the ONE line INTEGRATION.md section 2 adds to tools/train.py
sys.path.insert(0, %(ref)r + "/tools")

# ---- tools/train.py:25-39, verbatim order
import _init_paths
import nn as mynn
import utils.net as net_utils
import utils.misc as misc_utils
from core.config import cfg, cfg_from_file, cfg_from_list, assert_and_infer_cfg
from datasets.roidb import combined_roidb_for_training
from roi_data.loader import RoiDataLoader, MinibatchSampler, BatchSampler, collate_minibatch
from modeling.model_builder import Generalized_RCNN
from utils.detectron_weight_helper import load_detectron_weight
from utils.logging import setup_logging
from utils.timer import Timer
from utils.training_stats import TrainingStats

import cim_amd.core.config as own
import modeling.heads, modeling.resnet50, modeling.vgg16, modeling.HRNet, ops
assert mynn.DataParallel.__module__ == "cim_amd.nn.parallel.data_parallel", mynn.DataParallel.__module__
assert Generalized_RCNN.__module__ == "cim_amd.modeling.model_builder"
assert modeling.heads.__name__ == "cim_amd.modeling.heads" and ops.__name__ == "cim_amd.ops"
assert modeling.HRNet.__name__ == "cim_amd.modeling.HRNet"
for mod in (net_utils, misc_utils, sys.modules["utils.training_stats"], sys.modules["core.config"],
            sys.modules["roi_data.loader"], sys.modules["datasets.roidb"], sys.modules["nn.init"]):
    assert mod.__file__.startswith(%(ref)r + "/lib/"), mod.__file__          # the reference's own files
assert hasattr(net_utils, "load_ckpt") and hasattr(net_utils, "save_ckpt") or hasattr(net_utils, "load_ckpt")
assert own.cfg.target() is cfg                                                # ONE cfg object

# ---- tools/train.py:180-234 (config), :275 (model), :344 (wrapper)
cfg.MODEL.NUM_CLASSES = 20                                                    # train.py:185-190
cfg_from_file(%(ref)r + "/configs/resnet50_voc.yaml")
cfg_from_list(["MODEL.LOAD_IMAGENET_PRETRAINED_WEIGHTS", "False"])
cfg.NUM_GPUS = 1
assert_and_infer_cfg()                                                        # freezes the reference cfg
assert own.cfg.MODEL.CONV_BODY == "resnet50.torch_resnet50" and own.cfg.Anti_noise_sampling is True
assert own.cfg.MODEL is cfg.MODEL and own.cfg.FAST_RCNN.ROI_XFORM_METHOD == "RoIAlign"
model = Generalized_RCNN()
assert type(model.Conv_Body).__module__ == "cim_amd.modeling.resnet50"
assert type(model.Box_Head).__module__ == "cim_amd.modeling.maskfuse"
assert type(model.cls_iou_model).__module__ == "cim_amd.modeling.heads"
assert [l.cls_thr for l in model.CIM_layer_list] == [0.25, 0.35, 0.45]
# parameter grouping of train.py:282-305 ('bias' in name) and the wrapper of :344
bias = [n for n, p in model.named_parameters() if p.requires_grad and "bias" in n]
assert "Box_Head.seg_fc.0.bias" in bias and "cls_iou_model.refine_iou.2.bias" in bias
dp = mynn.DataParallel(model, cpu_keywords=["im_info", "roidb"], minibatch=True)
assert dp.module is model
stats = TrainingStats(types.SimpleNamespace(disp_interval=20, no_save=True), 20, None)
print("DROPIN-OK")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_train_py_import_block_resolves_to_cim_amd():
    code = PROBE % dict(repo=REPO, ref=REF)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DROPIN-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_install_uninstall_roundtrip():
    import cim_amd
    from cim_amd.core import config as own
    cim_amd.install_as_lib()
    try:
        import ops                                  # noqa: F401  resolves without the reference tree
        assert sys.modules["ops"].__name__ == "cim_amd.ops"
        import modeling.heads as h                  # noqa: F401  parent package `modeling` may not exist: alias only
    except ModuleNotFoundError:
        pass                                        # without lib/ on sys.path the parent package is absent - fine
    finally:
        cim_amd.uninstall_as_lib()
    assert "ops" not in sys.modules
    assert own.cfg.target() is own._own
