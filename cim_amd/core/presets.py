"""The BASELINE.json configurations as programmatic config overrides.

Values are the hyper-parameters of /root/reference/configs/{vgg16_voc,resnet50_voc,
resnet50_coco2017,hrnet48_coco2017}.yaml (which also load unchanged through
`cfg_from_file`); they are restated here because /root/reference does not exist on the GPU box.
"""
from .config import cfg, reset_cfg

_COMMON = dict(REFINE_TIMES=3, DEDUP_BOXES=0.0, transform_mode="ToTensor", step_rate=0.1,
               Anti_noise_sampling=True, p_seed=0.1)
_FAST_RCNN = dict(ROI_XFORM_METHOD="RoIAlign", ROI_XFORM_RESOLUTION=7, MLP_HEAD_DIM=4096, MASK_SIZE=7,
                  ROI_XFORM_SAMPLING_RATIO=0)

PRESETS = {
    "vgg16_voc": dict(CONV_BODY="vgg16.dilated_conv5_body", ROI_BOX_HEAD="vgg16.MaskFuse", NUM_CLASSES=20,
                      flags=dict(VGG_CLS_FEATURE=True)),
    "resnet50_voc": dict(CONV_BODY="resnet50.torch_resnet50", ROI_BOX_HEAD="resnet50.MaskFuse", NUM_CLASSES=20,
                         flags={}),
    "hrnet48_voc": dict(CONV_BODY="HRNet.get_HRNet", ROI_BOX_HEAD="HRNet.MaskFuse", NUM_CLASSES=20,
                        flags=dict(HRNET_CLS_FEATURE=True)),
    "hrnet48_coco2017": dict(CONV_BODY="HRNet.get_HRNet", ROI_BOX_HEAD="HRNet.MaskFuse", NUM_CLASSES=80,
                             flags=dict(HRNET_CLS_FEATURE=True)),
    "resnet50_coco2017": dict(CONV_BODY="resnet50.torch_resnet50", ROI_BOX_HEAD="resnet50.MaskFuse", NUM_CLASSES=80,
                              flags={}),
}


def apply_preset(name):
    p = PRESETS[name]
    reset_cfg()
    for k, v in _COMMON.items():
        cfg[k] = v
    for k, v in _FAST_RCNN.items():
        cfg.FAST_RCNN[k] = v
    cfg.FAST_RCNN.ROI_BOX_HEAD = p["ROI_BOX_HEAD"]
    cfg.MODEL.CONV_BODY = p["CONV_BODY"]
    cfg.MODEL.NUM_CLASSES = p["NUM_CLASSES"]
    cfg.MODEL.LOAD_IMAGENET_PRETRAINED_WEIGHTS = False
    # TEST section of the configs (configs/resnet50_voc.yaml:40-53): 5 scales x flip test-time augmentation
    cfg.TEST.SCALE, cfg.TEST.MAX_SIZE = 480, 2000
    cfg.TEST.BBOX_AUG.update(ENABLED=True, H_FLIP=True, SCALES=(576, 688, 864, 1200), SCALE_H_FLIP=True,
                             SCORE_HEUR="AVG", COORD_HEUR="ID")
    for k, v in p["flags"].items():
        cfg[k] = v
    return cfg
