"""`Generalized_RCNN`: the CIM per-image training / inference forward on MI355X.

Mirrors /root/reference/lib/modeling/model_builder.py (`Generalized_RCNN` :71-263, `get_func`
:16-34, `testing_function` :60-68): constructor without arguments reading the global `cfg`;
`forward(data, rois, masks, labels, gtrois, mat, path=None, index=None)` returning
`{'blob_conv', 'losses': {bag_loss, pcl_loss, cls_loss, iou_loss}}` (each loss shape [1]) in
training and `{'blob_conv', 'refine_score'}` in eval; attributes `Conv_Body`, `Box_Head`,
`cls_iou_model`, `CIM_layer_list`, `using_CIM`; `roi_feature_transform`, `convbody_net`,
`detectron_weight_mapping`.

The training forward never waits for the device: there is no host copy of `labels` / `mat`, no per-image cache,
the mining's host-visible results (stream position of the NumPy generator, format errors) are settled at the end
of the backward pass (heads._RngLedger).

Extension: `forward` also accepts `iou_map=` / `asy_iou_map=` float16 device
tensors [N,N] (e.g. built on device by cim_amd.mask_iou) so that no disk I/O happens inside
the step; when absent the reference's pickle path (model_builder.py:147-159) is used.
"""
import importlib
import logging
import os
import pickle
from functools import wraps

import torch
import torch.nn as nn

from ..core.config import cfg
from ..ops import RoIAlign, RoIPool
from ..ops import gemm as _gemm_ops
from . import heads

logger = logging.getLogger(__name__)


def get_func(func_name):
    """Resolve 'resnet50.MaskFuse'-style names relative to this `modeling` package
    (reference model_builder.py:16-34)."""
    if func_name == "":
        return None
    try:
        parts = func_name.split(".")
        if len(parts) == 1:
            return globals()[parts[0]]
        module = importlib.import_module(__package__ + "." + ".".join(parts[:-1]))
        return getattr(module, parts[-1])
    except Exception:
        logger.error("Failed to find function: %s", func_name)
        raise


def check_inference(net_func):
    @wraps(net_func)
    def wrapper(self, *args, **kwargs):
        if self.training:
            raise ValueError("You should call this function only on inference."
                             "Set the network in inference mode by net.eval().")
        with torch.no_grad():
            return net_func(self, *args, **kwargs)
    return wrapper


def testing_function(predict_cls, predict_det, ref_cls_score, ref_iou_score, return_dict):
    return_dict["refine_score"] = [(c * i)[:, 1:] for c, i in zip(ref_cls_score, ref_iou_score)]
    return return_dict


def _load_map(directory, stem, what, device, index):
    """The reference's per-step pickle path (model_builder.py:147-159)."""
    path = os.path.join(directory, stem + ".pkl")
    try:
        with open(path, "rb") as f:
            m = pickle.load(f)
        return torch.tensor(m, device=device)[index][:, index]
    except Exception:
        print(what + " lose " + path)
        raise NotImplementedError("Please generate or download " + what)


PCL_GENERAL = os.environ.get("CIM_PCL_GENERAL", "0") == "1"            # losses in ATen ops (general `mat` format)
GRAPH_BACKBONE = os.environ.get("CIM_GRAPH_BACKBONE", "0") == "1"     # opt-in: measured SLOWER than eager (see _conv_body)
GRAPH_AFTER = 3         # graph a shape from its 3rd occurrence
GRAPH_SHAPES = 4        # distinct image shapes kept as graphs


class _BodyWrapper(nn.Module):
    """make_graphed_callables patches the forward of the module it is given; one wrapper per image shape shares the
    body's parameters without touching the body (or the model's state_dict: wrappers live outside the module tree)."""

    def __init__(self, body):
        super().__init__()
        self.body = body

    def forward(self, x):
        return self.body(x)


class Generalized_RCNN(nn.Module):
    def __init__(self):
        super().__init__()
        self.mapping_to_detectron = None
        self.orphans_in_detectron = None
        cls_num = cfg.MODEL.NUM_CLASSES + 1
        self.Conv_Body = get_func(cfg.MODEL.CONV_BODY)()
        self.Box_Head = get_func(cfg.FAST_RCNN.ROI_BOX_HEAD)(self.Conv_Body.dim_out, self.roi_feature_transform,
                                                            self.Conv_Body.spatial_scale)
        self.cls_iou_model = heads.cls_iou_model(self.Box_Head.dim_out, cls_num, cfg.REFINE_TIMES, class_agnostic=False)
        # a plain list like the reference (no parameters, not in state_dict) - model_builder.py:88-94
        self.CIM_layer_list = [heads.CIM_layer(p_seed=cfg.p_seed,
                                               cls_thr=0.25 + cfg.step_rate * t,
                                               iou_thr=0.5 + cfg.step_rate * t,
                                               Anti_noise_sampling=cfg.Anti_noise_sampling)
                               for t in range(cfg.REFINE_TIMES)]
        self.using_CIM = [True, True, True]
        self._init_modules()

    def _init_modules(self):
        """model_builder.py:101-115: ImageNet weights through the reference's own helper modules (resolved by their
        reference names, i.e. with lib/ on sys.path as under tools/train.py); the ResNet body loads torchvision's
        weights itself (resnet50.py:20)."""
        if cfg.MODEL.LOAD_IMAGENET_PRETRAINED_WEIGHTS:
            for flag, helper in (("VGG_CLS_FEATURE", "utils.vgg_weights_helper"),
                                 ("HRNET_CLS_FEATURE", "utils.hrnet_weights_helper")):
                if cfg[flag]:
                    try:
                        mod = importlib.import_module(helper)
                    except ImportError as e:
                        raise ImportError("cfg.MODEL.LOAD_IMAGENET_PRETRAINED_WEIGHTS needs the reference's %s on "
                                          "sys.path (run under tools/train.py) - %s" % (helper, e))
                    mod.load_pretrained_imagenet_weights(self)
        if cfg.TRAIN.FREEZE_CONV_BODY:
            for p in self.Conv_Body.parameters():
                p.requires_grad = False

    def state_dict(self, *args, **kwargs):
        _gemm_ops.wait_pending_updates()        # (cim_amd.optim.SGD.overlap_update: big weights may still be updated on the side stream)
        return super().state_dict(*args, **kwargs)

    def forward(self, data, rois, masks, labels, gtrois=None, mat=None, path=None, index=None,
                iou_map=None, asy_iou_map=None):
        """model_builder.py:117-213.  With `ops.gemm.HIGH_PRIO = True` (a module attribute; opt-in experiment, no measured gain: ops/maskfuse_pair.py) the TRAINING step
        runs on a high-priority HIP stream of its own (ops/gemm.py: main_stream_high_priority; its backward follows it there); the
        caller's stream is ordered before and after, so drivers see ordinary tensors."""
        if not (self.training and _gemm_ops.HIGH_PRIO and torch.is_tensor(data) and data.is_cuda) or torch.cuda.is_current_stream_capturing():
            return self._forward_impl(data, rois, masks, labels, gtrois, mat, path, index, iou_map, asy_iou_map)
        dev = data.device
        cur, hp = torch.cuda.current_stream(dev), _gemm_ops.main_stream_high_priority(dev)
        if cur == hp:
            return self._forward_impl(data, rois, masks, labels, gtrois, mat, path, index, iou_map, asy_iou_map)
        hp.wait_stream(cur)
        for t in (data, rois, masks, labels, gtrois, mat, index, iou_map, asy_iou_map):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(hp)
        with torch.cuda.stream(hp):
            out = self._forward_impl(data, rois, masks, labels, gtrois, mat, path, index, iou_map, asy_iou_map)
        cur.wait_stream(hp)
        out["blob_conv"].record_stream(cur)
        for v in out.get("losses", {}).values():
            v.record_stream(cur)
        return out

    def _forward_impl(self, data, rois, masks, labels, gtrois=None, mat=None, path=None, index=None,
                      iou_map=None, asy_iou_map=None):
        with torch.set_grad_enabled(self.training):
            im_data = data
            if self.training:
                # (weight gradients a previous backward pass left on the side stream - only if that pass was ABORTED before
                # its end-of-backward callback: normally nothing is pending here.  They are dropped, not installed: the
                # driver's zero_grad() for this step has already run)
                _gemm_ops.join_side(discard=True)
                if hasattr(self.Box_Head, "prefetch"):
                    self.Box_Head.prefetch()        # the MaskFuse filter transform runs under the backbone forward
                dev, dt = im_data.device, im_data.dtype
                rois = rois.squeeze(dim=0).to(device=dev, dtype=dt)
                masks = masks.squeeze(dim=0).to(device=dev, dtype=dt)
                labels = labels.squeeze(dim=0).to(device=dev, dtype=dt)
                mat = mat.squeeze(dim=0).to(device=dev, dtype=dt)
                if index is not None:
                    index = index.squeeze(dim=0).to(device=dev).long()
            return_dict = {}
            prep = None
            if self.training and asy_iou_map is not None and asy_iou_map.is_cuda:
                # what the mining needs from the containment map alone (flags, transposed copy) runs on the side stream under the
                # backbone forward (heads.prepare_containment); maps that come from the reference's pickles are loaded after the
                # heads as the reference does (model_builder.py:147-159) and prepared there
                prep = heads.prepare_containment(self.CIM_layer_list, asy_iou_map, self.using_CIM)
            blob_conv = self._conv_body(im_data)
            return_dict["blob_conv"] = blob_conv
            seg_x = self.Box_Head(blob_conv, rois, masks.detach())
            predict_cls, predict_det, ref_cls_score, ref_iou_score = self.cls_iou_model(seg_x)

            if not self.training:
                return testing_function(predict_cls, predict_det, ref_cls_score, ref_iou_score, return_dict)

            if iou_map is None or asy_iou_map is None:
                stem = os.path.splitext(os.path.split(path)[1])[0]
                iou_map = _load_map(cfg.iou_dir, stem, "iou_map", labels.device, index)
                asy_iou_map = _load_map(cfg.asy_iou_dir, stem, "asy_iou_map", labels.device, index)

            # model_builder.py:170-187: layer 0 mines on the MIL scores, layer i on the (i-1)-th refinement's; every
            # layer reads head outputs only, so all of them run in ONE fused set of launches - no host round trip:
            # classes, pseudo-GT counts, the anti-noise sampling and "no pseudo GT -> skip the layer" stay on the device
            scores = [(predict_cls, predict_det) if i == 0 else (ref_cls_score[i - 1], ref_iou_score[i - 1])
                      for i in range(len(self.CIM_layer_list))]
            # (measurement aid: `model.__dict__["_fixed_mining"]` = the mined labels of ANOTHER run of the same image (an object with
            # .pseudo / .valid / .status / .commit()) replaces this step's mining - bench.py compares two arithmetic classes on the
            # same pseudo labels, so that the deviation measures arithmetic and not flipped labels)
            mined = self.__dict__.get("_fixed_mining")
            if mined is None:
                mined = heads.mine_step(self.CIM_layer_list, scores, labels, iou_map, asy_iou_map, self.using_CIM, prep=prep)
            scales = [3 if i == 0 else 1 for i in range(len(self.CIM_layer_list))]      # lmda, model_builder.py:172
            if cfg.REFINE_TIMES <= 3 and not PCL_GENERAL:
                # all four losses + their gradient components in one HIP launch (csrc/losses.hip)
                bag, pcl, cls_l, _, iou3, total = heads.fused_losses(predict_cls, predict_det, ref_cls_score, ref_iou_score,
                                                                     labels, mined.pseudo, scales, mat, valid=mined.valid,
                                                                     status=mined.status, with_total=True)
                mined.commit()
                losses = dict(bag_loss=bag, pcl_loss=pcl, cls_loss=cls_l, iou_loss=iou3)       # model_builder.py:199: 3 * iou_loss
                # the sum the driver differentiates (lib/utils/training_stats.py:72-83 builds the same `total_loss` from the four
                # entries with a mean and an add each, tools/train.py:435 calls backward on it): an extra key, made by the loss
                # launch's finishing kernel - a loop that uses it saves ~25 scalar launches per step
                return_dict["total_loss"] = total.unsqueeze(0)
            else:   # general `mat` (several non-zeros per row) / more than 3 refinements: the reference's formulation
                    # in ATen ops; which layers count is decided on the host as the reference does (one device wait)
                mined.commit()
                heads.settle_rng()
                zero = seg_x.new_zeros(())
                losses = dict(bag_loss=zero.clone(), pcl_loss=zero.clone(), cls_loss=zero.clone(), iou_loss=zero.clone())
                for i, ps in enumerate(mined.pseudo):
                    if int(mined.host[2 + 2 * i]) == 0:                          # model_builder.py:189-190
                        continue
                    cls_loss, iou_loss, bag_loss = heads.cls_iou_loss(ref_cls_score[i], ref_iou_score[i], ps[0], ps[1],
                                                                      scales[i] * ps[2], labels)
                    losses["cls_loss"] = losses["cls_loss"] + cls_loss
                    losses["iou_loss"] = losses["iou_loss"] + 3 * iou_loss
                    losses["bag_loss"] = losses["bag_loss"] + bag_loss
                losses["bag_loss"] = losses["bag_loss"] + heads.mil_bag_loss(predict_cls, predict_det, labels)
                losses["pcl_loss"] = losses["pcl_loss"] + heads.PCL_loss(predict_cls, mat, labels)
            self.__dict__["_last_mining"] = mined              # debugging / tests: device-side intermediates of the step
            return_dict["losses"] = {k: v.unsqueeze(0) for k, v in losses.items()}
            return return_dict

    def _conv_body(self, im_data):
        """Backbone forward.  OPT-IN (CIM_GRAPH_BACKBONE=1): for an image shape that keeps coming back (training scales are a small
        set: SURVEY App. C) the body's forward AND backward are captured as HIP graphs (torch.cuda.make_graphed_callables) and
        replayed - from the shape's GRAPH_AFTER-th occurrence, at most GRAPH_SHAPES shapes (each capture pins its activations).
        The default stays eager: the host keeps ahead of the body's ~90 forward / ~170 backward launches (bench.py --phases: the
        body's forward takes its kernels' time, no launch gaps), a replayed graph runs on ONE stream (the deferred weight gradients
        lose their side stream) and hipGraph kernel nodes cost about what eager launches do: +4.6 ms per step when last measured
        (round 3).  Kept as a tested path (tests/test_gpu_parity.py) because it proves that every body kernel
        is capture-safe (no allocation, no host synchronisation inside)."""
        state = self.__dict__.setdefault("_graphed_bodies", {"seen": {}, "graphs": {}})
        if not (GRAPH_BACKBONE and self.training and im_data.is_cuda and torch.is_grad_enabled()
                and any(p.requires_grad for p in self.Conv_Body.parameters())):
            return self.Conv_Body(im_data)
        key = (tuple(im_data.shape), im_data.dtype, im_data.device)
        g = state["graphs"].get(key)
        if g is None:
            n = state["seen"][key] = state["seen"].get(key, 0) + 1
            if n < GRAPH_AFTER or len(state["graphs"]) >= GRAPH_SHAPES:
                return self.Conv_Body(im_data)
            wrapper = _BodyWrapper(self.Conv_Body)
            from ..ops import gemm as _gemm
            defer, _gemm.DEFER_DW = _gemm.DEFER_DW, False       # (make_graphed_callables differentiates with torch.autograd.grad and on one stream)
            try:
                g = torch.cuda.make_graphed_callables(wrapper, (im_data.detach().clone(),))
            except Exception as e:      # capture not possible (e.g. a library kernel that allocates): stay eager
                _gemm.DEFER_DW = defer
                print("cim_amd: backbone graph capture failed (%s: %s); running eagerly" % (type(e).__name__, e))
                state["graphs"][key] = False
                return self.Conv_Body(im_data)
            _gemm.DEFER_DW = defer
            state["graphs"][key] = g
        if g is False:
            return self.Conv_Body(im_data)
        return g(im_data)

    def roi_feature_transform(self, blobs_in, rois, method="RoIPoolF", resolution=7, spatial_scale=1.0 / 16.0,
                              sampling_ratio=0):
        """Same dispatch as model_builder.py:215-233 (kept for API compatibility; MaskFuse uses the
        fused ROIAlign+mask-cat kernel directly)."""
        assert method in {"RoIPoolF", "RoICrop", "RoIAlign"}, "Unknown pooling method: {}".format(method)
        if method == "RoIPoolF":
            return RoIPool(resolution, spatial_scale)(blobs_in, rois)
        if method == "RoIAlign":
            return RoIAlign(resolution, spatial_scale, sampling_ratio)(blobs_in.contiguous(), rois.contiguous())

    @check_inference
    def convbody_net(self, data):
        return self.Conv_Body(data)

    @property
    def detectron_weight_mapping(self):
        if self.mapping_to_detectron is None:
            d_wmap, d_orphan = {}, []
            for name, child in self.named_children():
                if list(child.parameters()):
                    child_map, child_orphan = child.detectron_weight_mapping()
                    d_orphan.extend(child_orphan)
                    for key, value in child_map.items():
                        d_wmap[name + "." + key] = value
            self.mapping_to_detectron = d_wmap
            self.orphans_in_detectron = d_orphan
        return self.mapping_to_detectron, self.orphans_in_detectron

    def _add_loss(self, return_dict, key, value):
        return_dict["losses"][key] = value
