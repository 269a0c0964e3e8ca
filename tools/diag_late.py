#!/usr/bin/env python3
"""The last phase of the backward pass by stream (HIP events, no profiler): from the point MaskFuse's late weight-gradient products are
launched (behind the ROIAlign backward) to the end of each stream's work - the step's own stream (the backbone's data-gradient chains),
the body stream (the backbone's weight gradients), the late stream (MaskFuse's weight gradients).

    python3 tools/diag_late.py [--dw-form 0|1] [--dw-wgs N]"""
import gc
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cim_amd import _lib, mask_iou, synthetic  # noqa: E402
from cim_amd.core.config import cfg  # noqa: E402
from cim_amd.core.presets import apply_preset  # noqa: E402
from cim_amd.modeling import heads  # noqa: E402
from cim_amd.modeling.model_builder import Generalized_RCNN  # noqa: E402
from cim_amd.ops import gemm as G, maskfuse_pair as MP  # noqa: E402


def arg(name, default=None):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


if arg("--dw-form") is not None:
    MP.DW_FORM = arg("--dw-form")
if arg("--dw-wgs") is not None:
    if MP.DW_FORM == 1:
        MP.DW_FORM1_WGS = arg("--dw-wgs")
    else:
        MP.DW_WGS = arg("--dw-wgs")
dev = torch.device("cuda:0")
_lib.load()
apply_preset("resnet50_voc")
torch.manual_seed(cfg.RNG_SEED)
model = Generalized_RCNN()
bench.init_for_synthetic(model)
model = model.to(dev).train()
opt = bench.make_optimizer(model, torch)
opt.overlap_update = True
heads.LAZY_SETTLE = True
inp = synthetic.make_image_inputs("resnet50_voc", seed=3, n=1000)
iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(dev))
t = lambda a: torch.from_numpy(a).unsqueeze(0).to(dev)
batch = dict(data=torch.from_numpy(inp["data"]).to(dev), rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]),
             mat=t(inp["mat"]), index=t(inp["index"]), iou_map=iou, asy_iou_map=asy, gtrois=None)
np.random.seed(3)
marks = {}


def rec(stream=None):
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream if stream is not None else torch.cuda.current_stream())
    return e


run_postponed, join_side = G.run_postponed, G.join_side


def run_postponed_marked(d=None):
    if any(G._POSTPONED.get(x) for x in G._POSTPONED) and "late0" not in marks:
        marks["late0"] = rec()
    run_postponed(d)


def join_side_marked(discard=False):
    run_postponed_marked()
    if "late0" in marks and "main_end" not in marks:
        marks["main_end"] = rec()
        marks["side_end"] = rec(G._side_stream(dev))
        marks["body_end"] = rec(G._body_stream(dev))
        marks["late_end"] = rec(G._late_stream(dev))
    join_side(discard)


G.run_postponed, G.join_side = run_postponed_marked, join_side_marked
from cim_amd.ops import roi_align as _ra  # noqa: E402  (calls G.run_postponed through the module attribute)
rows = []
for it in range(40):
    if it == 6:
        gc.collect(); gc.freeze()
    marks.clear()
    opt.zero_grad(set_to_none=True)
    marks["step0"] = rec()
    out = model(**batch)
    marks["bwd0"] = rec()
    out["total_loss"].backward()
    marks["bwd_end"] = rec()
    opt.step()
    rows.append(dict(marks))
heads.settle_rng()
torch.cuda.synchronize()
names = [("bwd0", "backward starts"), ("late0", "late products launched (behind the ROIAlign backward)"), ("main_end", "step's stream: data-gradient chains done"),
         ("body_end", "body stream: backbone weight gradients done"), ("side_end", "side stream done"), ("late_end", "late stream: MaskFuse weight gradients done"),
         ("bwd_end", "backward done (joined)")]
r = np.array([[m["step0"].elapsed_time(m[k]) for k, _ in names] for m in rows[10:]])
for (k, n), v in zip(names, r.mean(0)):
    print("%-62s %7.3f ms" % (n, v))
print("last phase: %.3f ms; DW_FORM %d, DW_WGS %d / %d" % ((r[:, 6] - r[:, 1]).mean(), MP.DW_FORM, MP.DW_WGS, MP.DW_FORM1_WGS))
