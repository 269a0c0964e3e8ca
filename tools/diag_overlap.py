#!/usr/bin/env python3
"""Where the side-stream chain of a step (big weights' update -> their pair images -> containment-map prep -> transposed 3 x 3
weights) ends relative to the step's own stream (HIP events, no profiler): ms from the optimizer launch that opens a step."""
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
from cim_amd.ops import gemm  # noqa: E402

dev = torch.device("cuda:0")
_lib.load()
apply_preset("resnet50_voc")
torch.manual_seed(cfg.RNG_SEED)
model = Generalized_RCNN()
bench.init_for_synthetic(model)
model = model.to(dev).train()
opt = bench.make_optimizer(model, torch)
opt.overlap_update = "--no-overlap-update" not in sys.argv
if "--trail-wgs" in sys.argv:
    opt.trail_workgroups = int(sys.argv[sys.argv.index("--trail-wgs") + 1])
heads.LAZY_SETTLE = True
inp = synthetic.make_image_inputs("resnet50_voc", seed=3, n=1000)
iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(dev))
t = lambda a: torch.from_numpy(a).unsqueeze(0).to(dev)
batch = dict(data=torch.from_numpy(inp["data"]).to(dev), rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]),
             mat=t(inp["mat"]), index=t(inp["index"]), iou_map=iou, asy_iou_map=asy, gtrois=None)
np.random.seed(3)
E = lambda: torch.cuda.Event(enable_timing=True)
marks = {}
h1 = model.Conv_Body.register_forward_pre_hook(lambda m, i: marks.__setitem__("body0", _rec()))
h2 = model.Conv_Body.register_forward_hook(lambda m, i, o: marks.__setitem__("body1", _rec()))
h3 = model.Box_Head.register_forward_hook(lambda m, i, o: marks.__setitem__("maskfuse1", _rec()))


def _rec(stream=None):
    e = E()
    e.record(stream if stream is not None else torch.cuda.current_stream())
    return e


rows = []
side = gemm._side_stream(dev)
import gc
for it in range(48):
    if it == 6:
        gc.collect(); gc.freeze()
    opt.zero_grad(set_to_none=True)
    out = model(**batch)
    marks["side_fwd_end"] = _rec(side)              # everything the forward put on the side stream (images, prep, transposes)
    out["total_loss"].backward()
    marks["bwd_end"] = _rec()
    marks["opt0"] = _rec()
    opt.step()
    marks["opt_main_end"] = _rec()
    marks["opt_side_end"] = _rec(side)              # the big weights' update (overlap_update)
    if it >= 1:
        rows.append((keep, dict(marks)))
    keep = dict(marks)
heads.settle_rng()
torch.cuda.synchronize()                           # (ONE wait at the end: the host runs ahead of the device as in a training loop)
rows = [[p["opt0"].elapsed_time(p["opt_main_end"]), p["opt0"].elapsed_time(p["opt_side_end"]), p["opt0"].elapsed_time(m["body0"]),
         p["opt0"].elapsed_time(m["side_fwd_end"]), p["opt0"].elapsed_time(m["body1"]), p["opt0"].elapsed_time(m["maskfuse1"]),
         p["opt0"].elapsed_time(m["opt0"])] for p, m in rows]
r = np.array(rows[8:])
names = ["update on the step's stream done", "big weights' update done (side stream)", "backbone forward starts", "side stream's forward work done",
         "backbone forward done", "box head forward done", "next optimizer launch (= step)"]
for n, v in zip(names, r.mean(0)):
    print("%-45s %7.3f ms" % (n, v))
