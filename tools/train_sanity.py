#!/usr/bin/env python3
"""Loss trajectory of a short training run on one synthetic image (bench.py's model, optimizer and inputs), printed every
10 steps: run it twice, e.g. with and without --no-defer (ops.gemm.DEFER_DW = False) or --no-overlap (ops.gemm.OVERLAP = False), to
see that a scheduling option does not change what is learned.  Runs are not bit-reproducible (float atomics in the wide
BatchNorm-backward reductions)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cim_amd import mask_iou, synthetic  # noqa: E402
from cim_amd.core.config import cfg  # noqa: E402
from cim_amd.core.presets import apply_preset  # noqa: E402
from cim_amd.modeling.model_builder import Generalized_RCNN  # noqa: E402

from cim_amd.ops import gemm as _gemm  # noqa: E402
if "--no-defer" in sys.argv:
    _gemm.DEFER_DW = False
if "--no-overlap" in sys.argv:
    _gemm.OVERLAP = _gemm.DEFER_DW = False
_args = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(_args[0]) if _args else 60
dev = torch.device("cuda:0")
apply_preset("resnet50_voc")
torch.manual_seed(cfg.RNG_SEED)
model = Generalized_RCNN()
bench.init_for_synthetic(model)
model = model.to(dev).train()
opt = bench.make_optimizer(model, torch)
inp = synthetic.make_image_inputs("resnet50_voc", seed=3, n=600)
iou, asy = mask_iou.mask_iou_maps(torch.from_numpy(inp["full_masks"]).to(dev))
t = lambda a: torch.from_numpy(a).unsqueeze(0).to(dev)
batch = dict(data=torch.from_numpy(inp["data"]).to(dev), rois=t(inp["rois"]), masks=t(inp["masks"]), labels=t(inp["labels"]),
             mat=t(inp["mat"]), index=t(inp["index"]), iou_map=iou, asy_iou_map=asy, gtrois=None)
np.random.seed(cfg.RNG_SEED)
hist = []
for s in range(steps):
    opt.zero_grad(set_to_none=True)
    out = model(**batch)
    loss = sum(v.sum() for v in out["losses"].values())
    loss.backward()
    opt.step()
    hist.append([float(v) for v in out["losses"].values()])
    if (s + 1) % 10 == 0:
        print("step %3d  total %.5f  %s" % (s + 1, sum(hist[-1]), " ".join("%s %.5f" % (k, v) for k, v in zip(out["losses"], hist[-1]))), flush=True)
gn = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in model.parameters() if p.grad is not None)))
print("final grad norm %.5f" % gn)
