#!/usr/bin/env python3
"""On-device mask-IoU / containment map construction (a-7 / f-1) at cfg2 and cfg4 sizes: ms per stage and the
roofline figures of SURVEY.md 8(d): N*ceil(HW/8) bytes in + 4 N^2 out, N^2/2 * ceil(HW/64) 64-bit and+popcount ops."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib, mask_iou, synthetic  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for cfgname in ("resnet50_voc", "resnet50_coco2017"):
    inp = synthetic.make_image_inputs(cfgname, seed=3, with_image=False)
    masks = torch.from_numpy(inp["full_masks"]).to(dev)
    n, h, w = masks.shape
    hw = h * w
    words = (hw + 63) // 64
    t_pack = timeit(lambda: mask_iou.pack_masks(masks))
    packed = mask_iou.pack_masks(masks)
    t_pair = timeit(lambda: mask_iou.maps_from_packed(packed))
    t_all = timeit(lambda: mask_iou.mask_iou_maps(masks))
    ops = n * n / 2.0 * words                       # 64-bit and + popcount pairs (symmetric half)
    print(json.dumps(dict(config=cfgname, N=n, HW=hw, pack_ms=t_pack, pack_GBs=n * hw / t_pack / 1e6,
                          maps_ms=t_pair, and_popcount_Gops=ops / t_pair / 1e6, total_ms=t_all,
                          algorithmic_bytes=n * ((hw + 7) // 8) + 4.0 * n * n,
                          hbm_frac_of_8TBs=(n * hw + n * words * 8 + 4.0 * n * n) / t_all / 1e6 / 8000)))
