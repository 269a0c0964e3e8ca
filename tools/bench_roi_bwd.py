#!/usr/bin/env python3
"""ROIAlign(+mask-cat) backward (region form), the ROI-group-size sweep and any ablation builds (cim_amd/libcim_hip_alt*.so):
ms per launch, fraction of the 8 TB/s HBM roofline on SURVEY.md 8(d)'s algorithmic bytes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib, synthetic  # noqa: E402

dev = torch.device("cuda:0")
out = []
for cfgname, n, target in (("resnet50_voc", None, None), ("resnet50_voc", 800, 576), ("resnet50_voc", 1200, 864),
                           ("resnet50_coco2017", None, None), ("vgg16_voc", None, None)):
    inp = synthetic.make_image_inputs(cfgname, seed=3, with_image=False, n=n, target=target)
    C = 512 if cfgname.startswith("vgg") else 1024
    stride = 8 if cfgname.startswith("vgg") else 16
    H, W = -(-inp["image_hw"][0] // stride), -(-inp["image_hw"][1] // stride)
    K = inp["rois"].shape[0]
    rois = torch.from_numpy(inp["rois"]).to(dev)
    masks = torch.from_numpy(inp["masks"]).to(dev)
    gcat = torch.randn(K, 7, 7, 2 * C, device=dev)
    gin = torch.empty(1, H, W, C, device=dev)
    st = _lib.stream_ptr()
    ws = torch.empty(_lib.call("cim_roi_align_bwd_workspace", K, 7, H, W) // 4 + 1, device=dev)

    def timeit(fn, n=20):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    scratch = torch.empty(_lib.call("cim_roi_align_bwd_scratch", K, 1, C, H, W) // 4 + 1, device=dev)
    bwd = lambda: _lib.call("cim_roi_align_maskcat_bwd_ws", gcat.data_ptr(), rois.data_ptr(), masks.data_ptr(), gin.data_ptr(), 1, C, H, W, K, 7, 1.0 / stride, 0, 1, ws.data_ptr(), 0, scratch.data_ptr(), st)
    nbytes = 4.0 * (C * H * W + 5 * K + 49 * K) + 4.0 * K * 2 * C * 49
    t_region = timeit(bwd)
    sweep = {}
    # (the ROI-group size is the launcher's own choice: the CIM_ROI_RG_GS sweep switch went with the library's getenv reads)
    # the PLAIN backward (cim_roi_align_bwd_ws on dbox [K,7,7,C]: what the step runs since round 5 - the mask multiply + concat
    # backward is folded into cim_wino7_dx_maskfold)
    gbox = torch.randn(K, 7, 7, C, device=dev)
    plain = lambda lib_call: lib_call("cim_roi_align_bwd_ws", gbox.data_ptr(), rois.data_ptr(), gin.data_ptr(), 1, C, H, W, K, 7, 1.0 / stride, 0, 1, ws.data_ptr(), 0, scratch.data_ptr(), st)
    t_plain = timeit(lambda: plain(_lib.call))
    plain_bytes = 4.0 * (C * H * W + 5 * K) + 4.0 * K * C * 49
    import ctypes, glob
    alts = {}
    for path in sorted(glob.glob(os.path.join(_lib.HERE, "libcim_hip_alt*.so"))):     # ablation builds
        alt = ctypes.CDLL(path)
        alt.cim_roi_align_maskcat_bwd_ws.argtypes = _lib.SIGNATURES["cim_roi_align_maskcat_bwd_ws"]
        alt.cim_roi_align_bwd_scratch.restype = ctypes.c_longlong
        sc = torch.empty(alt.cim_roi_align_bwd_scratch(K, 1, C, H, W) // 4 + 1, device=dev)
        f = lambda: alt.cim_roi_align_maskcat_bwd_ws(gcat.data_ptr(), rois.data_ptr(), masks.data_ptr(), gin.data_ptr(), 1, C, H, W, K, 7, 1.0 / stride, 0, 1, ws.data_ptr(), 0, sc.data_ptr(), st)
        alts[os.path.basename(path)] = timeit(f)
        alt.cim_roi_align_bwd_ws.argtypes = _lib.SIGNATURES["cim_roi_align_bwd_ws"]
        alts[os.path.basename(path) + " plain"] = timeit(lambda: alt.cim_roi_align_bwd_ws(gbox.data_ptr(), rois.data_ptr(), gin.data_ptr(), 1, C, H, W, K, 7, 1.0 / stride, 0, 1, ws.data_ptr(), 0, sc.data_ptr(), st))
    out.append(dict(alts=alts, config=cfgname, K=K, C=C, H=H, W=W, alg_MB=nbytes / 1e6, plain_ms=t_plain, plain_alg_MB=plain_bytes / 1e6,
                    plain_frac=plain_bytes / t_plain / 1e6 / 8000, region_ms=t_region,
                    region_frac=nbytes / t_region / 1e6 / 8000, group_size_sweep_ms=sweep))
    print(json.dumps(out[-1]))
