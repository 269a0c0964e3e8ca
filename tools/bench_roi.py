#!/usr/bin/env python3
"""Micro-benchmark of the fused ROIAlign+mask-cat kernels at BASELINE cfg2 size (1000 synthetic
proposals, 1024 x 33 x 43 map): ms per launch and fraction of the 8 TB/s HBM roofline on the
algorithmic bytes of SURVEY.md 8(d)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib, synthetic  # noqa: E402

dev = torch.device("cuda:0")
cfgname = sys.argv[1] if len(sys.argv) > 1 else "resnet50_voc"
inp = synthetic.make_image_inputs(cfgname, seed=3, with_image=False)
C = {"resnet50_voc": 1024, "resnet50_coco2017": 1024, "vgg16_voc": 512}[cfgname]
stride = 8 if cfgname.startswith("vgg") else 16
H, W = -(-inp["image_hw"][0] // stride), -(-inp["image_hw"][1] // stride)
K = inp["rois"].shape[0]
rois = torch.from_numpy(inp["rois"]).to(dev)
masks = torch.from_numpy(inp["masks"]).to(dev)
feat = torch.randn(1, H, W, C, device=dev)
cat = torch.empty(K, 7, 7, 2 * C, device=dev)
gcat = torch.randn(K, 7, 7, 2 * C, device=dev)
gin = torch.empty(1, H, W, C, device=dev)
st = _lib.stream_ptr()


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ws = torch.empty(_lib.call("cim_roi_align_bwd_workspace", K, 7, H, W) // 4 + 1, device=dev)
fwd = lambda: _lib.call("cim_roi_align_maskcat_fwd_ws", feat.data_ptr(), rois.data_ptr(), masks.data_ptr(), cat.data_ptr(), 1, C, H, W, K, 7, 1.0 / stride, 0, 1, ws.data_ptr(), st)
bwd = lambda: _lib.call("cim_roi_align_maskcat_bwd", gcat.data_ptr(), rois.data_ptr(), masks.data_ptr(), gin.data_ptr(), 1, C, H, W, K, 7, 1.0 / stride, 0, 1, ws.data_ptr(), st)
nbytes = 4.0 * (C * H * W + 5 * K + 49 * K) + 4.0 * K * 2 * C * 49
tf, tb = timeit(fwd), timeit(bwd)
# what the training step runs since round 5: the forward that writes the Winograd input pair image, the plain backward on dbox
from cim_amd.ops import maskfuse_pair, pair  # noqa: E402
rp = pair.pad32(K)
sV = maskfuse_pair._input_scales(pair.amax_of(feat), dev)
V = torch.empty(121, rp, 2 * C, dtype=torch.int32, device=dev)
gbox = torch.randn(K, 7, 7, C, device=dev)
scratch = torch.empty(_lib.call("cim_roi_align_bwd_scratch", K, 1, C, H, W) // 4 + 1, device=dev)
fwdw = lambda: _lib.call("cim_roi_align_wino7_pair_fwd", feat.data_ptr(), rois.data_ptr(), masks.data_ptr(), V.data_ptr(), sV.data_ptr(), 1, C, H, W, K, rp, 7, 1.0 / stride, 0, 1, ws.data_ptr(), st)
bwdp = lambda: _lib.call("cim_roi_align_bwd_ws", gbox.data_ptr(), rois.data_ptr(), gin.data_ptr(), 1, C, H, W, K, 7, 1.0 / stride, 0, 1, ws.data_ptr(), 0, scratch.data_ptr(), st)
tfw, tbp = timeit(fwdw), timeit(bwdp)
wbytes = 4.0 * (C * H * W + 5 * K + 49 * K) + 4.0 * 121 * rp * 2 * C
pbytes = 4.0 * (C * H * W + 5 * K) + 4.0 * K * C * 49
import ctypes, glob
for path in sorted(glob.glob(os.path.join(_lib.HERE, "libcim_hip_alt*.so"))):     # ALT builds (ablations)
    alt = ctypes.CDLL(path)
    for name, argt in _lib.SIGNATURES.items():
        getattr(alt, name).argtypes = argt
    f = lambda: alt.cim_roi_align_maskcat_fwd_ws(feat.data_ptr(), rois.data_ptr(), masks.data_ptr(), cat.data_ptr(), 1, C, H, W, K, 7, 1.0 / stride, 0, 1, ws.data_ptr(), st)
    bw = lambda: alt.cim_roi_align_maskcat_bwd(gcat.data_ptr(), rois.data_ptr(), masks.data_ptr(), gin.data_ptr(), 1, C, H, W, K, 7, 1.0 / stride, 0, 1, ws.data_ptr(), st)
    fwd(); torch.cuda.synchronize(); ref_cat = cat.clone(); cat.zero_()
    f(); torch.cuda.synchronize()
    print(os.path.basename(path), "fwd_ms %.4f bwd_ms %.4f  max |fwd - base| %.3g" % (timeit(f), timeit(bw), float((cat - ref_cat).abs().max())))
print(json.dumps(dict(config=cfgname, K=K, C=C, H=H, W=W, alg_MB=nbytes / 1e6, fwd_ms=tf, fwd_frac=nbytes / tf / 1e6 / 8000,
                      bwd_ms=tb, bwd_frac=nbytes / tb / 1e6 / 8000,
                      wino7_pair_fwd_ms=tfw, wino7_pair_fwd_alg_MB=wbytes / 1e6, wino7_pair_fwd_frac=wbytes / tfw / 1e6 / 8000,
                      plain_bwd_ms=tbp, plain_bwd_alg_MB=pbytes / 1e6, plain_bwd_frac=pbytes / tbp / 1e6 / 8000)))
