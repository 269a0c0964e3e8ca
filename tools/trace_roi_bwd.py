#!/usr/bin/env python3
"""Per-workgroup timeline of the pipelined ROIAlign backward (dev tool): needs a library built with
-DCIM_ROI_PL_TRACE=1 as cim_amd/libcim_hip_alt_trace.so.  Prints, per workgroup: XCC / CU, entries, set-up, stream,
flush durations (us) and the kernel-wide picture (span, busy time per CU)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib, synthetic  # noqa: E402

dev = torch.device("cuda:0")
cfgname = sys.argv[1] if len(sys.argv) > 1 else "resnet50_voc"
inp = synthetic.make_image_inputs(cfgname, seed=3, with_image=False)
C, stride = 1024, 16
H, W = -(-inp["image_hw"][0] // stride), -(-inp["image_hw"][1] // stride)
K = inp["rois"].shape[0]
rois = torch.from_numpy(inp["rois"]).to(dev)
masks = torch.from_numpy(inp["masks"]).to(dev)
gcat = torch.randn(K, 7, 7, 2 * C, device=dev)
gin = torch.empty(1, H, W, C, device=dev)
st = _lib.stream_ptr()
ws = torch.empty(_lib.call("cim_roi_align_bwd_workspace", K, 7, H, W) // 4 + 1, device=dev)
alt = ctypes.CDLL(os.path.join(_lib.HERE, "libcim_hip_alt_trace.so"))
alt.cim_roi_align_maskcat_bwd_ws.argtypes = _lib.SIGNATURES["cim_roi_align_maskcat_bwd_ws"]
alt.cim_roi_align_bwd_scratch.restype = ctypes.c_longlong
nscr = alt.cim_roi_align_bwd_scratch(K, 1, C, H, W) // 4
NWG_MAX = 8192
scratch = torch.zeros(nscr + NWG_MAX * 32 + 16, device=dev)
run = lambda: alt.cim_roi_align_maskcat_bwd_ws(gcat.data_ptr(), rois.data_ptr(), masks.data_ptr(), gin.data_ptr(), 1, C, H, W, K, 7, 1.0 / stride, 0, 1, ws.data_ptr(), 0, scratch.data_ptr(), st)
for _ in range(3):
    run()
torch.cuda.synchronize()
scratch[nscr:].zero_()
run()
torch.cuda.synchronize()
tr = scratch[nscr:nscr + NWG_MAX * 32].cpu().numpy().view(np.uint64).reshape(-1, 16)
tr = tr[tr[:, 0] > 0]
t0 = tr[:, 0].min()
us = lambda a: (a.astype(np.float64) - float(t0)) / 100.0          # 100 MHz counter
hw = (tr[:, 7] >> np.uint64(32)).astype(np.int64)
xcc = (tr[:, 7] & np.uint64(0xf)).astype(np.int64)
cu = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7)
start, setup, stream, end = us(tr[:, 0]), us(tr[:, 1]), us(tr[:, 2]), us(tr[:, 3])
pstream = us(tr[:, 8 + 2])
ent = tr[:, 6].astype(np.int64)
print("workgroups", len(tr), "span us", end.max(), "first start", start.min(), "last start", start.max())
print("set-up us: mean %.2f max %.2f | stream us: mean %.2f max %.2f | flush us: mean %.2f max %.2f" % (
    (setup - start).mean(), (setup - start).max(), (stream - setup).mean(), (stream - setup).max(), (end - stream).mean(), (end - stream).max()))
print("set-up phases us (A inspect, B tables + entry map, C touch masks): %.2f %.2f %.2f" % ((us(tr[:, 4]) - start).mean(), (us(tr[:, 5]) - us(tr[:, 4])).mean(), (setup - us(tr[:, 5])).mean()))
print("wave 15: loads issued %.2f landed %.2f stored %.2f | wave 0: ranges landed %.2f at barrier %.2f (us after start)" % tuple(
    float((us(tr[:, c]) - (start if c >= 14 else us(tr[:, 8]))).mean()) for c in (11, 12, 13, 14, 15)))
nz = ent > 0
print("us per entry (stream / entries), by entries quartile:", [round(float(((stream - setup)[nz] / ent[nz])[np.argsort(ent[nz])][i::4].mean()), 4) for i in range(4)])
print("producer done vs consumer done (us): mean %.2f" % (stream - pstream)[nz].mean())
order = np.argsort(-ent)
for i in order[:12]:
    print("wg %4d xcc %d cu %3d entries %5d start %7.2f setup %6.2f stream %7.2f flush %6.2f end %7.2f" % (i, xcc[i], cu[i], ent[i], start[i], setup[i] - start[i], stream[i] - setup[i], end[i] - stream[i], end[i]))
key = xcc * 1000 + cu
busy = {}
for k in np.unique(key):
    m = key == k
    busy[k] = (float((end[m] - start[m]).sum()), int(m.sum()), float(end[m].max()), int(ent[m].sum()))
b = np.array([v[0] for v in busy.values()])
print("CUs seen", len(busy), "busy us per CU: mean %.1f min %.1f max %.1f; last end per CU: min %.1f max %.1f; entries per CU: min %d max %d" % (
    b.mean(), b.min(), b.max(), min(v[2] for v in busy.values()), max(v[2] for v in busy.values()), min(v[3] for v in busy.values()), max(v[3] for v in busy.values())))
