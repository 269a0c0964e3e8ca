#!/usr/bin/env python3
"""The tile-walking small GEMM / convolution kernels against the one-workgroup-per-item kernels (an alternative build with
-DCIM_SMALL_WALK=0): outputs must be BIT-IDENTICAL (same items, same slab order, same epilogue), and us per call of both.

    tools/build_alt.sh nowalk conv1x1.hip -DCIM_SMALL_WALK=0
    python3 tools/check_walk.py [cim_amd/libcim_hip_alt_nowalk.so]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(_lib.__file__), "libcim_hip_alt_nowalk.so")
alt = ctypes.CDLL(path)
NAMES = ("cim_gemm_small_f32", "cim_gemm_small_splits", "cim_conv1x1_bn_act_bwd", "cim_conv1x1_bwd_workspace", "cim_conv3x3_nchw_f32",
         "cim_conv3x3_nchw_splits", "cim_conv3x3_nchw_bn_act_bwd", "cim_conv3x3_nchw_bwd_workspace", "cim_conv7x7_nchw_f32")
for name in NAMES:
    getattr(alt, name).argtypes = _lib.SIGNATURES[name]
    getattr(alt, name).restype = ctypes.c_int
alt.cim_conv1x1_bwd_workspace.restype = ctypes.c_longlong
alt.cim_conv3x3_nchw_bwd_workspace.restype = ctypes.c_longlong
new = lambda name, *a: _lib.call(name, *a)
old = lambda name, *a: getattr(alt, name)(*a)
st = _lib.stream_ptr()
P = lambda t: None if t is None else t.data_ptr()
LAYERS = [("res2.conv1", 256, 64, 22188), ("res2.conv3", 64, 256, 22188), ("res3.conv1", 512, 128, 5590),
          ("res3.conv3", 128, 512, 5590), ("res4.conv1", 1024, 256, 1419), ("res4.conv3", 256, 1024, 1419),
          ("res4.0.conv1", 512, 256, 5590)]
CONVS = [("res2", 64, 64, 129, 172, 1), ("res3.0", 128, 128, 129, 172, 2), ("res3.1", 128, 128, 65, 86, 1), ("res4.0", 256, 256, 65, 86, 2),
         ("res4.1", 256, 256, 33, 43, 1)]


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def same(x, y):
    return bool(torch.equal(x.view(torch.int32), y.view(torch.int32)))


bad = 0
tot = {}
torch.manual_seed(1)
for name, cin, cout, hw in LAYERS:
    w = torch.randn(cout, cin, device=dev)
    x = torch.randn(cin, hw, device=dev)
    res = torch.randn(cout, hw, device=dev)
    bn = [torch.rand(cout, device=dev) + 0.5 for _ in range(4)]
    dy = torch.randn(cout, hw, device=dev)
    sf, sx, sw = (new("cim_gemm_small_splits", cout, hw, cin), new("cim_gemm_small_splits", cin, hw, cout), new("cim_gemm_small_splits", cout, cin, hw))
    ws = torch.empty(64 * max(cout * hw, cin * hw, cout * cin) // 8 + 1, device=dev)
    outs = {}
    rec = dict(layer=name, splits=(sf, sx, sw))
    for tag, call in (("new", new), ("old", old)):
        y = torch.full((cout, hw), float("nan"), device=dev)
        xr = torch.full((cout, hw), float("nan"), device=dev)
        dx = torch.full((cin, hw), float("nan"), device=dev)
        dw = torch.full((cout, cin), float("nan"), device=dev)
        fwd = lambda: call("cim_gemm_small_f32", P(w), P(x), P(y), cout, hw, cin, cin, hw, hw, 0, 0, P(xr), P(bn[0]), P(bn[1]), P(bn[2]), P(bn[3]),
                           1e-5, P(res), 1, sf, P(ws), st)
        fdx = lambda: call("cim_gemm_small_f32", P(w), P(dy), P(dx), cin, hw, cout, cin, hw, hw, 1, 0, None, None, None, None, None, 0.0, None, 0, sx, P(ws), st)
        fdw = lambda: call("cim_gemm_small_f32", P(dy), P(x), P(dw), cout, cin, hw, hw, hw, cin, 0, 1, None, None, None, None, None, 0.0, None, 0, sw, P(ws), st)
        for k, f in (("fwd", fwd), ("dx", fdx), ("dw", fdw)):
            rec["%s_%s_us" % (k, tag)] = round(timeit(f), 1)
            tot["%s_%s" % (k, tag)] = tot.get("%s_%s" % (k, tag), 0.0) + rec["%s_%s_us" % (k, tag)]
        outs[tag] = (y, xr, dx, dw)
    rec["identical"] = [same(a, b) for a, b in zip(outs["new"], outs["old"])]
    bad += sum(1 for v in rec["identical"] if not v)
    print(json.dumps(rec), flush=True)
    # the chained backward call (BatchNorm backward of the producer in the data gradient's epilogue + its affine partial sums)
    B = 1
    outs = {}
    for tag, call in (("new", new), ("old", old)):
        nb = call("cim_conv1x1_bwd_workspace", B, cin, cout, hw)
        wsb = torch.empty(nb // 4, device=dev)
        dx = torch.full((cin, hw), float("nan"), device=dev)
        dw = torch.full((cout, cin), float("nan"), device=dev)
        part = torch.full((1, 2, (hw + 31) // 32, cin), float("nan"), device=dev)
        ig, iv, im = (torch.rand(cin, device=dev) + 0.5 for _ in range(3))
        torch.manual_seed(7)
        xin = torch.randn(cin, hw, device=dev)
        xrin = torch.randn(cin, hw, device=dev)
        call("cim_conv1x1_bn_act_bwd", P(dy), None, P(dy), P(xin), P(w), P(bn[0]), P(bn[2]), P(bn[3]), 1e-5, 0, None, None, None, P(dx), P(dw),
             B, cin, cout, hw, P(wsb), st, None, None, None, 1, 1, P(ig), P(iv), 1e-5, P(xrin), P(im), P(part), None, 0)
        torch.cuda.synchronize()
        outs[tag] = (dx, dw, part)
    ok = [same(a, b) for a, b in zip(outs["new"], outs["old"])]
    bad += sum(1 for v in ok if not v)
    print(json.dumps(dict(layer=name, chained_backward_identical=ok)), flush=True)
print(json.dumps({k: round(v, 1) for k, v in tot.items()}))

tot = {}
for name, cin, cout, H, W, stride in CONVS:
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    x = torch.randn(cin, H, W, device=dev)
    bn = [torch.rand(cout, device=dev) + 0.5 for _ in range(4)]
    dy = torch.randn(cout, Ho, Wo, device=dev)
    sp = new("cim_conv3x3_nchw_splits", cin, cout, H, W, stride)
    ws = torch.empty(max(sp, 1) * cout * Ho * Wo, device=dev)
    outs = {}
    rec = dict(layer=name, splits=sp)
    for tag, call in (("new", new), ("old", old)):
        y = torch.full((cout, Ho, Wo), float("nan"), device=dev)
        xr = torch.full((cout, Ho, Wo), float("nan"), device=dev)
        fwd = lambda: call("cim_conv3x3_nchw_f32", P(x), P(w), P(y), cin, cout, H, W, stride, 1, P(xr), P(bn[0]), P(bn[1]), P(bn[2]), P(bn[3]), 1e-5,
                           None, 1, sp, P(ws), st)
        nb = call("cim_conv3x3_nchw_bwd_workspace", 1, cin, cout, H, W, stride)
        wsb = torch.empty(nb // 4, device=dev)
        dx = torch.full((cin, H, W), float("nan"), device=dev)
        dw = torch.full((cout, cin, 3, 3), float("nan"), device=dev)
        dg, db = torch.zeros(cout, device=dev), torch.zeros(cout, device=dev)
        fwd()
        bwd = lambda: call("cim_conv3x3_nchw_bn_act_bwd", P(dy), P(y), P(xr), P(x), P(w), P(bn[0]), P(bn[2]), P(bn[3]), 1e-5, 1, None, P(dg), P(db),
                           P(dx), P(dw), 1, cin, cout, H, W, stride, 1, P(wsb), st, None, None, None, 1, 0, None, None, 0.0, None, None, None, None)
        for k, f in (("fwd", fwd), ("bwd", bwd)):
            rec["%s_%s_us" % (k, tag)] = round(timeit(f), 1)
            tot["%s_%s" % (k, tag)] = tot.get("%s_%s" % (k, tag), 0.0) + rec["%s_%s_us" % (k, tag)]
        outs[tag] = (y, xr, dx, dw)
    rec["identical"] = [same(a, b) for a, b in zip(outs["new"], outs["old"])]
    bad += sum(1 for v in rec["identical"] if not v)
    print(json.dumps(rec), flush=True)
print(json.dumps({k: round(v, 1) for k, v in tot.items()}))
# the stem
x = torch.randn(3, 516, 688, device=dev)
w = torch.randn(64, 3, 7, 7, device=dev) * 0.05
bn = [torch.rand(64, device=dev) + 0.5 for _ in range(4)]
outs = {}
rec = {}
for tag, call in (("new", new), ("old", old)):
    y = torch.full((64, 258, 344), float("nan"), device=dev)
    f = lambda: call("cim_conv7x7_nchw_f32", P(x), P(w), P(y), 3, 64, 516, 688, 2, P(bn[0]), P(bn[1]), P(bn[2]), P(bn[3]), 1e-5, 1, st)
    rec["stem_%s_us" % tag] = round(timeit(f), 1)
    outs[tag] = y
rec["identical"] = same(outs["new"], outs["old"])
bad += 0 if rec["identical"] else 1
print(json.dumps(rec))
print("MISMATCHES", bad)
sys.exit(1 if bad else 0)
