#!/usr/bin/env python3
"""Backbone 3 x 3 convolutions (+ BatchNorm + ReLU), forward and backward: the implicit-GEMM HIP op against ATen / MIOpen
on the C4 body's layer shapes at cfg2 (516 x 688 image).  us per call (whole backward = BN / ReLU backward + dX + dW).
`hip_*_us` / `aten_*_us` time the Python operators (autograd included: at these sizes the HIP operator's ~130 us of host work per
forward + backward is LONGER than its kernels - round 5's "slower than MIOpen" was this loop being host-bound);
`hip_direct_*_us` time the same kernels through the C ABI on preallocated buffers (GPU-bound: what the step pays, where the host runs
ahead of the device)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib  # noqa: E402
from cim_amd.ops import bn_act, conv3x3_bn_act  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


out = []
for name, c, H, W, stride, train in (("res2 (frozen)", 64, 129, 172, 1, False), ("res3.0", 128, 129, 172, 2, True), ("res3.1-3", 128, 65, 86, 1, True),
                                     ("res4.0", 256, 65, 86, 2, True), ("res4.1-5", 256, 33, 43, 1, True)):
    conv = torch.nn.Conv2d(c, c, 3, stride=stride, padding=1, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(c).to(dev).eval()
    x = torch.randn(1, c, H, W, device=dev, requires_grad=train)
    res = dict(layer=name, cin=c, H=H, W=W, stride=stride, gflop=2e-9 * 9 * c * c * ((H - 1) // stride + 1) * ((W - 1) // stride + 1))
    for label, fn in (("hip", lambda: conv3x3_bn_act(x, conv, bn)), ("aten", lambda: bn_act(conv(x), bn))):
        if not train:
            with torch.no_grad():
                res[label + "_fwd_us"] = timeit(fn)
            continue
        y = fn()
        g = torch.randn_like(y)
        res[label + "_fwd_us"] = timeit(fn)
        res[label + "_fwd_bwd_us"] = timeit(lambda: fn().backward(g))
    # the same launches through the C ABI (no Python operator, no autograd): cim_conv3x3_nchw_f32 / cim_conv3x3_nchw_bn_act_bwd
    P = lambda t: None if t is None else t.data_ptr()
    st = _lib.stream_ptr()
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    w, xs = conv.weight.detach(), x.detach()[0].contiguous()
    bnp = [bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var]
    sp = _lib.call("cim_conv3x3_nchw_splits", c, c, H, W, stride)
    ws = torch.empty(max(sp, 1) * c * Ho * Wo, device=dev)
    yd, xr = torch.empty(c, Ho, Wo, device=dev), torch.empty(c, Ho, Wo, device=dev)
    fwd = lambda: _lib.call("cim_conv3x3_nchw_f32", P(xs), P(w), P(yd), c, c, H, W, stride, 1, P(xr), P(bnp[0]), P(bnp[1]), P(bnp[2]), P(bnp[3]),
                            1e-5, None, 1, sp, P(ws), st)
    res["hip_direct_fwd_us"] = timeit(fwd)
    if train:
        wsb = torch.empty(_lib.call("cim_conv3x3_nchw_bwd_workspace", 1, c, c, H, W, stride) // 4, device=dev)
        dyd, dxd, dwd = torch.randn(c, Ho, Wo, device=dev), torch.empty(c, H, W, device=dev), torch.empty(c, c, 3, 3, device=dev)
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        bwd = lambda: _lib.call("cim_conv3x3_nchw_bn_act_bwd", P(dyd), P(yd), P(xr), P(xs), P(w), P(bnp[0]), P(bnp[2]), P(bnp[3]), 1e-5, 1, None,
                                P(dg), P(db), P(dxd), P(dwd), 1, c, c, H, W, stride, 1, P(wsb), st, None, None, None, 1, 0, None, None, 0.0,
                                None, None, None, None)
        res["hip_direct_bwd_us"] = timeit(bwd)
        res["hip_direct_fwd_bwd_us"] = res["hip_direct_fwd_us"] + res["hip_direct_bwd_us"]
    out.append(res)
    print(json.dumps(res))
