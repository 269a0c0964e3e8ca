#!/usr/bin/env python3
"""Backbone 3 x 3 convolutions (+ BatchNorm + ReLU), forward and backward: the implicit-GEMM HIP op against ATen / MIOpen
on the C4 body's layer shapes at cfg2 (516 x 688 image).  us per call (whole backward = BN / ReLU backward + dX + dW)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    out.append(res)
    print(json.dumps(res))
