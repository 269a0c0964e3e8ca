#!/usr/bin/env python3
"""Backbone 1 x 1 convolution + BatchNorm (+ ReLU) layers of ResNet-50 C4 at cfg2 (516 x 688 image): own fused kernel
(cim_amd.ops.conv1x1_bn_act) vs ATen (MIOpen / rocBLAS) + the fused bn_act launch, forward and forward + backward,
us per layer and TFLOP/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd.ops import bn_act, conv1x1_bn_act  # noqa: E402

dev = torch.device("cuda:0")
LAYERS = [("res2.conv1", 256, 64, 129, 172), ("res2.conv3", 64, 256, 129, 172), ("res3.conv1", 512, 128, 65, 86),
          ("res3.conv3", 128, 512, 65, 86), ("res4.conv1", 1024, 256, 33, 43), ("res4.conv3", 256, 1024, 33, 43),
          ("res4.0.conv1", 512, 256, 65, 86)]


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


rows = []
for name, cin, cout, H, W in LAYERS:
    conv = torch.nn.Conv2d(cin, cout, 1, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(cout).to(dev).eval()
    x = torch.randn(1, cin, H, W, device=dev, requires_grad=True)
    up = torch.randn(1, cout, H, W, device=dev)
    flops = 2.0 * cin * cout * H * W

    def own_f():
        with torch.no_grad():
            return conv1x1_bn_act(x, conv, bn)

    def aten_f():
        with torch.no_grad():
            return bn_act(conv(x), bn)

    def own_fb():
        y = conv1x1_bn_act(x, conv, bn)
        y.backward(up)

    def aten_fb():
        y = bn_act(conv(x), bn)
        y.backward(up)

    r = dict(layer=name, cin=cin, cout=cout, hw=H * W, own_fwd_us=timeit(own_f), aten_fwd_us=timeit(aten_f),
             own_fwdbwd_us=timeit(own_fb), aten_fwdbwd_us=timeit(aten_fb))
    r["own_fwd_TF"] = flops / r["own_fwd_us"] / 1e6
    r["own_bwd_TF"] = 2 * flops / (r["own_fwdbwd_us"] - r["own_fwd_us"]) / 1e6
    rows.append(r)
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()}))
