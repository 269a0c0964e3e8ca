#!/usr/bin/env python3
"""Backbone (a-11, ATen/MIOpen) forward+backward time: NCHW vs channels_last parameters/input."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd.core.presets import apply_preset
apply_preset(sys.argv[1] if len(sys.argv) > 1 else "resnet50_voc")
from cim_amd.modeling.model_builder import get_func
from cim_amd.core.config import cfg
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(1, 3, 516, 688, device=dev)
for fmt in ("nchw", "channels_last"):
    body = get_func(cfg.MODEL.CONV_BODY)().to(dev).train()
    xi = x
    if fmt == "channels_last":
        body = body.to(memory_format=torch.channels_last)
        xi = x.contiguous(memory_format=torch.channels_last)
    def step():
        y = body(xi)
        y.backward(torch.ones_like(y))
    for _ in range(3): step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): step()
    b.record(); torch.cuda.synchronize()
    print(fmt, "fwd+bwd ms", a.elapsed_time(b) / 10)
