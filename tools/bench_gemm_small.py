#!/usr/bin/env python3
"""cim_gemm_small_f32 alone (direct C-ABI calls on preallocated buffers, GPU-bound loop): the forward / data-gradient /
weight-gradient products of the ResNet-50 C4 1 x 1 convolutions at cfg2 sizes, us per call and TFLOP/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
LAYERS = [("res2.conv1", 256, 64, 22188), ("res2.conv3", 64, 256, 22188), ("res3.conv1", 512, 128, 5590),
          ("res3.conv3", 128, 512, 5590), ("res4.conv1", 1024, 256, 1419), ("res4.conv3", 256, 1024, 1419),
          ("res4.0.conv1", 512, 256, 5590)]
lib = sys.argv[1] if len(sys.argv) > 1 else None
if lib:
    import ctypes
    alt = ctypes.CDLL(lib)
    for name in ("cim_gemm_small_f32", "cim_gemm_small_splits"):
        getattr(alt, name).argtypes = _lib.SIGNATURES[name]
    call = lambda name, *a: getattr(alt, name)(*a)
else:
    call = _lib.call
st = _lib.stream_ptr()


def timeit(fn, n=50):
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


tot = dict(fwd=0.0, dx=0.0, dw=0.0)
for name, cin, cout, hw in LAYERS:
    w = torch.randn(cout, cin, device=dev)
    x = torch.randn(cin, hw, device=dev)
    y = torch.empty(cout, hw, device=dev)
    xr = torch.empty(cout, hw, device=dev)
    g = [torch.rand(cout, device=dev) + 0.5 for _ in range(4)]
    dy = torch.randn(cout, hw, device=dev)
    dx = torch.empty(cin, hw, device=dev)
    dw = torch.empty(cout, cin, device=dev)
    ws = torch.empty(64 * max(cout * hw, cin * hw, cout * cin) // 8 + 1, device=dev)
    P = lambda t: t.data_ptr()
    sf, sx, sw = (call("cim_gemm_small_splits", cout, hw, cin), call("cim_gemm_small_splits", cin, hw, cout),
                  call("cim_gemm_small_splits", cout, cin, hw))
    fwd = lambda: call("cim_gemm_small_f32", P(w), P(x), P(y), cout, hw, cin, cin, hw, hw, 0, 0, P(xr), P(g[0]), P(g[1]), P(g[2]), P(g[3]),
                       1e-5, None, 1, sf, P(ws), st)
    fdx = lambda: call("cim_gemm_small_f32", P(w), P(dy), P(dx), cin, hw, cout, cin, hw, hw, 1, 0, None, None, None, None, None, 0.0, None, 0, sx, P(ws), st)
    fdw = lambda: call("cim_gemm_small_f32", P(dy), P(x), P(dw), cout, cin, hw, hw, hw, cin, 0, 1, None, None, None, None, None, 0.0, None, 0, sw, P(ws), st)
    flops = 2.0 * cin * cout * hw
    r = dict(layer=name, cin=cin, cout=cout, hw=hw, splits=(sf, sx, sw), fwd_us=timeit(fwd), dx_us=timeit(fdx), dw_us=timeit(fdw))
    for k in ("fwd", "dx", "dw"):
        r[k + "_TF"] = flops / r[k + "_us"] / 1e6
        tot[k] += r[k + "_us"]
    print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()}))
print(json.dumps({k: round(v, 1) for k, v in tot.items()}))
