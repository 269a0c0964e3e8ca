#!/usr/bin/env python3
"""Micro-benchmark of the mixed-tiling Winograd transform kernels at BASELINE cfg2 size (1000 ROIs, 2048 -> 1024
channels); libcim_hip_alt*.so builds next to the library are timed in the same process (A/B)."""
import ctypes
import glob
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
libs = {"base": _lib.load()}
for path in sorted(glob.glob(os.path.join(_lib.HERE, "libcim_hip_alt*.so"))):
    alt = ctypes.CDLL(path)
    for name, argt in _lib.SIGNATURES.items():
        getattr(alt, name).argtypes = argt
        getattr(alt, name).restype = ctypes.c_int
    libs[os.path.basename(path)[len("libcim_hip_"):-3]] = alt
st = torch.cuda.current_stream().cuda_stream
R, P, Cin, Cout, NPOS = 1000, 7, 2048, 1024, 121
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(R, P, P, Cin, device=dev, generator=g)
dy = torch.randn(R, P, P, Cout, device=dev, generator=g)
w = torch.randn(Cout, Cin, 3, 3, device=dev, generator=g)
V = torch.empty(NPOS, R, Cin, device=dev)
D = torch.empty(NPOS, R, Cout, device=dev)
U = torch.empty(NPOS, Cin, Cout, device=dev)
M = torch.randn(NPOS, R, Cout, device=dev, generator=g)
Md = torch.randn(NPOS, R, Cin, device=dev, generator=g)
dU = torch.randn(NPOS, Cin, Cout, device=dev, generator=g)
y = torch.empty(R, P, P, Cout, device=dev)
dx = torch.empty(R, P, P, Cin, device=dev)
dw = torch.empty(Cout, Cin, 3, 3, device=dev)
ra = torch.empty(NPOS * R, dtype=torch.int32, device=dev)
P_ = lambda t: t.data_ptr()


def cases(lib):
    return {
        "input(+bounds)  x->V": (lambda: lib.cim_wino_input_transform_amax(P_(x), P_(V), P_(ra), R, P, Cin, 7, st), (x.numel() + V.numel()) * 4),
        "filter          w->U": (lambda: lib.cim_wino_filter_transform(P_(w), P_(U), Cout, Cin, 0, 7, st), (w.numel() + U.numel()) * 4),
        "output          M->y": (lambda: lib.cim_wino_output_transform(P_(M), None, P_(y), R, P, Cout, 1, 7, st), (M.numel() + y.numel()) * 4),
        "dy (wgrad)     dy->D": (lambda: lib.cim_wino_dy_transform(P_(dy), P_(D), R, P, Cout, 7, st), (dy.numel() + D.numel()) * 4),
        "dy (adjoint)   dy->E": (lambda: lib.cim_wino_dy_adjoint_transform(P_(dy), P_(D), P_(ra), R, P, Cout, 7, st), (dy.numel() + D.numel()) * 4),
        "dx (adjoint)  Md->dx": (lambda: lib.cim_wino_dx_adjoint_output(P_(Md), P_(dx), R, P, Cin, 7, st), (Md.numel() + dx.numel()) * 4),
        "wgrad out     dU->dW": (lambda: lib.cim_wino_wgrad_output(P_(dU), P_(dw), Cout, Cin, 7, st), (dU.numel() + dw.numel()) * 4),
    }


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        assert fn() == 0
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


names = list(cases(libs["base"]))
res = {k: {n: [] for n in names} for k in libs}
for rnd in range(3):
    for k in libs:
        for n, (fn, _) in cases(libs[k]).items():
            res[k][n].append(timeit(fn))
for n in names:
    nbytes = cases(libs["base"])[n][1]
    print("%-22s " % n + "   ".join("%s %.3f ms (%.2f TB/s)" % (k, min(res[k][n]), nbytes / min(res[k][n]) / 1e9) for k in libs))
