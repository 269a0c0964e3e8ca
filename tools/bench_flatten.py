#!/usr/bin/env python3
"""Micro-benchmark of cim_flatten_chw (forward / backward with the ReLU mask) at cfg2 size; libcim_hip_alt*.so builds are
timed in the same process."""
import ctypes, glob, os, sys
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
R, PP, C = 1000, 49, 1024
y = torch.randn(R, PP, C, device=dev)
flat = torch.empty(R, C * PP, device=dev)
dflat = torch.randn(R, C * PP, device=dev)
dy = torch.empty(R, PP, C, device=dev)


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        assert fn() == 0
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ref = None
for k, lib in libs.items():
    f = timeit(lambda: lib.cim_flatten_chw(y.data_ptr(), None, flat.data_ptr(), R, PP, C, 0, st))
    b = timeit(lambda: lib.cim_flatten_chw(dflat.data_ptr(), y.data_ptr(), dy.data_ptr(), R, PP, C, 1, st))
    cur = (flat.clone(), dy.clone())
    same = "" if ref is None else " identical=%s" % (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]))
    ref = ref or cur
    print("%-10s fwd %.4f ms (%.2f TB/s)  bwd %.4f ms (%.2f TB/s)%s" % (k, f, 2 * y.numel() * 4 / f / 1e9, b, 3 * y.numel() * 4 / b / 1e9, same))
