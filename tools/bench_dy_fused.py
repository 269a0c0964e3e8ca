"""cim_wino7_flatten_bwd_dy_pair alone (dX [R][C*49] + saved conv output -> E', D' pair images) beside the three launches it replaces;
every cim_amd/libcim_hip_alt_fb*.so is timed next to the product library.    python tools/bench_dy_fused.py [R] [C]"""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cim_amd import _lib
from cim_amd.ops import pair
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
Rs = pair.pad32(R)
dX = torch.randn(R, C * 49, device=dev)
y = torch.randn(R, 7, 7, C, device=dev)
dy = torch.empty(R, 7, 7, C, device=dev)
bp = torch.empty(R, C, device=dev)
E, D = torch.empty(121, Rs, C, dtype=torch.int32, device=dev), torch.empty(121, Rs, C, dtype=torch.int32, device=dev)
amax = dX.abs().max().reshape(1).view(torch.int32)
sE, sD = torch.empty(121, device=dev), torch.empty(121, device=dev)
st = _lib.stream_ptr()
_lib.call("cim_wino7_pair_scales", amax.data_ptr(), 1, None, 3, sE.data_ptr(), st)
_lib.call("cim_wino7_pair_scales", amax.data_ptr(), 1, None, 2, sD.data_ptr(), st)
P = lambda t: t.data_ptr()


def timeit(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


libs = [("product", None)] + [(os.path.basename(p)[len("libcim_hip_alt_"):-3], ctypes.CDLL(p)) for p in sorted(glob.glob(os.path.join(os.path.dirname(_lib.__file__), "libcim_hip_alt_fb*.so")))]
gb = (2 * R * C * 49 + 2 * 121 * Rs * C) * 4 / 1e9
for rnd in range(3):
    for name, lib in libs:
        if lib is None:
            f = lambda: _lib.call("cim_wino7_flatten_bwd_dy_pair", P(dX), P(y), P(E), P(sE), P(D), P(sD), P(bp), R, Rs, C, st)
        else:
            fn = lib.cim_wino7_flatten_bwd_dy_pair
            fn.argtypes = _lib.SIGNATURES["cim_wino7_flatten_bwd_dy_pair"]
            f = lambda fn=fn: fn(P(dX), P(y), P(E), P(sE), P(D), P(sD), P(bp), R, Rs, C, st)
        ms = timeit(f)
        print("%-10s one launch  %.4f ms  %.2f TB/s" % (name, ms, gb / ms))
    t1 = timeit(lambda: _lib.call("cim_flatten_chw_bwd_bias", P(dX), P(y), P(dy), P(bp), R, 49, C, st))
    t2 = timeit(lambda: _lib.call("cim_wino7_dy_pair", P(dy), P(E), P(sE), R, Rs, C, 1, st))
    t3 = timeit(lambda: _lib.call("cim_wino7_dy_pair", P(dy), P(D), P(sD), R, Rs, C, 0, st))
    print("three launches: flatten_bwd %.4f + dy_pair(adjoint) %.4f + dy_pair(wgrad) %.4f = %.4f ms" % (t1, t2, t3, t1 + t2 + t3))
