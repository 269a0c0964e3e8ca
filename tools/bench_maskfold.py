"""cim_wino7_dx_maskfold alone (Md [121][R][2Cb] -> dbox [R][7][7][Cb]); every cim_amd/libcim_hip_alt_mf*.so beside the product library.
    python tools/bench_maskfold.py [R] [Cb]"""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cim_amd import _lib
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
Cb = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
M = torch.randn(121, R, 2 * Cb, device=dev)
masks = (torch.rand(R, 7, 7, device=dev) > 0.5).float()
out = torch.empty(R, 7, 7, Cb, device=dev)
libs = [("product", None)] + [(os.path.basename(p)[len("libcim_hip_alt_"):-3], ctypes.CDLL(p)) for p in sorted(glob.glob(os.path.join(os.path.dirname(_lib.__file__), "libcim_hip_alt_mf*.so")))]
st = _lib.stream_ptr()
ref = None
for rnd in range(3):
    for name, lib in libs:
        if lib is None:
            f = lambda: _lib.call("cim_wino7_dx_maskfold", M.data_ptr(), masks.data_ptr(), out.data_ptr(), R, Cb, st)
        else:
            fn = lib.cim_wino7_dx_maskfold
            fn.argtypes = _lib.SIGNATURES["cim_wino7_dx_maskfold"]
            f = lambda fn=fn: fn(M.data_ptr(), masks.data_ptr(), out.data_ptr(), R, Cb, st)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        assert torch.equal(out, ref), name
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            f()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 50
        gb = (121 * R * 2 * Cb + R * 49 * Cb) * 4 / 1e9
        print("%-12s %.4f ms  %.2f TB/s" % (name, ms, gb / ms))
