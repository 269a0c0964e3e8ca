"""cim_wino_wgrad_output alone (dU [121][Cin][Cout] -> dW [Cout][Cin][3][3]); every cim_amd/libcim_hip_alt_wg*.so beside the product."""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from cim_amd import _lib
Cout = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = torch.device("cuda:0")
dU = torch.randn(121, Cin, Cout, device=dev)
dW = torch.empty(Cout, Cin, 3, 3, device=dev)
st = _lib.stream_ptr()
libs = [("product", None)] + [(os.path.basename(p)[len("libcim_hip_alt_"):-3], ctypes.CDLL(p)) for p in sorted(glob.glob(os.path.join(os.path.dirname(_lib.__file__), "libcim_hip_alt_wg*.so")))]
ref = None
for rnd in range(3):
    for name, lib in libs:
        if lib is None:
            f = lambda: _lib.call("cim_wino_wgrad_output", dU.data_ptr(), dW.data_ptr(), Cout, Cin, 7, st)
        else:
            fn = lib.cim_wino_wgrad_output
            fn.argtypes = _lib.SIGNATURES["cim_wino_wgrad_output"]
            f = lambda fn=fn: fn(dU.data_ptr(), dW.data_ptr(), Cout, Cin, 7, st)
        dW.zero_()
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        if ref is None:
            ref = dW.clone()
        same = bool(torch.equal(dW, ref))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            f()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 50
        print("%-10s %.4f ms  %.2f TB/s  %s" % (name, ms, (121 + 9) * Cin * Cout * 4 / 1e9 / ms, "bit-identical" if same else "DIFFERENT (or a probe)"))
