"""Phase stamps of the mining launches 1 and 3 (needs the alt build: bash tools/build_alt.sh clk mining.hip -DCIM_MINING_CLOCKS=1;
run with CIM_HIP_LIB=cim_amd/libcim_hip_alt_clk.so)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from cim_amd import _lib, synthetic
import bench
dev = torch.device("cuda:0")
lib = ctypes.CDLL(os.environ["CIM_HIP_LIB"])
sys.argv = ["bench.py", "--steps", "6", "--warmup", "6", "--no-cpu-baseline", "--no-extra"]
bench.main()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 8 * 16))()
assert lib.cim_debug_mining_clocks(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(2, 8, 16).astype(np.int64)
for k, n in ((0, 5), (1, 9)):
    for wg in range(8):
        st = a[k, wg, :n]
        if st[0] == 0:
            continue
        print("kernel", k, "wg", wg, "start %d" % (st[0] - a[k][a[k][:, 0] > 0][:, 0].min()), "deltas (10 ns):", list(np.diff(st)))
