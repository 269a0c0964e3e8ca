"""Phase stamps of the loss launch's jobs (needs the alt build: bash tools/build_alt.sh lclk losses.hip -DCIM_LOSS_CLOCKS=1; run with
CIM_HIP_LIB=cim_amd/libcim_hip_alt_lclk.so).  Stamps per job: 0 entry, 1 gradients zeroed, 2 row pass 1, 3 sums, 4 row pass 2,
5 column pass, 6 stored, 7 exit (refinement jobs); 0 entry, 1 zeroed, 2 cluster plan, 3 column jobs, 7 exit (PCL job)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
lib = ctypes.CDLL(os.environ["CIM_HIP_LIB"])
sys.argv = ["bench.py", "--steps", "6", "--warmup", "6", "--no-cpu-baseline", "--no-extra"]
bench.main()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 16))()
assert lib.cim_debug_loss_clocks(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(8, 16).astype(np.int64)
t0 = a[a[:, 0] > 0][:, 0].min()
for job in range(8):
    st = a[job, :8]
    if st[0] == 0:
        continue
    print("job", job, "stamps (us from the first job's entry):", [round((x - t0) / 100.0, 1) if x else None for x in st])
