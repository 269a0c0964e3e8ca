import cProfile, pstats, sys, os, runpy, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.autograd.set_multithreading_enabled(False)
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-extra", "--steps", "100", "--warmup", "8"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(r"cim_amd|torch|built-in|method", 60)
out = s.getvalue()
print(out[out.index("ncalls"):][:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(r"cim_amd", 45)
out = s.getvalue()
print(out[out.index("ncalls"):][:7000])
