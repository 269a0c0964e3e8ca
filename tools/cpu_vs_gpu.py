#!/usr/bin/env python3
"""Host time vs wall time of the benchmark step: is the step launch-bound?  (process CPU seconds per step next to wall
seconds per step; the difference is time the host spent blocked on the GPU.)"""
import os, sys, time
sys.argv = [sys.argv[0], "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

real_print = print
captured = {}
orig_perf = time.perf_counter

# run bench.main() pieces by hand: reuse its step through a tiny monkeypatch of its timing loop
src = open(bench.__file__).read()
src = src.replace("    for _ in range(args.warmup):\n        step()", "    for _ in range(6):\n        step()\n    torch.cuda.synchronize()\n    import time as _t\n    c0, w0 = _t.process_time(), _t.perf_counter()\n    for _ in range(20):\n        step()\n    torch.cuda.synchronize()\n    c1, w1 = _t.process_time(), _t.perf_counter()\n    print('host CPU %.2f ms / step, wall %.2f ms / step' % ((c1 - c0) * 50, (w1 - w0) * 50), file=sys.stderr)\n    import cProfile, pstats\n    pr = cProfile.Profile(); pr.enable()\n    for _ in range(20):\n        step()\n    torch.cuda.synchronize(); pr.disable()\n    pstats.Stats(pr, stream=sys.stderr).sort_stats('tottime').print_stats(28)")
exec(compile(src, bench.__file__, "exec"), {"__name__": "__main__", "__file__": bench.__file__})
