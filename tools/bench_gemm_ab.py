#!/usr/bin/env python3
"""Within-process A/B of two builds of the GEMM kernel (libcim_hip.so vs libcim_hip_alt.so) on the
step's contraction shapes; interleaved rounds, median ms.  CIM_AB_ENGINE=f16x2 (default) times the f16x2 entry
points (operand scales precomputed; the amax passes are timed separately), bf16x3 / fp32 the cim_gemm_f32 ones."""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
libs = {"base": _lib.load()}
import glob
for path in sorted(glob.glob(os.path.join(_lib.HERE, "libcim_hip_alt*.so"))):
    alt = ctypes.CDLL(path)
    for name, argt in _lib.SIGNATURES.items():
        getattr(alt, name).argtypes = argt
        getattr(alt, name).restype = ctypes.c_int
    libs[os.path.basename(path)[len("libcim_hip_"):-3]] = alt
st = torch.cuda.current_stream().cuda_stream
N, C = 1000, 1024
NPOS = 36
mt = N * 4
g = torch.Generator(device=dev).manual_seed(0)
V = torch.randn(NPOS, mt, 2 * C, device=dev, generator=g)
U = torch.randn(NPOS, 2 * C, C, device=dev, generator=g)
M = torch.empty(NPOS, mt, C, device=dev)
D = torch.randn(NPOS, mt, C, device=dev, generator=g)
dU = torch.empty(NPOS, 2 * C, C, device=dev)
K1 = 49 * C
xf = torch.randn(N, K1, device=dev, generator=g)
w1 = torch.randn(4096, K1, device=dev, generator=g) * 0.01
dyf = torch.randn(N, 4096, device=dev, generator=g)
y1 = torch.empty(N, 4096, device=dev)
dx1 = torch.empty(N, K1, device=dev)
dw1 = torch.empty(4096, K1, device=dev)
ws = torch.empty(4 * 4096 * 50176, device=dev)     # 3.3 GB: up to 4 splits of the largest C


ENGINE = os.environ.get("CIM_AB_ENGINE", "f16x2")
ENG = 0 if ENGINE == "fp32" else 1          # `engine` argument of the cim_gemm_f32* entry points


def _amax(x, rows, cols, ld, want_rows, want_cols, batch=1, bs=0):
    ra = torch.zeros(batch * rows, dtype=torch.int32, device=dev) if want_rows else None
    ca = torch.zeros(batch * cols, dtype=torch.int32, device=dev) if want_cols else None
    rc = libs["base"].cim_amax_rowcol(x.data_ptr(), rows, cols, ld, batch, bs, _lib.ptr(ra), _lib.ptr(ca), st)
    assert rc == 0
    return ra, ca


Vr, Vc = _amax(V, mt, 2 * C, 2 * C, True, True, NPOS, mt * 2 * C)
_, Uc = _amax(U, 2 * C, C, C, False, True, NPOS, 2 * C * C)
_, Dc = _amax(D, mt, C, C, False, True, NPOS, mt * C)
xr, xc = _amax(xf, N, K1, K1, True, True)
wr, wc = _amax(w1, 4096, K1, K1, True, True)
dr, dc = _amax(dyf, N, 4096, 4096, True, True)
P = lambda t: t.data_ptr()


def cases_f16x2(lib):
    sp = lambda m, n, k: min(lib.cim_gemm_f16x2_splits(m, n, k), (4 * 4096 * 50176) // (m * n))
    return {
        "wino_fwd": (lambda: lib.cim_gemm_f16x2_batched(P(V), P(U), P(M), mt, C, 2 * C, 2 * C, C, C, 0, 0, NPOS, mt * 2 * C, 2 * C * C, mt * C, P(Vr), P(Uc), st), NPOS * 2.0 * mt * 2 * C * C),
        "wino_wgrad": (lambda: lib.cim_gemm_f16x2_batched(P(V), P(D), P(dU), 2 * C, C, mt, 2 * C, C, C, 1, 0, NPOS, mt * 2 * C, mt * C, 2 * C * C, P(Vc), P(Dc), st), NPOS * 2.0 * mt * 2 * C * C),
        "fc1_fwd": (lambda: lib.cim_gemm_f16x2(P(xf), P(w1), P(y1), None, N, 4096, K1, K1, K1, 4096, 0, 1, 0, sp(N, 4096, K1), P(ws), P(xr), P(wr), st), 2.0 * N * K1 * 4096),
        "fc1_dgrad": (lambda: lib.cim_gemm_f16x2(P(dyf), P(w1), P(dx1), None, N, K1, 4096, 4096, K1, K1, 0, 0, 0, sp(N, K1, 4096), P(ws), P(dr), P(wc), st), 2.0 * N * K1 * 4096),
        "fc1_wgrad": (lambda: lib.cim_gemm_f16x2(P(dyf), P(xf), P(dw1), None, 4096, K1, N, 4096, K1, K1, 1, 0, 0, sp(4096, K1, N), P(ws), P(dc), P(xc), st), 2.0 * N * K1 * 4096),
        "amax_V(rows+cols)": (lambda: lib.cim_amax_rowcol(P(V), mt, 2 * C, 2 * C, NPOS, mt * 2 * C, P(Vr), P(Vc), st), 0.0),
        "amax_U(cols)": (lambda: lib.cim_amax_rowcol(P(U), 2 * C, C, C, NPOS, 2 * C * C, None, P(Uc), st), 0.0),
        "amax_w1(rows+cols)": (lambda: lib.cim_amax_rowcol(P(w1), 4096, K1, K1, 1, 0, P(wr), P(wc), st), 0.0),
        "amax_x(rows+cols)": (lambda: lib.cim_amax_rowcol(P(xf), N, K1, K1, 1, 0, P(xr), P(xc), st), 0.0),
    }


def cases(lib):
    if ENGINE == "f16x2":
        return cases_f16x2(lib)
    sp = lambda m, n, k: min(lib.cim_gemm_f32_splits(m, n, k, ENG), (4 * 4096 * 50176) // (m * n))
    return {
        "wino_fwd": (lambda: lib.cim_gemm_f32_batched(V.data_ptr(), U.data_ptr(), M.data_ptr(), mt, C, 2 * C, 2 * C, C, C, 0, 0, NPOS, mt * 2 * C, 2 * C * C, mt * C, ENG, st), NPOS * 2.0 * mt * 2 * C * C),
        "wino_wgrad": (lambda: lib.cim_gemm_f32_batched(V.data_ptr(), D.data_ptr(), dU.data_ptr(), 2 * C, C, mt, 2 * C, C, C, 1, 0, NPOS, mt * 2 * C, mt * C, 2 * C * C, ENG, st), NPOS * 2.0 * mt * 2 * C * C),
        "fc1_fwd": (lambda: lib.cim_gemm_f32(xf.data_ptr(), w1.data_ptr(), y1.data_ptr(), None, N, 4096, K1, K1, K1, 4096, 0, 1, 0, sp(N, 4096, K1), ws.data_ptr(), ENG, st), 2.0 * N * K1 * 4096),
        "fc1_dgrad": (lambda: lib.cim_gemm_f32(dyf.data_ptr(), w1.data_ptr(), dx1.data_ptr(), None, N, K1, 4096, 4096, K1, K1, 0, 0, 0, sp(N, K1, 4096), ws.data_ptr(), ENG, st), 2.0 * N * K1 * 4096),
        "fc1_wgrad": (lambda: lib.cim_gemm_f32(dyf.data_ptr(), xf.data_ptr(), dw1.data_ptr(), None, 4096, K1, N, 4096, K1, K1, 1, 0, 0, sp(4096, K1, N), ws.data_ptr(), ENG, st), 2.0 * N * K1 * 4096),
    }


def timeit(fn, n=3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        rc = fn()
        assert rc == 0, rc
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {k: {n: [] for n in cases(libs["base"])} for k in libs}
for k in libs:
    for n, (fn, _) in cases(libs[k]).items():
        timeit(fn, 1)
for rnd in range(5):
    for k in libs:
        for n, (fn, fl) in cases(libs[k]).items():
            res[k][n].append(timeit(fn))
fl = {n: f for n, (_, f) in cases(libs["base"]).items()}
for n in fl:
    print("%-19s " % n + "   ".join("%s %.3f ms (%.1f TF)" % (k, statistics.median(res[k][n]), fl[n] / statistics.median(res[k][n]) / 1e9) for k in libs))

if "--clocks" in sys.argv:      # shader clock while the dominant GEMM runs back to back for ~2 s (power / thermal state)
    import glob as _g
    import threading
    import time
    files = _g.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
    samples, stop = [], False

    def poll():
        while not stop:
            for f in files:
                try:
                    cur = [l for l in open(f).read().splitlines() if l.strip().endswith("*")]
                    if cur:
                        samples.append(cur[0].split(":")[1].strip().rstrip("*").strip())
                except OSError:
                    pass
            time.sleep(0.05)

    th = threading.Thread(target=poll)
    th.start()
    fn = cases(libs["base"])["wino_fwd"][0]
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
    stop = True
    th.join()
    print("sclk samples under load (%d files):" % len(files), sorted(set(samples)), "n=%d" % len(samples))
    print("steady-state wino_fwd: %.3f ms" % timeit(fn, 20))
