#!/usr/bin/env python3
"""The step's contraction shapes (cfg2: 1000 proposals, Cf = 1024, mixed 4 + 3 Winograd tiling = 121 positions) on the
f16x2p engine (pre-split pair images, cim_gemm_pair*) next to the f16x2 engine (split in the main loop, cim_gemm_f16x2*):
interleaved rounds in one process, median ms, executed f16 TFLOP/s (3 products per multiply-add).
--n ROIS, --json PATH, --only NAME[,NAME]"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib  # noqa: E402
from cim_amd.ops import pair  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--json", default=None)
ap.add_argument("--only", default=None)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--no-old", action="store_true")
ap.add_argument("--no-alts", action="store_true", help="ignore the ablation builds next to the product library")
ap.add_argument("--one-product", action="store_true", help="the h * h term alone (TF32-class; bench.py's extra.tf32_class)")
args = ap.parse_args()

dev = torch.device("cuda:0")
lib = _lib.load()
# ablation / variant builds of the library next to the product one (python -m cim_amd.build --out=cim_amd/libcim_hip_alt<tag>.so
# with CIM_HIPCC_FLAGS=-DCIM_PAIR_EXP=n): their pair GEMMs are timed in the same interleaved rounds
import ctypes
import glob
alts = {}
for path in ([] if args.no_alts else sorted(glob.glob(os.path.join(_lib.HERE, "libcim_hip_alt*.so")))):
    a = ctypes.CDLL(path)
    for name, argt in _lib.SIGNATURES.items():
        getattr(a, name).argtypes = argt
        getattr(a, name).restype = ctypes.c_int
    alts[os.path.basename(path)[len("libcim_hip_alt"):-3]] = a
st = torch.cuda.current_stream().cuda_stream
N, C, NPOS = args.n, 1024, 121
NP = pair.pad32(N)
g = torch.Generator(device=dev).manual_seed(0)
P = lambda t: t.data_ptr()


def rnd(*shape):
    return torch.randn(*shape, device=dev, generator=g)


V = torch.zeros(NPOS, NP, 2 * C, device=dev)
V[:, :N] = rnd(NPOS, N, 2 * C)
U = rnd(NPOS, 2 * C, C)
D = torch.zeros(NPOS, NP, C, device=dev)
D[:, :N] = rnd(NPOS, N, C)
M = torch.empty(NPOS, N, C, device=dev)
M2 = torch.empty(NPOS, N, 2 * C, device=dev)
dU = torch.empty(NPOS, 2 * C, C, device=dev)
K1 = 49 * C
xf = torch.zeros(NP, K1, device=dev)
xf[:N] = rnd(N, K1)
w1 = rnd(4096, K1) * 0.01
dyf = torch.zeros(NP, 4096, device=dev)
dyf[:N] = rnd(N, 4096)
y1 = torch.empty(N, 4096, device=dev)
dx1 = torch.empty(N, K1, device=dev)
dw1 = torch.empty(4096, K1, device=dev)
ws = torch.empty(16 * N * 4096 + 16, device=dev)

pV = pair.split(V, NP, 2 * C, 2 * C, batch=NPOS, x_bs=NP * 2 * C)
pU = pair.split(U, 2 * C, C, C, batch=NPOS, x_bs=2 * C * C)
pD = pair.split(D, NP, C, C, batch=NPOS, x_bs=NP * C)
pX = pair.split(xf)
pW = pair.split(w1)
pY = pair.split(dyf)


def zi(n):
    return torch.zeros(n, dtype=torch.int32, device=dev)


def xlib():
    """the superseded f16x2 engine (experiments/libcim_exp.so): the comparison rows of this table"""
    from experiments import _lib as _x
    return _x.load()


def _amax(x, rows, cols, ld, want_rows, want_cols, batch=1, bs=0):
    ra = zi(batch * rows) if want_rows else None
    ca = zi(batch * cols) if want_cols else None
    assert xlib().cim_amax_rowcol(P(x), rows, cols, ld, batch, bs, _lib.ptr(ra), _lib.ptr(ca), st) == 0
    return ra, ca


cases = {}
PRODUCTS = 1 if args.one_product else 3      # --one-product: the h * h term alone (TF32-class, extra.tf32_class of bench.py)
sp = lambda m, n, k: lib.cim_gemm_pair_splits(m, n, k)
fl_conv = NPOS * 2.0 * N * 2 * C * C
fl_fc = 2.0 * N * K1 * 4096
cases["pair wino_fwd   (KC x MC)"] = (lambda: lib.cim_gemm_pair_batched(P(pV.buf), P(pU.buf), P(M), N, C, 2 * C, 2 * C, C, C, 0, 0, NPOS, pV.bs, pU.bs, N * C, P(pV.scale), P(pU.scale), 0, PRODUCTS, 0, st), fl_conv)
cases["pair wino_dgrad (KC x KC)"] = (lambda: lib.cim_gemm_pair_batched(P(pD.buf), P(pU.buf), P(M2), N, 2 * C, C, C, C, 2 * C, 0, 1, NPOS, pD.bs, pU.bs, N * 2 * C, P(pD.scale), P(pU.scale), 0, PRODUCTS, 0, st), fl_conv)
cases["pair wino_wgrad (MC x MC)"] = (lambda: lib.cim_gemm_pair_batched(P(pV.buf), P(pD.buf), P(dU), 2 * C, C, NP, 2 * C, C, C, 1, 0, NPOS, pV.bs, pD.bs, 2 * C * C, P(pV.scale), P(pD.scale), 0, PRODUCTS, 0, st), fl_conv)
cases["pair fc1_fwd    (KC x KC)"] = (lambda: lib.cim_gemm_pair(P(pX.buf), P(pW.buf), P(y1), None, N, 4096, K1, K1, K1, 4096, 0, 1, 0, sp(N, 4096, K1), P(ws), P(pX.scale), P(pW.scale), None, 0, PRODUCTS, 0, st), fl_fc)
cases["pair fc1_dgrad  (KC x MC)"] = (lambda: lib.cim_gemm_pair(P(pY.buf), P(pW.buf), P(dx1), None, N, K1, 4096, 4096, K1, K1, 0, 0, 0, 1, None, P(pY.scale), P(pW.scale), None, 0, PRODUCTS, 0, st), fl_fc)
cases["pair fc1_wgrad  (MC x MC)"] = (lambda: lib.cim_gemm_pair(P(pY.buf), P(pX.buf), P(dw1), None, 4096, K1, NP, 4096, K1, K1, 1, 0, 0, 1, None, P(pY.scale), P(pX.scale), None, 0, PRODUCTS, 0, st), fl_fc)
# the co-resident form (128 x 256 tiles of four waves) of the two weight-gradient products
cases["form1 wino_wgrad"] = (lambda: lib.cim_gemm_pair_batched(P(pV.buf), P(pD.buf), P(dU), 2 * C, C, NP, 2 * C, C, C, 1, 0, NPOS, pV.bs, pD.bs, 2 * C * C, P(pV.scale), P(pD.scale), 0, PRODUCTS, 1, st), fl_conv)
cases["form1 fc1_wgrad"] = (lambda: lib.cim_gemm_pair(P(pY.buf), P(pX.buf), P(dw1), None, 4096, K1, NP, 4096, K1, K1, 1, 0, 0, 1, None, P(pY.scale), P(pX.scale), None, 0, PRODUCTS, 1, st), fl_fc)
if os.environ.get("CIM_BENCH_FORM1_LIMITS"):        # (diagnosis: per-CU latency or shared bandwidth?  launches of n workgroups)
    for lim in (64, 128, 256, 512):
        cases["form1 fc1_wgrad launches of %d" % lim] = (lambda lim=lim: lib.cim_gemm_pair(P(pY.buf), P(pX.buf), P(dw1), None, 4096, K1, NP, 4096, K1, K1, 1, 0, 0, 1, None, P(pY.scale), P(pX.scale), None, lim, PRODUCTS, 1, st), fl_fc)
        cases["pair fc1_wgrad launches of %d" % lim] = (lambda lim=lim: lib.cim_gemm_pair(P(pY.buf), P(pX.buf), P(dw1), None, 4096, K1, NP, 4096, K1, K1, 1, 0, 0, 1, None, P(pY.scale), P(pX.scale), None, lim, PRODUCTS, 0, st), fl_fc)
cases["split V (generic producer)"] = (lambda: lib.cim_pair_split(P(V), P(pV.buf), NP, NP, 2 * C, 2 * C, 2 * C, NPOS, NP * 2 * C, pV.bs, P(pV.scale), None, st), 0.0)
for tag, al in alts.items():
    cases["%-4s wino_fwd" % tag] = (lambda al=al: al.cim_gemm_pair_batched(P(pV.buf), P(pU.buf), P(M), N, C, 2 * C, 2 * C, C, C, 0, 0, NPOS, pV.bs, pU.bs, N * C, P(pV.scale), P(pU.scale), 0, PRODUCTS, 0, st), fl_conv)
    cases["%-4s wino_dgrad" % tag] = (lambda al=al: al.cim_gemm_pair_batched(P(pD.buf), P(pU.buf), P(M2), N, 2 * C, C, C, C, 2 * C, 0, 1, NPOS, pD.bs, pU.bs, N * 2 * C, P(pD.scale), P(pU.scale), 0, PRODUCTS, 0, st), fl_conv)
    cases["%-4s wino_wgrad" % tag] = (lambda al=al: al.cim_gemm_pair_batched(P(pV.buf), P(pD.buf), P(dU), 2 * C, C, NP, 2 * C, C, C, 1, 0, NPOS, pV.bs, pD.bs, 2 * C * C, P(pV.scale), P(pD.scale), 0, PRODUCTS, 0, st), fl_conv)
    cases["%-4s fc1_fwd" % tag] = (lambda al=al: al.cim_gemm_pair(P(pX.buf), P(pW.buf), P(y1), None, N, 4096, K1, K1, K1, 4096, 0, 1, 0, sp(N, 4096, K1), P(ws), P(pX.scale), P(pW.scale), None, 0, PRODUCTS, 0, st), fl_fc)
    cases["%-4s form1 wino_wgrad" % tag] = (lambda al=al: al.cim_gemm_pair_batched(P(pV.buf), P(pD.buf), P(dU), 2 * C, C, NP, 2 * C, C, C, 1, 0, NPOS, pV.bs, pD.bs, 2 * C * C, P(pV.scale), P(pD.scale), 0, PRODUCTS, 1, st), fl_conv)
    cases["%-4s form1 fc1_wgrad" % tag] = (lambda al=al: al.cim_gemm_pair(P(pY.buf), P(pX.buf), P(dw1), None, 4096, K1, NP, 4096, K1, K1, 1, 0, 0, 1, None, P(pY.scale), P(pX.scale), None, 0, PRODUCTS, 1, st), fl_fc)
    cases["%-4s fc1_dgrad" % tag] = (lambda al=al: al.cim_gemm_pair(P(pY.buf), P(pW.buf), P(dx1), None, N, K1, 4096, 4096, K1, K1, 0, 0, 0, 1, None, P(pY.scale), P(pW.scale), None, 0, PRODUCTS, 0, st), fl_fc)
    cases["%-4s fc1_wgrad" % tag] = (lambda al=al: al.cim_gemm_pair(P(pY.buf), P(pX.buf), P(dw1), None, 4096, K1, NP, 4096, K1, K1, 1, 0, 0, 1, None, P(pY.scale), P(pX.scale), None, 0, PRODUCTS, 0, st), fl_fc)
if not args.no_old:
    Vr, Vc = _amax(V, NP, 2 * C, 2 * C, True, True, NPOS, NP * 2 * C)
    _, Uc = _amax(U, 2 * C, C, C, False, True, NPOS, 2 * C * C)
    Ur, _ = _amax(U, 2 * C, C, C, True, False, NPOS, 2 * C * C)
    Dr, Dc = _amax(D, NP, C, C, True, True, NPOS, NP * C)
    xr, xc = _amax(xf, NP, K1, K1, True, True)
    wr, wc = _amax(w1, 4096, K1, K1, True, True)
    dr, dc = _amax(dyf, NP, 4096, 4096, True, True)
    so = lambda m, n, k: min(xlib().cim_gemm_f16x2_splits(m, n, k), 16)
    cases["f16x2 wino_fwd"] = (lambda: xlib().cim_gemm_f16x2_batched(P(V), P(U), P(M), N, C, 2 * C, 2 * C, C, C, 0, 0, NPOS, NP * 2 * C, 2 * C * C, N * C, P(Vr), P(Uc), st), fl_conv)
    cases["f16x2 wino_dgrad"] = (lambda: xlib().cim_gemm_f16x2_batched(P(D), P(U), P(M2), N, 2 * C, C, C, C, 2 * C, 0, 1, NPOS, NP * C, 2 * C * C, N * 2 * C, P(Dr), P(Ur), st), fl_conv)
    cases["f16x2 wino_wgrad"] = (lambda: xlib().cim_gemm_f16x2_batched(P(V), P(D), P(dU), 2 * C, C, N, 2 * C, C, C, 1, 0, NPOS, NP * 2 * C, NP * C, 2 * C * C, P(Vc), P(Dc), st), fl_conv)
    cases["f16x2 fc1_fwd"] = (lambda: xlib().cim_gemm_f16x2(P(xf), P(w1), P(y1), None, N, 4096, K1, K1, K1, 4096, 0, 1, 0, so(N, 4096, K1), P(ws), P(xr), P(wr), st), fl_fc)
    cases["f16x2 fc1_dgrad"] = (lambda: xlib().cim_gemm_f16x2(P(dyf), P(w1), P(dx1), None, N, K1, 4096, 4096, K1, K1, 0, 0, 0, 1, None, P(dr), P(wc), st), fl_fc)
    cases["f16x2 fc1_wgrad"] = (lambda: xlib().cim_gemm_f16x2(P(dyf), P(xf), P(dw1), None, 4096, K1, N, 4096, K1, K1, 1, 0, 0, 1, None, P(dc), P(xc), st), fl_fc)
if args.only:
    keep = args.only.split(",")
    cases = {k: v for k, v in cases.items() if any(s in k for s in keep)}


def timeit(fn, n=3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        rc = fn()
        assert rc == 0, (rc, lib.cim_last_error())
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


# correctness spot check of the three conv products against the f16x2 engine's results
if not args.no_old and not args.only:
    for a, b, out in (("pair wino_fwd   (KC x MC)", "f16x2 wino_fwd", M), ("pair wino_dgrad (KC x KC)", "f16x2 wino_dgrad", M2),
                      ("pair wino_wgrad (MC x MC)", "f16x2 wino_wgrad", dU), ("pair fc1_fwd    (KC x KC)", "f16x2 fc1_fwd", y1),
                      ("pair fc1_dgrad  (KC x MC)", "f16x2 fc1_dgrad", dx1), ("pair fc1_wgrad  (MC x MC)", "f16x2 fc1_wgrad", dw1)):
        assert cases[a][0]() == 0, lib.cim_last_error()
        r1 = out.clone()
        assert cases[b][0]() == 0, lib.cim_last_error()
        err = float((r1 - out).abs().max() / out.abs().max())
        print("check %-28s vs %-18s rel diff %.2e" % (a, b, err), flush=True)

res = {k: [] for k in cases}
for k, (fn, _) in cases.items():
    timeit(fn, 1)
for _ in range(args.rounds):
    for k, (fn, _) in cases.items():
        res[k].append(timeit(fn))
out = {}
for k, (fn, fl) in cases.items():
    ms = statistics.median(res[k])
    out[k] = {"ms": ms, "algorithmic_tflops": fl / ms / 1e9, "executed_f16_tflops": 3 * fl / ms / 1e9, "frac_of_2516.6": 3 * fl / ms / 1e9 / 2516.6}
    print("%-30s %.3f ms   %.0f TF algorithmic   %.0f TF executed = %.3f of peak" % (k, ms, fl / ms / 1e9, 3 * fl / ms / 1e9, 3 * fl / ms / 1e9 / 2516.6), flush=True)
if args.json:
    json.dump({"n": N, "cases": out}, open(args.json, "w"), indent=1)
