#!/usr/bin/env python3
"""cim_gemm_small_f32 against float64 for the four operand layouts (debugging aid; CIM_SMALL_KERNEL=tiled|direct)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cim_amd import _lib
dev = torch.device("cuda:0")
torch.manual_seed(0)
st = _lib.stream_ptr()
P = lambda t: t.data_ptr()
worst = 0.0
for (M, N, K) in [(128, 1530, 256), (128, 391, 512), (512, 391, 128), (64, 77, 64), (256, 1419, 1024), (100, 333, 200), (128, 256, 1419), (33, 65, 47), (256, 1024, 391)]:
    for am in (0, 1):
        for bk in (0, 1):
            A = torch.randn((K, M) if am else (M, K), device=dev)
            B = torch.randn((N, K) if bk else (K, N), device=dev)
            C = torch.full((M, N), float("nan"), device=dev)
            sp = _lib.call("cim_gemm_small_splits", M, N, K)
            ws = torch.empty(max(sp, 1) * M * N, device=dev)
            _lib.call("cim_gemm_small_f32", P(A), P(B), P(C), M, N, K, M if am else K, K if bk else N, N, am, bk, None, None, None, None, None, 0.0, None, 0, sp, P(ws), st)
            ref = (A.double().t() if am else A.double()) @ (B.double().t() if bk else B.double())
            err = float((C.double() - ref).abs().max() / ref.abs().max())
            worst = max(worst, err)
            flag = "" if err < 1e-5 else "   <<<<<< BAD"
            print("M=%d N=%d K=%d am=%d bk=%d splits=%d err=%.2e%s" % (M, N, K, am, bk, sp, err, flag))
print("worst", worst)
